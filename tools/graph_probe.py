"""Dev tool (GPU box): single-episode forward, eager vs captured in a HIP graph (torch.cuda.CUDAGraph over the ctypes launches)."""
import sys, time, os
if os.environ.get("VIT_DBG"):
    os.environ["CFSAR_DEV_LIB"] = "1"          # developer library: cfsar_debug_set_vit_dbg (0x1000 / 0x2000 force 128- / 256-row tiles)
import torch
sys.path.insert(0, "/root/repo")
import clip_fsar_amd.synth as synth
from clip_fsar_amd.engine import ClipFsarEngine
if os.environ.get("VIT_DBG"):
    import ctypes
    from clip_fsar_amd import hip
    L = hip.lib(); L.cfsar_debug_set_vit_dbg.argtypes = [ctypes.c_int]; L.cfsar_debug_set_vit_dbg.restype = None
    L.cfsar_debug_set_vit_dbg(int(os.environ["VIT_DBG"], 0))
a = synth.ARCHS["ViT-B/16"]
sd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict("ViT-B/16", seed=18, depth=1).items()}
tt = torch.from_numpy(synth.text_features(64, a["embed"], "train", 18)); te = torch.from_numpy(synth.text_features(24, a["embed"], "test", 18))
eng = ClipFsarEngine(a, sd, tt, te, precision=os.environ.get("PROBE_PRECISION", "bf16"), device="cuda")
ep = {k: torch.from_numpy(v).cuda() for k, v in synth.make_episode(5, 1, 1, 8, 224, 24, 0, 18).items()}
args = (ep["support_set"][None], ep["target_set"][None], ep["support_labels"][None], ep["real_support_labels"][None])
kw = dict(way=5, T=8)
def eager():
    return eng.forward(*args, **kw)
for _ in range(3): lo, cl = eager()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20): lo, cl = eager()
torch.cuda.synchronize(); t_e = (time.time() - t0) / 20
ref = lo.clone()
# host enqueue time of one forward (no sync inside): if it is close to the step time the path is launch-bound
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20):
    lo, cl = eager()
t_q = (time.time() - t0) / 20
torch.cuda.synchronize()
print("host enqueue time %.3f ms per forward" % (t_q * 1e3))
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): eager()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    glo, gcl = eager()
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20): g.replay()
torch.cuda.synchronize(); t_g = (time.time() - t0) / 20
print("eager %.3f ms/episode (%.0f eps/s)   graph %.3f ms/episode (%.0f eps/s)   max|dlogits| %.2e" % (t_e * 1e3, 1 / t_e, t_g * 1e3, 1 / t_g, float((glo - ref).abs().max())))
