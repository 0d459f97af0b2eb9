"""Dev tool (GPU box): is the one-episode step bound by the host's launch path?  Enqueue time (no sync) vs step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from clip_fsar_amd import synth
from clip_fsar_amd.engine import ClipFsarEngine
import numpy as np
cfg = B.CONFIGS["cfg2"]
a = synth.ARCHS[cfg["arch"]]
sd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict(cfg["arch"], seed=18).items()}
tt = torch.from_numpy(synth.text_features(64, a["embed"], "train", 18)); te = torch.from_numpy(synth.text_features(24, a["embed"], "test", 18))
eng = ClipFsarEngine(a, sd, tt, te, precision="bf16")
ep = synth.make_episode(way=5, shot=1, query_per_class=1, frames=8, res=a["res"], n_test_classes=24, episode=0, seed=18)
ep = {k: torch.from_numpy(v).cuda() for k, v in ep.items()}
def step():
    return eng.forward(ep["support_set"], ep["target_set"], ep["support_labels"], ep["real_support_labels"], 5, 8)
for _ in range(10): step()
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.3f ms per step, total %.3f ms per step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
# graph capture
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): out = step()
torch.cuda.current_stream().wait_stream(s)
try:
    with torch.cuda.graph(g):
        out = step()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("graph replay %.3f ms per step" % ((t1 - t0) / n * 1e3))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
