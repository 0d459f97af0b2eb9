"""Dev tool (GPU box): episodes/s when every step's frames start in pinned HOST memory (the C ABI takes device pointers; the harness
uploads fp32 frames, 48.2 MB per cfg2 episode): upload on the compute stream vs on a copy stream one step ahead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import clip_fsar_amd.synth as synth
from clip_fsar_amd.engine import ClipFsarEngine
B, steps = 16, 10
a = synth.ARCHS["ViT-B/16"]
sd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict("ViT-B/16", 18).items()}
tt = torch.from_numpy(synth.text_features(64, a["embed"], "train", 18)); te = torch.from_numpy(synth.text_features(24, a["embed"], "test", 18))
eng = ClipFsarEngine(a, sd, tt, te, precision="bf16", device="cuda", max_frames=B * 80)
eps = [synth.make_episode(5, 1, 1, 8, 224, 24, e, 18) for e in range(2)]
host = {k: torch.stack([torch.from_numpy(eps[i % 2][k]) for i in range(B)]).pin_memory() for k in ("support_set", "target_set", "support_labels", "real_support_labels")}
mb = sum(v.numel() * v.element_size() for v in host.values()) / 1e6
def fwd(d):
    return eng.forward(d["support_set"], d["target_set"], d["support_labels"], d["real_support_labels"], way=5, T=8)
dev = {k: v.cuda() for k, v in host.items()}
for _ in range(2): fwd(dev)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps): fwd(dev)
torch.cuda.synchronize(); t_res = (time.time() - t0) / steps
t0 = time.time()
for _ in range(steps):
    d = {k: v.cuda(non_blocking=True) for k, v in host.items()}
    fwd(d)
torch.cuda.synchronize(); t_ser = (time.time() - t0) / steps
cs = torch.cuda.Stream()
bufs = [{k: torch.empty_like(v, device="cuda") for k, v in host.items()} for _ in range(2)]
evs = [torch.cuda.Event() for _ in range(2)]
done = [torch.cuda.Event() for _ in range(2)]
def upload(i):
    with torch.cuda.stream(cs):
        cs.wait_event(done[i]) if i in used else None
        for k, v in host.items(): bufs[i][k].copy_(v, non_blocking=True)
        evs[i].record(cs)
used = set()
upload(0)
torch.cuda.synchronize()
t0 = time.time()
for s in range(steps):
    i = s & 1
    if s + 1 < steps: upload(1 - i)
    torch.cuda.current_stream().wait_event(evs[i])
    fwd(bufs[i])
    done[i].record(); used.add(i)
torch.cuda.synchronize(); t_ovl = (time.time() - t0) / steps
print("%.0f MB of frames per step (B = %d)" % (mb, B))
print("inputs resident in HBM         : %.2f ms/step  %.1f episodes/s" % (t_res * 1e3, B / t_res))
print("upload on the compute stream   : %.2f ms/step  %.1f episodes/s  (%.1f GB/s effective H2D)" % (t_ser * 1e3, B / t_ser, mb / 1e3 / max(t_ser - t_res, 1e-9)))
print("upload on a copy stream, 1 ahead: %.2f ms/step  %.1f episodes/s" % (t_ovl * 1e3, B / t_ovl))
