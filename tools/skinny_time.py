"""Dev tool (GPU box): the fp32 skinny GEMMs of the one-episode step (temporal head M = 85, final ViT projection M = 40)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_fsar_amd import hip
def bench(fn, rounds=5, iters=20):
    ts = []
    for _ in range(rounds):
        fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / iters * 1e3)
    return statistics.median(ts)
for tag, M, N, K in (("qkv", 85, 1536, 512), ("out", 85, 512, 512), ("ff1", 85, 2048, 512), ("ff2", 85, 512, 2048), ("proj", 40, 512, 768)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    t = bench(lambda: hip.gemm(A, W, out, bias=b))
    print("%-4s M=%3d N=%4d K=%4d: %6.1f us  (W %.1f MB -> %.2f TB/s)" % (tag, M, N, K, t, N * K * 4 / 1e6, N * K * 4 / t / 1e6))
print("-- 16 episodes (1 360 rows)")
for tag, M, N, K in (("qkv", 1360, 1536, 512), ("out", 1360, 512, 512), ("ff1", 1360, 2048, 512), ("ff2", 1360, 512, 2048), ("proj", 640, 512, 768)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    t = bench(lambda: hip.gemm(A, W, out, bias=b))
    ref = A.double() @ W.double().t() + b
    print("%-4s M=%4d N=%4d K=%4d: %6.1f us   max err %.2e" % (tag, M, N, K, t, float((out.double() - ref).abs().max())))
