import os, sys, statistics, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CFSAR_DEV_LIB"] = "1"
import torch
from clip_fsar_amd import hip
L = hip.lib()
def bench(fn, rounds=5, iters=20):
    ts = []
    for _ in range(rounds):
        fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / iters * 1e3)
    return statistics.median(ts)
for tag, M, N, K in (("qkv", 1360, 1536, 512), ("out", 1360, 512, 512), ("ff1", 1360, 2048, 512), ("ff2", 1360, 512, 2048), ("proj", 640, 512, 768)):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    r = []
    for v in (0, 1, 0, 1):
        L.cfsar_debug_set_gemm_variant(v, 0)
        r.append(bench(lambda: hip.gemm(A, W, out, bias=b)))
    L.cfsar_debug_set_gemm_variant(0, 0)
    print("%-4s M=%4d N=%4d K=%4d: skinny2 %6.1f us | fp32-MFMA v1 %6.1f us" % (tag, M, N, K, min(r[0], r[2]), min(r[1], r[3])))
