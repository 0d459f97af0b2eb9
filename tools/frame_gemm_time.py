"""Dev tool (GPU box): cfsar_frame_gemm (the fp16 mode's per-frame GEMMs, csrc/frame_gemm.hip) against the generic cfsar_gemm dispatch on the
same operands, at one episode (80 frames) and 16 episodes (1 280 frames) per call.  usage: python tools/frame_gemm_time.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_fsar_amd import hip

def timeit(fn, iters=50):
    for _ in range(5): fn()
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    return statistics.median(ts)

for M in (80, 240, 1280):
    for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072), (1024, 4096)):
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        o1, o2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        t_new = timeit(lambda: hip.corr_gemm(A, W, o1))
        t_old = timeit(lambda: hip.gemm(A, W, o2))
        d = float((o1 - o2).abs().max())
        print("M=%4d N=%4d K=%4d : frame_gemm %6.1f us   cfsar_gemm %6.1f us   max |diff| %.2e" % (M, N, K, t_new, t_old, d), flush=True)
