"""Dev tool (GPU box): A/B of the GEMM variants on the RN50 tower's 1x1-conv / im2col GEMM shapes (8 episodes = 640 frames)."""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CFSAR_DEV_LIB"] = "1"      # the -DCFSAR_DEV library (clip-fsar_amd/build.py --dev) carries the variant hook
import torch
from clip_fsar_amd import hip

variants = [tuple(int(x) for x in (v.split(':') + ['0'])[:2]) for v in sys.argv[1:]] or [(1, 0), (2, 0), (6, 0), (10, 0)]
L = hip.lib()
F = 640
# (tag, M, N, K, residual)
shapes = [("stem1 im2col", F * 112 * 112, 32, 64, False), ("stem2 im2col", F * 112 * 112, 32, 320, False),
          ("stem3 im2col", F * 112 * 112, 64, 320, False),
          ("l1.c1", F * 56 * 56, 64, 256, False), ("l1.c2 im2col", F * 56 * 56, 64, 576, False), ("l1.c3+res", F * 56 * 56, 256, 64, True),
          ("l2.c1", F * 28 * 28, 128, 512, False), ("l2.c2 im2col", F * 28 * 28, 128, 1152, False), ("l2.c3+res", F * 28 * 28, 512, 128, True),
          ("l3.c1", F * 14 * 14, 256, 1024, False), ("l3.c2 im2col", F * 14 * 14, 256, 2304, False), ("l3.c3+res", F * 14 * 14, 1024, 256, True),
          ("l4.c1", F * 7 * 7, 512, 2048, False), ("l4.c2 im2col", F * 7 * 7, 512, 4608, False), ("l4.c3+res", F * 7 * 7, 2048, 512, True)]
for tag, M, N, K, res in shapes:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if res else None
    line = "%-14s M=%8d N=%4d K=%4d " % (tag, M, N, K)
    for v in variants:
        L.cfsar_debug_set_gemm_variant(v[0], v[1])
        ts = []
        for rnd in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            hip.gemm(A, W, out, bias=bias, residual=r, relu=True)
            s.record()
            for _ in range(4):
                hip.gemm(A, W, out, bias=bias, residual=r, relu=True)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 4 * 1e3)
        med = statistics.median(ts)
        byts = M * K * 2 + M * N * 2 * (2 if res else 1)
        line += "| v%-2d:%-3d %7.1f us %5.0f TF %4.1f TB/s " % (v[0], v[1], med, 2.0 * M * N * K / med / 1e6, byts / med / 1e6)
    print(line, flush=True)
    del A, out, r
L.cfsar_debug_set_gemm_variant(0, 0)
