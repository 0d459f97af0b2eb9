"""Dev tool: per-tile phase timing of the p6 GEMM via in-kernel cycle counter stamps."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_fsar_amd import hip
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
outf32 = len(sys.argv) > 4 and sys.argv[4] == "f32"
act = hip.ACT_QUICKGELU if (len(sys.argv) > 4 and sys.argv[4] == "gelu") else hip.ACT_NONE
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.float32 if outf32 else torch.bfloat16)
ntiles = ((M + 255) // 256) * ((N + 255) // 256)
trace = torch.zeros(ntiles * 8, dtype=torch.int64, device="cuda")
L = hip.lib()
for _ in range(3):
    hip.gemm(A, W, out, bias=bias, residual=out if outf32 else None, act=act)
torch.cuda.synchronize()
L.cfsar_debug_set_gemm_trace.argtypes = [ctypes.c_void_p]
L.cfsar_debug_set_gemm_trace(ctypes.c_void_p(trace.data_ptr()))
hip.gemm(A, W, out, bias=bias, residual=out if outf32 else None, act=act)
torch.cuda.synchronize()
L.cfsar_debug_set_gemm_trace(None)
t = trace.cpu().reshape(ntiles, 8).double()
t0 = t[:, 0].min()
names = ["prologue(DMA fill)", "main loop", "epilogue half0", "epilogue half1", "store drain"]
d = [t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4]]
tot = t[:, 5] - t[:, 0]
print("tiles %d; s_memtime/readcyclecounter ticks (100 MHz const clock? or shader clock) -- ratios matter" % ntiles)
for n, x in zip(names, d):
    print("  %-20s mean %9.0f  median %9.0f  (%.1f%% of tile)" % (n, x.mean(), x.median(), 100 * x.mean() / tot.mean()))
print("  %-20s mean %9.0f" % ("tile total", tot.mean()))
print("  kernel span (first start -> last end): %.0f ticks; sum(tile)/256 CUs = %.0f" % (t[:, 5].max() - t0, tot.sum() / 256))
# gaps between consecutive tiles on the same CU are unknown (CU ids not recorded); estimate from span
