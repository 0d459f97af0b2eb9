"""Dev tool (GPU box): LN-folded GEMM (fp16 MFMA) vs plain bf16 GEMM + separate LayerNorm, and the f16-vs-bf16 MFMA instruction A/B."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CFSAR_DEV_LIB"] = "1"
import ctypes, torch
from clip_fsar_amd import hip
L = hip.lib()
L.cfsar_debug_set_vit_dbg.argtypes = [ctypes.c_int]; L.cfsar_debug_set_vit_dbg.restype = None
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
M, D = 80 * 197 * B, 768
x = (torch.randn(M, D, device="cuda") * 1.5 + 0.3).to(torch.float16)
h = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
rstat = torch.empty(M, 4, device="cuda"); hip.row_stats(x, rstat, M, D)
def bench(fn, rounds=5, iters=6):
    ts = []
    for _ in range(rounds):
        fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / iters * 1e3)
    return statistics.median(ts)
print("layernorm f16->bf16: %.1f us" % bench(lambda: hip.layernorm(x, h, g, b, M, D)))
for tag, N, act in (("qkv", 2304, hip.ACT_NONE), ("fc", 3072, hip.ACT_QUICKGELU)):
    W = (torch.randn(N, D, device="cuda") * D ** -0.5)
    Wb, Wh = W.to(torch.bfloat16), W.to(torch.float16)
    c = Wh.float().sum(1); d = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    hip.layernorm(x, h, g, b, M, D)
    t_plain = bench(lambda: hip.gemm(h, Wb, out, bias=d, act=act))
    L.cfsar_debug_set_vit_dbg(0)
    t_fold = bench(lambda: hip.gemm_lnfold(x, Wh, out, c, d, rstat, act=act))
    L.cfsar_debug_set_vit_dbg(32)
    t_fold_bf = bench(lambda: hip.gemm_lnfold(x, Wh, out, c, d, rstat, act=act))
    L.cfsar_debug_set_vit_dbg(0)
    fl = 2.0 * M * N * D
    print("%-4s plain bf16 %.1f us (%.0f TF) | lnfold f16 MFMA %.1f us (%.0f TF) | lnfold, bf16 MFMA instr on the same bits %.1f us" % (
        tag, t_plain, fl / t_plain / 1e6, t_fold, fl / t_fold / 1e6, t_fold_bf))
