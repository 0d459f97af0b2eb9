"""Dev tool (GPU): what the fused LayerNorm statistics cost the residual GEMMs' epilogues: cfsar_gemm_residual_stats (bf16 mode) and
cfsar_gemm_residual_wide (fp16 mode) with and without stats_partial, at the bench's M, interleaved."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_fsar_amd import hip
M, D = 252160, 768
for tag, K in (("out_proj", 768), ("c_proj", 3072)):
    for mode in ("bf16", "fp16"):
        td = torch.bfloat16 if mode == "bf16" else torch.float16
        A = torch.randn(M, K, device="cuda").to(td)
        W = (torch.randn(D, K, device="cuda") * K ** -0.5).to(td)
        bias = torch.randn(D, device="cuda")
        x = torch.randn(M, D, device="cuda").to(torch.float16)
        xl = torch.zeros(M, D, device="cuda", dtype=torch.float16)
        part = torch.empty(M, D // 64, 2, device="cuda")
        def call(p):
            if mode == "bf16":
                hip.gemm_residual_stats(A, W, x, bias, p)
            else:
                hip.gemm_residual_wide(A, W, x, xl, bias, p)
        ts = {True: [], False: []}
        for rnd in range(6):
            for with_stats in (True, False):
                p = part if with_stats else None
                call(p)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(6):
                    call(p)
                e.record(); torch.cuda.synchronize()
                if rnd: ts[with_stats].append(s.elapsed_time(e) / 6 * 1e3)
        print("%-8s %s: with statistics %7.1f us, without %7.1f us" % (tag, mode, statistics.median(ts[True]), statistics.median(ts[False])), flush=True)
