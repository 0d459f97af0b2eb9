"""Dev tool: is the GEMM clock/power limited?  Same kernel on zero, small-integer and random operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_fsar_amd import hip
from perf_probe import timeit
M, N, K = 63040, 2304, 768
for name, gen in (("zeros", lambda *s: torch.zeros(*s, device="cuda")),
                  ("uniform[-1,1)", lambda *s: torch.rand(*s, device="cuda") * 2 - 1),
                  ("randn", lambda *s: torch.randn(*s, device="cuda")),
                  ("randn*0.02", lambda *s: torch.randn(*s, device="cuda") * 0.02)):
    A = gen(M, K).to(torch.bfloat16); W = gen(N, K).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    us = timeit(lambda: hip.gemm(A, W, out, bias=bias), iters=30)
    print("%-14s %.1f us  %.1f TFLOP/s" % (name, us, 2.0 * M * N * K / us / 1e6))
