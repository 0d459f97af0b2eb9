"""Dev tool: run ONE bf16 GEMM shape repeatedly (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_fsar_amd import hip
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
outf32 = len(sys.argv) > 4 and sys.argv[4] == "f32"
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.float32 if outf32 else torch.bfloat16)
for _ in range(5):
    hip.gemm(A, W, out, bias=bias, residual=out if outf32 else None)
torch.cuda.synchronize()
