"""Dev tool: run ONE bf16 GEMM shape repeatedly (for rocprofv3 --pmc runs).
usage: python tools/gemm_only.py M N K [plain|gelu|res16|res32]     env AB_VARIANT=v[:dbg] forces a kernel variant (dev library)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
var = os.environ.get("AB_VARIANT")
if var:
    os.environ["CFSAR_DEV_LIB"] = "1"
import torch
from clip_fsar_amd import hip
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = sys.argv[4] if len(sys.argv) > 4 else "plain"
if mode == "f32":
    mode = "res32"
if var:
    v = [int(x) for x in (var.split(":") + ["0"])[:2]]
    hip.lib().cfsar_debug_set_gemm_variant(v[0], v[1])
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
od = {"plain": torch.bfloat16, "gelu": torch.bfloat16, "res16": torch.float16, "res32": torch.float32}[mode]
out = torch.zeros(M, N, device="cuda", dtype=od)
for _ in range(5):
    hip.gemm(A, W, out, bias=bias, residual=out if mode.startswith("res") else None,
             act=hip.ACT_QUICKGELU if mode == "gelu" else hip.ACT_NONE)
torch.cuda.synchronize()
