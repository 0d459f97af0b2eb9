"""Dev tool: run the bf16 ViT attention kernel repeatedly (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_fsar_amd import hip
F_ = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 197
D, H = 768, 12
qkv = torch.randn(F_ * N, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(F_ * N, D, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    hip.vit_attention(qkv, o, F_, N, D, H)
torch.cuda.synchronize()
