"""Dev tool (GPU box): per-phase cycle stamps of the one-head bf16 attention kernel (wave 0 and wave 7 of every workgroup)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_fsar_amd import hip
F_ = int(sys.argv[1]) if len(sys.argv) > 1 else 640
N, D, H = 197, 768, 12
qkv = torch.randn(F_ * N, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(F_ * N, D, device="cuda", dtype=torch.bfloat16)
L = hip.lib()
L.cfsar_debug_set_attn_trace.argtypes = [ctypes.c_void_p]
for _ in range(3):
    hip.vit_attention(qkv, o, F_, N, D, H)
torch.cuda.synchronize()
tr = torch.zeros(F_ * H * 8 * 16, dtype=torch.int64, device="cuda")
L.cfsar_debug_set_attn_trace(ctypes.c_void_p(tr.data_ptr()))
hip.vit_attention(qkv, o, F_, N, D, H)
torch.cuda.synchronize()
L.cfsar_debug_set_attn_trace(None)
t = tr.cpu().reshape(F_ * H, 8, 16).double()
names = ["staging loads + LDS writes (0->1)", "barrier wait (1->2)", "Q fragment load (2->3)", "S = K.Q^T 26 MFMA (3->4)",
         "softmax (4->5)", "PV 28 MFMA (5->6)", "store issue (6->7)", "second tile + tail (7->8)"]
for w in (0, 7):
    x = t[:, w, :]
    print("wave %d: workgroup lifetime mean %.0f cycles" % (w, (x[:, 8] - x[:, 0]).mean()))
    for i, n in enumerate(names):
        d = x[:, i + 1] - x[:, i]
        print("   %-36s mean %8.0f  median %8.0f  p90 %8.0f" % (n, d.mean(), d.median(), d.quantile(0.9)))
span = t[:, :, 8].max() - t[:, :, 0][t[:, :, 0] > 0].min()
print("kernel span %.0f cycles; workgroups %d -> %.1f per CU" % (span, F_ * H, F_ * H / 256.0))
