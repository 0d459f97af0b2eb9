"""Dev tool (GPU box): time the bf16 ViT attention kernel.  usage: python tools/attn_time.py F [ntok]"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
var = os.environ.get("ATTN_VARIANT")
if var:
    os.environ["CFSAR_DEV_LIB"] = "1"
import ctypes, torch
from clip_fsar_amd import hip
if var:
    L = hip.lib(); L.cfsar_debug_set_attn_variant.argtypes = [ctypes.c_int]; L.cfsar_debug_set_attn_variant.restype = None
    L.cfsar_debug_set_attn_variant(int(var))
F_ = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 197
D, H = (768, 12) if N != 257 else (1024, 16)
qkv = torch.randn(F_ * N, 3 * D, device="cuda").to(torch.bfloat16)
o = torch.empty(F_ * N, D, device="cuda", dtype=torch.bfloat16)
ts = []
for _ in range(6):
    hip.vit_attention(qkv, o, F_, N, D, H)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): hip.vit_attention(qkv, o, F_, N, D, H)
    e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 10 * 1e3)
us = statistics.median(ts)
byt = F_ * N * 4 * D * 2
print("attention bf16 F=%d ntok=%d: %.1f us  %.1f TFLOP/s  %.2f TB/s (algorithmic %d MB)" % (F_, N, us, F_ * H * 4.0 * N * N * 64 / us / 1e6, byt / us / 1e6, byt >> 20))
x = qkv.float().reshape(F_, N, 3, H, 64)[:4].permute(2, 0, 3, 1, 4)
ref = (torch.softmax(x[0] @ x[1].transpose(-1, -2) / 8.0, -1) @ x[2]).permute(0, 2, 1, 3).reshape(4 * N, D)
print("max |err| first 4 frames: %.4f" % float((o[:4 * N].float() - ref).abs().max()))
