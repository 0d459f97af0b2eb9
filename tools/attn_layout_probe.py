"""Dev tool (GPU box): would a head-blocked qkv layout ([frame][head][token][q|k|v], 75 KB contiguous per item) speed the attention
kernel up?  The same kernel runs it today when called with D = 64, heads = 1 and F x 12 "frames"."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_fsar_amd import hip
F_, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1280, 197
def t(fn):
    ts = []
    for _ in range(6):
        fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 10 * 1e3)
    return statistics.median(ts)
qkv = torch.randn(F_ * N, 3 * 768, device="cuda").to(torch.bfloat16)
o = torch.empty(F_ * N, 768, device="cuda", dtype=torch.bfloat16)
print("row-major  [F*N, 2304], 12 heads : %.1f us" % t(lambda: hip.vit_attention(qkv, o, F_, N, 768, 12)))
qb = torch.randn(F_ * 12 * N, 192, device="cuda").to(torch.bfloat16)
ob = torch.empty(F_ * 12 * N, 64, device="cuda", dtype=torch.bfloat16)
print("head-blocked [F*12][N][192]      : %.1f us" % t(lambda: hip.vit_attention(qb, ob, F_ * 12, N, 64, 1)))
