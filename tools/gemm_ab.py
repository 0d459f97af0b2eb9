"""Dev tool (GPU box): in-process A/B of GEMM variants on the ViT shapes.  Configs are interleaved round-robin so that clock /
thermal drift hits all of them equally; reports the median and min over rounds.
usage: python tools/gemm_ab.py B "variant:dbg" "variant:dbg" ...      (B = episodes -> M = 80*197*B)"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CFSAR_DEV_LIB"] = "1"      # the -DCFSAR_DEV library (clip-fsar_amd/build.py --dev) carries the variant hook
import torch
from clip_fsar_amd import hip

B = int(sys.argv[1])
cfgs = [tuple(int(x) for x in c.split(":")) for c in sys.argv[2:]]
L = hip.lib()
D = int(os.environ.get("AB_D", "768"))                      # 1024 + AB_TOK=257 AB_FPE=160: ViT-L/14, 16 frames
M = int(os.environ.get("AB_FPE", "80")) * int(os.environ.get("AB_TOK", "197")) * B
dev = "cuda"
shapes = [("qkv", 3 * D, D), ("out", D, D), ("fc", 4 * D, D), ("proj", D, 4 * D)]
if os.environ.get("AB_SHAPES"):
    shapes = [s for s in shapes if s[0] in os.environ["AB_SHAPES"].split(",")]
ROUNDS, ITERS = 7, 8
for tag, n, k in shapes:
    A = torch.randn(M, k, device=dev).to(torch.bfloat16)
    W = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    sd = torch.float16 if os.environ.get("AB_STREAM") == "fp16" else torch.float32      # residual-stream dtype
    out = torch.empty(M, n, device=dev, dtype=torch.bfloat16 if tag in ("qkv", "fc") else sd)
    res = out if tag in ("out", "proj") else None
    act = hip.ACT_QUICKGELU if tag == "fc" else hip.ACT_NONE
    times = {c: [] for c in cfgs}
    for rnd in range(ROUNDS + 1):
        for c in cfgs:
            L.cfsar_debug_set_gemm_variant(c[0], c[1])
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            hip.gemm(A, W, out, bias=bias, residual=res, act=act)
            s.record()
            for _ in range(ITERS):
                hip.gemm(A, W, out, bias=bias, residual=res, act=act)
            e.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[c].append(s.elapsed_time(e) / ITERS * 1e3)
    for c in cfgs:
        med, mn = statistics.median(times[c]), min(times[c])
        print("%-4s M=%d N=%d K=%d  variant %2d dbg %3d : median %7.1f us (%6.1f TF)  min %7.1f us (%6.1f TF)" % (
            tag, M, n, k, c[0], c[1], med, 2.0 * M * n * k / med / 1e6, mn, 2.0 * M * n * k / mn / 1e6))
L.cfsar_debug_set_gemm_variant(0, 0)
