"""Dev tool: error map of cfsar_gemm_lnfold on one shape (per 256x256 tile max |err|)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
paths = os.environ.get("VIT_PATHS")
if paths:
    os.environ["CFSAR_DEV_LIB"] = "1"
import torch, torch.nn.functional as F
from clip_fsar_amd import hip
if paths:
    hip.lib().cfsar_debug_set_vit_paths(*[int(v) for v in paths.split(":")])
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
act = sys.argv[4] if len(sys.argv) > 4 else "none"
g = torch.Generator().manual_seed(3)
x = (torch.randn(M, K, generator=g) * (0.5 + torch.rand(M, 1, generator=g) * 3.0) + torch.randn(M, 1, generator=g) * 2.0)
x[:, 5] += 25.0
x = x.to(torch.float16).cuda()
W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
gamma = (1.0 + 0.5 * torch.randn(K, generator=g)).cuda(); beta = (0.3 * torch.randn(K, generator=g)).cuda(); bias = torch.randn(N, generator=g).cuda()
ref = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ W.t() + bias
if act == "gelu":
    ref = ref * torch.sigmoid(1.702 * ref)
Wg = (W * gamma[None, :]).to(torch.float16).contiguous()
c = Wg.double().sum(1).float(); d = (W.double() @ beta.double() + bias.double()).float()
rstat = torch.empty(M, 4, device="cuda")
hip.row_stats(x, rstat, M, K)
out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
hip.gemm_lnfold(x, Wg, out, c, d, rstat, act=hip.ACT_QUICKGELU if act == "gelu" else hip.ACT_NONE)
torch.cuda.synchronize()
err = (out.float() - ref).abs()
err = torch.nan_to_num(err, nan=1e9)
tm, tn = (M + 255) // 256, (N + 255) // 256
pad = torch.zeros(tm * 256, tn * 256, device="cuda"); pad[:M, :N] = err
tile = pad.reshape(tm, 256, tn, 256).amax((1, 3)).cpu()
bad = (tile > 0.2).nonzero()
print("max err", float(err.max()), "bad tiles", len(bad), "of", tm * tn)
print("bad (band, col) sample:", bad[:40].tolist())
if len(bad):
    b, cidx = bad[0].tolist()
    sub = pad[b * 256:(b + 1) * 256, cidx * 256:(cidx + 1) * 256]
    rows = (sub.amax(1) > 0.2).nonzero().flatten().tolist(); cols = (sub.amax(0) > 0.2).nonzero().flatten().tolist()
    print("tile", b, cidx, "bad rows", rows[:20], "... n=", len(rows), "bad cols", cols[:20], "... n=", len(cols))

    pre = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ W.t() + bias
    xs = x.float(); mean = xs.mean(1); var = xs.var(1, unbiased=False); rstd = torch.rsqrt(var + 1e-5)
    raw = xs @ Wg.float().t()                      # x W'^T
    for r in rows[:6]:
        gr, gc = b * 256 + r, cidx * 256 + cols[0]
        o = float(out[gr, gc]); pv = float(pre[gr, gc]); rf = float(ref[gr, gc])
        a = float(raw[gr, gc] - mean[gr] * c[gc] + d[gc] / rstd[gr])      # the accumulator the kernel should hold
        import math
        ge = lambda t: t / (1 + math.exp(-1.702 * t))
        print("row %d col %d: out %.4f ref %.4f | pre %.4f rstd %.4f acc %.4f | gelu(acc) %.4f gelu(pre*rstd) %.4f gelu(d) %.4f gelu(raw*rstd) %.4f" % (
            gr, gc, o, rf, pv, float(rstd[gr]), a, ge(a), ge(pv * float(rstd[gr])), ge(float(d[gc])), ge(float(raw[gr, gc]) * float(rstd[gr]))))
        # neighbours in the same row (other columns of the 8-group)
        print("   same row cols %d..%d out" % (gc - gc % 8, gc - gc % 8 + 7), [round(float(v), 3) for v in out[gr, gc - gc % 8: gc - gc % 8 + 8]], "ref", [round(float(v), 3) for v in ref[gr, gc - gc % 8: gc - gc % 8 + 8]])
