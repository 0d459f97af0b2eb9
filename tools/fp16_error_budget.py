"""Dev tool (CPU): where the fp16 numerics mode's deviation comes from.  Re-runs the oracle's ViT on a few frames with IEEE-half rounding
injected at chosen points (the points where the HIP path stores a 16-bit tensor) and reports the feature error of each subset.
usage: python tools/fp16_error_budget.py [arch=ViT-B/16] [frames=4]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import clipfsar_oracle as orc
import clip_fsar_amd.synth as synth

arch_name = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/16"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
a = synth.ARCHS[arch_name]
sd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict(arch_name, 18).items()}
ep = synth.make_episode(5, 1, 1, 8, a["res"], 24, 0, 18)
frames = torch.from_numpy(ep["support_set"])[:nf]
P = "backbone."


def r16(t, on):
    return t.half().float() if on else t


def vit(points):
    """points: set of names in {stream, w, patches, qkv, p, o, u}"""
    g = lambda n: sd[P + n]
    W = lambda n: r16(g(n), "w" in points)
    F_, C, H, Wd = frames.shape
    p_, D = a["patch"], a["width"]
    gg = H // p_
    pt = frames.reshape(F_, C, gg, p_, gg, p_).permute(0, 2, 4, 1, 3, 5).reshape(F_, gg * gg, C * p_ * p_)
    tok = r16(pt, "patches" in points) @ W("conv1.weight").reshape(D, -1).t()
    x = torch.cat([g("class_embedding").reshape(1, 1, D).expand(F_, 1, D), tok], 1) + g("positional_embedding")
    x = r16(x, "stream" in points)
    x = r16(orc.layer_norm(x, g("ln_pre.weight"), g("ln_pre.bias")), "stream" in points)
    heads = a["heads"]; hd = D // heads
    for i in range(a["layers"]):
        b = "transformer.resblocks.%d." % i
        N = x.shape[1]
        # LN folded: x (fp16 stream) x Wg (fp16), statistics in fp32 -> equivalent to LN in fp32 on the rounded stream with rounded W*gamma
        gam, bet = g(b + "ln_1.weight"), g(b + "ln_1.bias")
        mu, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
        Wg = r16(g(b + "attn.in_proj_weight") * gam[None, :], "w" in points)
        qkv = ((x - mu) / torch.sqrt(var + 1e-5)) @ Wg.t() + (g(b + "attn.in_proj_weight") @ bet + g(b + "attn.in_proj_bias"))
        qkv = r16(qkv, "qkv" in points)
        q, k, v = qkv.split(D, -1)
        q = q.reshape(F_, N, heads, hd).transpose(1, 2); k = k.reshape(F_, N, heads, hd).transpose(1, 2); v = v.reshape(F_, N, heads, hd).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        e = torch.exp(s - s.max(-1, keepdim=True).values)
        o = (r16(e, "p" in points) @ v) / e.sum(-1, keepdim=True)
        o = r16(o.transpose(1, 2).reshape(F_, N, D), "o" in points)
        x = r16(x + o @ W(b + "attn.out_proj.weight").t() + g(b + "attn.out_proj.bias"), "stream" in points)
        gam, bet = g(b + "ln_2.weight"), g(b + "ln_2.bias")
        mu, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
        Wg = r16(g(b + "mlp.c_fc.weight") * gam[None, :], "w" in points)
        u = ((x - mu) / torch.sqrt(var + 1e-5)) @ Wg.t() + (g(b + "mlp.c_fc.weight") @ bet + g(b + "mlp.c_fc.bias"))
        u = r16(orc.quick_gelu(u), "u" in points)
        x = r16(x + u @ W(b + "mlp.c_proj.weight").t() + g(b + "mlp.c_proj.bias"), "stream" in points)
    c = orc.layer_norm(x[:, 0, :], g("ln_post.weight"), g("ln_post.bias"))
    return c @ g("proj")


with torch.no_grad():
    ref = vit(set())
    allp = {"stream", "w", "patches", "qkv", "p", "o", "u"}
    rms = float(ref.pow(2).mean().sqrt())
    print("%s, %d frames: feature rms %.3f" % (arch_name, nf, rms))
    for name, pts in [("all", allp)] + [("only " + p, {p}) for p in sorted(allp)] + [("all but " + p, allp - {p}) for p in sorted(allp)]:
        f = vit(pts)
        print("  %-18s max |dfeat| %.2e   rms %.2e" % (name, float((f - ref).abs().max()), float((f - ref).pow(2).mean().sqrt())))
