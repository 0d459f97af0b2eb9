"""Dev tool (CPU): logits deviation of candidate 16-bit numerics schemes of the ViT tower on a golden case, before any kernel is written.

Re-runs the oracle's head on the case's episode with IEEE-half (or bf16) rounding injected where a scheme stores / feeds a 16-bit
tensor, and prints max |logits - fp32 oracle logits| per scheme.  Round 4 question: which scheme puts cfg2, cfg3
AND cfg4 inside 1e-3 (<= 7e-4 wanted)?

usage: python tools/numerics_lab.py CASE [scheme ...]        CASE e.g. cfg2_B16_5w1s_T8, cfg4_L14_5w1s_T16, t_5w1s_T8
Scheme grammar: comma-separated key=value over
  stream = f16 | f32 | hilo | hi8 residual stream storage (hilo: fp16 hi + fp16 lo, the consumer GEMMs read hi only; hi8: lo as e5m2)
  wres   = f16 | x               out_proj / c_proj weights (x = exact, i.e. a hi + lo split pair)
  wfold  = f16 | x               LN-folded QKV / c_fc weights
  wqkv, wfc, wout, wpr           one GEMM's weights (override wfold / wres)
  act    = f16 | bf16 | x        patches, qkv, probabilities, attention output, MLP hidden
  u      = (as act)              MLP hidden only, overrides act
  o      = (as act)              attention output only
  dr     = 0 | 1                 1: GEMM output rounded to fp16 before the residual add (the round-3 epilogue)
Round 6 (the per-site budget behind precision "fp16_strict", profiles/r06_strict_budget.md):
  feed   = f16 | x               what the LN-folded GEMMs read of the stream (x: both words)
  q, k, v, p                     one attention operand alone (override qkv / act)
  patch  = f16 | x               the patch embedding's pixels and weights (override act)
  pre    = same | f16            the two stores of the stream in front of the blocks (patch-embed output, ln_pre output): as `stream`, or one fp16 word
  wfold  = r, wres = m           the product's per-frame low-word correction of the weights (raw-stream form / token-mean form)
  clsx=1 | meanx=1               diagnostics: every rounding leaves the class-token row exact / gets the per-frame mean of its error added back
"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import clipfsar_oracle as orc
import _cases

torch.set_grad_enabled(False)


CLSX = [False]
MEANX = [False]
def rnd(t, kind):
    if MEANX[0] and kind in ("f16", "bf16") and t.dim() in (3, 4):
        r = t.half().float() if kind == "f16" else t.bfloat16().float()
        d = t.dim() - 2
        return r + (t - r).mean(d, keepdim=True)
    if CLSX[0] and t.dim() == 3 and kind in ("f16", "bf16"):
        r = t.half().float() if kind == "f16" else t.bfloat16().float()
        r[:, 0, :] = t[:, 0, :]
        return r
    if CLSX[0] and t.dim() == 4 and kind in ("f16", "bf16"):      # probabilities [F, h, Nq, Nk]: query row 0 exact
        r = t.half().float() if kind == "f16" else t.bfloat16().float()
        r[:, :, 0, :] = t[:, :, 0, :]
        return r
    if kind == "f16":
        return t.half().float()
    if kind == "bf16":
        return t.bfloat16().float()
    return t


def make_tower(s):
    stream, wres, wfold = s.get("stream", "f16"), s.get("wres", "f16"), s.get("wfold", "f16")
    act = s.get("act", "f16")
    CLSX[0] = s.get("clsx", "0") == "1"
    MEANX[0] = s.get("meanx", "0") == "1"
    ku, ko, kq, kp = s.get("u", act), s.get("o", act), s.get("qkv", act), s.get("p", act)
    wqkv, wfc, wout, wpr = s.get("wqkv", wfold), s.get("wfc", wfold), s.get("wout", wres), s.get("wpr", wres)
    dr = s.get("dr", "0") == "1"       # round-3 kernels: the GEMM output is rounded to fp16 BEFORE the (packed fp16) residual add

    def delta(t):
        return t.half().float() if dr else t

    def store(x):            # what the residual stream keeps
        if stream == "f16":
            return x.half().float()
        if stream == "hilo":
            hi = x.half().float()
            return hi + (x - hi).half().float()
        if stream == "hi8":          # second word kept as e5m2 (the high byte of the fp16 low word, rounded): ~14 bits instead of ~22
            hi = x.half().float()
            return hi + (x - hi).half().to(torch.float8_e5m2).float()
        return x

    fk = s.get("feed", "f16")
    def feed(x):             # what a consumer GEMM reads of the stream (16-bit operand)
        return rnd(x, fk)

    def mm(A, W32, kind):
        """A [F, N, K] @ W^T with the weight precision `kind`: f16 | x (exact) | m (fp16 word + the low word applied to the per-frame
        TOKEN MEAN of the operand only: a [F, K] x [K, N_out] GEMM, the coherent part of the weight-rounding error)"""
        if kind == "m":
            Wh = W32.half().float()
            return A @ Wh.t() + (A.mean(1, keepdim=True) @ (W32 - Wh).half().float().t())
        return A @ rnd(W32, kind).t()

    def mm_ln(xi, mu, var, Wg32, kind):
        """LN-folded GEMM.  kind 'r': low word applied to the per-frame mean of the RAW stream (what the producer GEMM can emit as column sums),
        the row mean's share handled exactly:  xhat W_lo ~ (mean_t(x) W_lo - mu_t c_lo) / sigma_t"""
        sd_ = torch.sqrt(var + 1e-5)
        if kind == "r":
            Wh = Wg32.half().float()
            Wl = (Wg32 - Wh).half().float()
            return ((xi - mu) / sd_) @ Wh.t() + (xi.mean(1, keepdim=True) @ Wl.t() - mu * Wl.sum(1)) / sd_
        return mm((xi - mu) / sd_, Wg32, kind)

    def tower(frames, sd, arch, prefix="backbone.", chunk=40):
        outs = []
        g = lambda n: sd[prefix + n]
        D, heads = arch["width"], arch["heads"]
        hd = D // heads
        for s0 in range(0, frames.shape[0], chunk):
            f = frames[s0:s0 + chunk]
            F_, C, H, Wd = f.shape
            p_ = arch["patch"]
            gg = H // p_
            pt = f.reshape(F_, C, gg, p_, gg, p_).permute(0, 2, 4, 1, 3, 5).reshape(F_, gg * gg, C * p_ * p_)
            tok = rnd(pt, s.get("patch", act)) @ rnd(g("conv1.weight").reshape(D, -1), s.get("patch", act)).t()
            x = torch.cat([g("class_embedding").reshape(1, 1, D).expand(F_, 1, D), tok], 1) + g("positional_embedding")
            pre = s.get("pre", "same")
            st0 = store if pre == "same" else (lambda t: rnd(t, pre))
            x = st0(x)
            x = st0(orc.layer_norm(x, g("ln_pre.weight"), g("ln_pre.bias")))
            for i in range(arch["layers"]):
                b = "transformer.resblocks.%d." % i
                N = x.shape[1]
                xi = feed(x)
                gam, bet = g(b + "ln_1.weight"), g(b + "ln_1.bias")
                mu, var = xi.mean(-1, keepdim=True), xi.var(-1, unbiased=False, keepdim=True)
                qkv = mm_ln(xi, mu, var, g(b + "attn.in_proj_weight") * gam[None, :], wqkv) + (g(b + "attn.in_proj_weight") @ bet + g(b + "attn.in_proj_bias"))
                q, k, v = qkv.split(D, -1)
                q, k, v = rnd(q, s.get("q", kq)), rnd(k, s.get("k", kq)), rnd(v, s.get("v", kq))
                q = q.reshape(F_, N, heads, hd).transpose(1, 2); k = k.reshape(F_, N, heads, hd).transpose(1, 2)
                v = v.reshape(F_, N, heads, hd).transpose(1, 2)
                sc = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
                e = torch.exp(sc - sc.max(-1, keepdim=True).values)
                o = (rnd(e, kp) @ v) / e.sum(-1, keepdim=True)
                o = rnd(o.transpose(1, 2).reshape(F_, N, D), ko)
                x = store(x + delta(mm(o, g(b + "attn.out_proj.weight"), wout) + g(b + "attn.out_proj.bias")))
                xi = feed(x)
                gam, bet = g(b + "ln_2.weight"), g(b + "ln_2.bias")
                mu, var = xi.mean(-1, keepdim=True), xi.var(-1, unbiased=False, keepdim=True)
                u = mm_ln(xi, mu, var, g(b + "mlp.c_fc.weight") * gam[None, :], wfc) + (g(b + "mlp.c_fc.weight") @ bet + g(b + "mlp.c_fc.bias"))
                u = rnd(orc.quick_gelu(u), ku)
                x = store(x + delta(mm(u, g(b + "mlp.c_proj.weight"), wpr) + g(b + "mlp.c_proj.bias")))
            c = orc.layer_norm(x[:, 0, :], g("ln_post.weight"), g("ln_post.bias"))
            outs.append(c @ g("proj"))
        return torch.cat(outs, 0)
    return tower


def run(case, schemes):
    gold = _cases.load_golden(case)
    meta = gold["meta"]
    a, sd, tt, te, ep = _cases.case_inputs(meta)
    kw = dict(frames=meta["T"], merge_before=meta.get("merge_before", False), single_direct=meta.get("single_direct", False),
              depth=meta.get("depth", 1))
    ref = torch.from_numpy(gold["logits"]).float().reshape(-1)
    orig = orc.vit_forward
    print("%s: logits spread %.3f" % (case, float(ref.max() - ref.min())), flush=True)
    for sch in schemes:
        s = dict(kv.split("=") for kv in sch.split(",") if kv)
        orc.vit_forward = make_tower(s)
        t0 = time.time()
        taps = {}
        out = orc.head_forward(ep, sd, tt, te, a, taps=taps, **kw)
        orc.vit_forward = orig
        lg = out["logits"].reshape(-1)
        print("  %-46s max |dlogits| %.2e  rms %.2e  (%.0f s)" % (sch, float((lg - ref).abs().max()), float((lg - ref).pow(2).mean().sqrt()),
                                                                  time.time() - t0), flush=True)


def run_feats(arch_name, nf, schemes, seed=18):
    """feature error of each scheme on the first nf frames of episode 0 (thousands of values: a stable statistic, unlike max |dlogits| of 5)"""
    import clip_fsar_amd.synth as synth
    a = synth.ARCHS[arch_name]
    sd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict(arch_name, seed).items()}
    ep = synth.make_episode(5, 1, 1, 8, a["res"], 24, 0, seed)
    frames = torch.cat([torch.from_numpy(ep["support_set"]), torch.from_numpy(ep["target_set"])])[:nf]
    ref = orc.vit_forward(frames, sd, a)
    print("%s, %d frames: feature rms %.4f" % (arch_name, nf, float(ref.pow(2).mean().sqrt())), flush=True)
    for sch in schemes:
        s = dict(kv.split("=") for kv in sch.split(",") if kv)
        f = make_tower(s)(frames, sd, a)
        d = f - ref
        # coherent part: the error of the MEAN feature over the frames (what survives the temporal / prototype averaging)
        print("  %-46s rms %.2e  max %.2e  rms of frame-mean error %.2e" % (sch, float(d.pow(2).mean().sqrt()), float(d.abs().max()),
                                                                          float(d.mean(0).pow(2).mean().sqrt())), flush=True)


def run_episode(arch_name, shot, q, T, schemes, seed=18, episode=3):
    """logits error of each scheme on a FRESH synthetic episode with q queries per class (5 q x 5 logits: a steadier statistic than the
    goldens' 5 ... 25), against the oracle's fp32 logits of the same episode"""
    import clip_fsar_amd.synth as synth
    a = synth.ARCHS[arch_name]
    sd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict(arch_name, seed).items()}
    tt = torch.from_numpy(synth.text_features(64, a["embed"], "train", seed))
    te = torch.from_numpy(synth.text_features(24, a["embed"], "test", seed))
    ep = {k: torch.from_numpy(v) for k, v in synth.make_episode(5, shot, q, T, a["res"], 24, episode, seed).items()}
    orig = orc.vit_forward
    ref = orc.head_forward(ep, sd, tt, te, a, frames=T)["logits"].reshape(-1)
    print("%s %d-shot q=%d T=%d: %d logits, spread %.3f" % (arch_name, shot, q, T, ref.numel(), float(ref.max() - ref.min())), flush=True)
    for sch in schemes:
        s = dict(kv.split("=") for kv in sch.split(",") if kv)
        orc.vit_forward = make_tower(s)
        lg = orc.head_forward(ep, sd, tt, te, a, frames=T)["logits"].reshape(-1)
        orc.vit_forward = orig
        print("  %-46s max |dlogits| %.2e  rms %.2e" % (sch, float((lg - ref).abs().max()), float((lg - ref).pow(2).mean().sqrt())), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "episode":        # episode ARCH SHOT Q T scheme...
        run_episode(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6:])
        sys.exit(0)
    if sys.argv[1] == "feats":
        run_feats(sys.argv[2], int(sys.argv[3]), sys.argv[4:])
        sys.exit(0)
    case = sys.argv[1]
    schemes = sys.argv[2:] or ["stream=f16", "stream=f32", "stream=hilo", "stream=f32,wres=x", "stream=f32,wres=x,wfold=x",
                               "stream=f32,wfold=x", "stream=f16,wres=x,wfold=x", "stream=f32,act=x", "stream=f32,u=x", "stream=f32,o=x"]
    run(case, schemes)
