"""Dev tool (CPU): logits deviation of candidate 16-bit numerics schemes of the CLIP RN50 tower (the ViT tower's counterpart is
tools/numerics_lab.py).  BatchNorm folded as the engine folds it; rounding injected where a scheme stores / feeds a 16-bit tensor.

usage: python tools/numerics_lab_rn.py CASE [scheme ...]                   CASE: rn50_5w1s_T2, rn_t_5w2s_T4
       python tools/numerics_lab_rn.py episode ARCH SHOT Q T [scheme ...]  fresh synthetic episode (steadier statistic)
Scheme grammar: comma-separated key=value over
  act  = f16 | bf16 | x     every stored activation (conv outputs after bias / ReLU, pools, block outputs)
  w    = f16 | bf16 | x | m what the convs' folded weights are rounded to (m: fp16 word + low word applied to the per-frame pixel mean)
  w3, w1, wd                3x3 convs / 1x1 convs / downsample convs only
  res  = (as act) | x       block output (the identity stream) only
  pool = (as act)           attention-pool tokens / k / v operands
"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as Fn
import clipfsar_oracle as orc
import _cases

torch.set_grad_enabled(False)


def rnd(t, kind):
    if kind == "f16":
        return t.half().float()
    if kind == "bf16":
        return t.bfloat16().float()
    return t


def make_tower(s):
    act = s.get("act", "f16")
    w = s.get("w", "f16")
    w3, w1, wd = s.get("w3", w), s.get("w1", w), s.get("wd", w)
    res = s.get("res", act)
    pool = s.get("pool", act)
    dr = s.get("dr", "0") == "1"          # conv3 output rounded to fp16 before a packed fp16 identity add (the ViT-block kernel's fp16 residual form)

    def conv(x, sd, cv, bn, kind, padding=0, stride=1, relu=True, out=None):
        g = lambda n: sd[n]
        sc = g(bn + ".weight") / torch.sqrt(g(bn + ".running_var") + 1e-5)
        b = g(bn + ".bias") - g(bn + ".running_mean") * sc
        W = g(cv + ".weight") * sc.reshape(-1, 1, 1, 1)
        if kind == "m":
            Wh = W.half().float()
            Wl = (W - Wh).half().float()
            y = Fn.conv2d(x, Wh, None, stride=stride, padding=padding)
            # low word on the per-frame mean pixel: every tap sees (almost) the same mean
            y = y + (x.mean((2, 3)) @ Wl.sum((2, 3)).t())[:, :, None, None]
        else:
            y = Fn.conv2d(x, rnd(W, kind), None, stride=stride, padding=padding)
        y = y + b.reshape(1, -1, 1, 1)
        if out is not None:
            return y
        return rnd(torch.relu(y) if relu else y, act)

    def tower(frames, sd, arch, prefix="backbone.", chunk=40):
        outs = []
        width, layers, heads = arch["width"], arch["layers"], arch["heads"]
        P = prefix
        for s0 in range(0, frames.shape[0], chunk):
            x = frames[s0:s0 + chunk]
            x = conv(x, sd, P + "conv1", P + "bn1", "x", padding=1, stride=2)       # fp32 VALU conv from the fp32 frames
            x = conv(x, sd, P + "conv2", P + "bn2", w3, padding=1)
            x = conv(x, sd, P + "conv3", P + "bn3", w3, padding=1)
            x = rnd(Fn.avg_pool2d(x, 2), act)
            inplanes = width
            for li, (planes, blocks) in enumerate(zip((width, width * 2, width * 4, width * 8), layers), start=1):
                for bi in range(blocks):
                    stride = 2 if (li > 1 and bi == 0) else 1
                    b = "%slayer%d.%d." % (P, li, bi)
                    o = conv(x, sd, b + "conv1", b + "bn1", w1)
                    o = conv(o, sd, b + "conv2", b + "bn2", w3, padding=1)
                    if stride > 1:
                        o = rnd(Fn.avg_pool2d(o, stride), act)
                    o = conv(o, sd, b + "conv3", b + "bn3", w1, relu=False, out="raw")          # fp32 accumulator
                    idn = x
                    if stride > 1 or inplanes != planes * 4:
                        idn = rnd(Fn.avg_pool2d(x, stride), act) if stride > 1 else x
                        idn = rnd(conv(idn, sd, b + "downsample.0", b + "downsample.1", wd, relu=False, out="raw"), act)
                    if dr:
                        o = (o.half() + idn.half()).float() if False else rnd(rnd(o, "f16") + idn, "f16")
                    x = rnd(torch.relu(o + idn), res) if not dr else torch.relu(o)
                    inplanes = planes * 4
            F_, C = x.shape[0], x.shape[1]
            t = x.flatten(start_dim=2).permute(0, 2, 1)
            t = rnd(torch.cat([t.mean(dim=1, keepdim=True), t], dim=1) + sd[P + "attnpool.positional_embedding"], pool)
            hd = C // heads
            q = (t[:, :1] @ rnd(sd[P + "attnpool.q_proj.weight"], pool).t() + sd[P + "attnpool.q_proj.bias"]).reshape(F_, 1, heads, hd).transpose(1, 2)
            k = (t @ rnd(sd[P + "attnpool.k_proj.weight"], pool).t() + sd[P + "attnpool.k_proj.bias"]).reshape(F_, -1, heads, hd).transpose(1, 2)
            v = (t @ rnd(sd[P + "attnpool.v_proj.weight"], pool).t() + sd[P + "attnpool.v_proj.bias"]).reshape(F_, -1, heads, hd).transpose(1, 2)
            att = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
            o = (att @ v).transpose(1, 2).reshape(F_, C)
            outs.append(o @ sd[P + "attnpool.c_proj.weight"].t() + sd[P + "attnpool.c_proj.bias"])
        return torch.cat(outs, 0)
    return tower


def report(sch, lg, ref, t0):
    print("  %-40s max |dlogits| %.2e  rms %.2e  (%.0f s)" % (sch, float((lg - ref).abs().max()), float((lg - ref).pow(2).mean().sqrt()),
                                                              time.time() - t0), flush=True)


def run(case, schemes):
    gold = _cases.load_golden(case)
    meta = gold["meta"]
    a, sd, tt, te, ep = _cases.case_inputs(meta)
    kw = dict(frames=meta["T"], merge_before=meta.get("merge_before", False), single_direct=meta.get("single_direct", False),
              depth=meta.get("depth", 1))
    ref = torch.from_numpy(gold["logits"]).float().reshape(-1)
    orig = orc.resnet_forward
    print("%s: %d logits, spread %.3f" % (case, ref.numel(), float(ref.max() - ref.min())), flush=True)
    for sch in schemes:
        s = dict(kv.split("=") for kv in sch.split(",") if kv)
        orc.resnet_forward = make_tower(s)
        t0 = time.time()
        lg = orc.head_forward(ep, sd, tt, te, a, **kw)["logits"].reshape(-1)
        orc.resnet_forward = orig
        report(sch, lg, ref, t0)


def run_episode(arch_name, shot, q, T, schemes, seed=18, episode=3, lowfreq=2.0):
    import importlib
    synth = importlib.import_module("clip-fsar_amd.synth")
    a = synth.ARCHS[arch_name]
    sd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict(arch_name, seed).items()}
    tt = torch.from_numpy(synth.text_features(64, a["embed"], "train", seed))
    te = torch.from_numpy(synth.text_features(24, a["embed"], "test", seed))
    ep = {k: torch.from_numpy(v) for k, v in synth.make_episode(5, shot, q, T, a["res"], 24, episode, seed, lowfreq=lowfreq).items()}
    orig = orc.resnet_forward
    ref = orc.head_forward(ep, sd, tt, te, a, frames=T)["logits"].reshape(-1)
    print("%s %d-shot q=%d T=%d: %d logits, spread %.3f" % (arch_name, shot, q, T, ref.numel(), float(ref.max() - ref.min())), flush=True)
    for sch in schemes:
        s = dict(kv.split("=") for kv in sch.split(",") if kv)
        orc.resnet_forward = make_tower(s)
        t0 = time.time()
        lg = orc.head_forward(ep, sd, tt, te, a, frames=T)["logits"].reshape(-1)
        orc.resnet_forward = orig
        report(sch, lg, ref, t0)


DEFAULT = ["act=x,w=x", "act=bf16,w=bf16", "act=f16,w=f16", "act=f16,w=x", "act=x,w=f16", "act=f16,w=m", "act=f16,w=f16,res=x"]

if __name__ == "__main__":
    if sys.argv[1] == "episode":
        run_episode(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6:] or DEFAULT)
    else:
        run(sys.argv[1], sys.argv[2:] or DEFAULT)
