"""Dev tool (GPU box): run-to-run bit-stability of whole engine forwards (same inputs, same engine) for the three towers and
both precisions -- a cheap net for rare data corruption anywhere in the path (the stale-lanes fault of round 2 first showed up as
1e-4 run-to-run differences of the logits).  usage: python tools/determinism_probe.py [repeats] [--partner]
--partner: a second engine (ViT-B/16 bf16, 4 episodes) runs back to back on ANOTHER stream during every repeat (the condition that made the
round-2 fault frequent: kernels of two streams sharing the chip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import clip_fsar_amd.synth as synth
from clip_fsar_amd.engine import ClipFsarEngine
R = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 12
PARTNER = "--partner" in sys.argv
dev = torch.device("cuda")
partner = None
if PARTNER:
    pa = synth.ARCHS["ViT-B/16"]
    psd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict("ViT-B/16", 19).items()}
    peng = ClipFsarEngine(pa, psd, torch.from_numpy(synth.text_features(64, pa["embed"], "train", 19)),
                          torch.from_numpy(synth.text_features(24, pa["embed"], "test", 19)), precision="bf16", device=dev)
    peps = [synth.make_episode(5, 1, 1, 8, pa["res"], 24, 50 + e, 19) for e in range(4)]
    pst = lambda k: torch.stack([torch.from_numpy(e[k]) for e in peps]).to(dev)
    pargs = (pst("support_set"), pst("target_set"), pst("support_labels"), pst("real_support_labels"))
    pstream = torch.cuda.Stream()
    def partner():
        with torch.cuda.stream(pstream):
            for _ in range(2):
                peng.forward(*pargs, way=5, T=8)
for arch, T, B, prec in (("ViT-B/16", 8, 1, "bf16"), ("ViT-B/16", 8, 4, "bf16"), ("ViT-B/16", 8, 16, "bf16"), ("ViT-B/16", 8, 16, "fp16"),
                         ("ViT-B/16", 8, 1, "fp16"), ("ViT-L/14", 16, 1, "bf16"), ("ViT-L/14", 16, 2, "fp16"), ("RN50", 8, 2, "bf16"),
                         ("RN50", 8, 8, "fp16"), ("ViT-B/16", 8, 1, "fp32")):
    a = synth.ARCHS[arch]
    sd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict(arch, 18).items()}
    tt = torch.from_numpy(synth.text_features(64, a["embed"], "train", 18)); te = torch.from_numpy(synth.text_features(24, a["embed"], "test", 18))
    eng = ClipFsarEngine(a, sd, tt, te, precision=prec, device=dev)
    eps = [synth.make_episode(5, 1, 1, T, a["res"], 24, e, 18) for e in range(B)]
    st = lambda k: torch.stack([torch.from_numpy(e[k]) for e in eps]).to(dev)
    args = (st("support_set"), st("target_set"), st("support_labels"), st("real_support_labels"))
    ref, bad, worst = None, 0, 0.0
    reps = R if prec != "fp32" else 3
    for it in range(reps):
        if partner is not None and it > 0:                   # the reference repeat runs alone
            partner()
        lo, cl = eng.forward(*args, way=5, T=T)
        torch.cuda.synchronize()
        if ref is None:
            ref = (lo.clone(), cl.clone())
        else:
            d = max(float((lo - ref[0]).abs().max()), float((cl - ref[1]).abs().max()))
            if d != 0.0:
                bad += 1; worst = max(worst, d)
    print("%-9s T=%-2d B=%-2d %s%s: %d of %d repeats differ (max |d| %.3e)" % (arch, T, B, prec, " + partner stream" if PARTNER else "", bad, reps - 1, worst), flush=True)
    del eng
    torch.cuda.empty_cache()
