"""Dev tool (GPU box): does an episode's fp16-mode result depend on the batch it is served in?  Episode 0 of cfg2 alone vs inside 16.
usage: python tools/batch_invariance_probe.py   (env: CFSAR_FP16_MCORR / CFSAR_FP16_SPLIT / CFSAR_FUSED_UMEANS / CFSAR_FUSED_XMEANS)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from _cases import case_inputs, load_golden, run_engine
g = load_golden("cfg2_B16_5w1s_T8"); m = g["meta"]
a, sd, tt, te, ep0 = case_inputs(m)
eps = [ep0] + [case_inputs(m, episode=m["episode"] + e)[4] for e in range(1, 16)]
ref = torch.from_numpy(g["logits"])
for prec in sys.argv[1:] or ["fp16"]:
    t1, t16 = {}, {}
    l1, _ = run_engine(m, a, sd, tt, te, [eps[0]], prec, taps=None)
    l16, _ = run_engine(m, a, sd, tt, te, eps, prec)
    print("%s MCORR=%s SPLIT=%s UM=%s XM=%s: |B1 - B16| max %.3e   B1 vs golden %.3e   B16 vs golden %.3e" % (
        prec, os.environ.get("CFSAR_FP16_MCORR"), os.environ.get("CFSAR_FP16_SPLIT"), os.environ.get("CFSAR_FUSED_UMEANS"), os.environ.get("CFSAR_FUSED_XMEANS"),
        float((l1[0] - l16[0]).abs().max()), float((l1[0] - ref).abs().max()), float((l16[0] - ref).abs().max())), flush=True)
