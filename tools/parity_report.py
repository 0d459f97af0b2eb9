"""Dev tool (GPU box): max |logits - reference golden| for every golden case in both precisions -> stdout table."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from _cases import SMALL_CASES, case_inputs, load_golden, maxdiff, run_engine
cases = SMALL_CASES + ["cfg2_B16_5w1s_T8", "cfg3_B16_5w5s_T8_mb", "cfg4_L14_5w1s_T16", "rn50_5w1s_T2"]
print("| case | logits spread | fp32: max abs dlogits | fp32: max abs dfeats | bf16: max abs dlogits | bf16 argmax agrees |")
print("|---|---|---|---|---|---|")
for name in cases:
    g = load_golden(name); m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    taps = {}
    l32, _ = run_engine(m, a, sd, tt, te, [ep], "fp32", taps=taps)
    S = m["way"] * m["shot"]
    f = taps["feats"].cpu()[0].reshape(-1, a["embed"])
    df = max(maxdiff(f[:S * m["T"]], g["feats_s"]), maxdiff(f[S * m["T"]:], g["feats_q"]))
    l16, _ = run_engine(m, a, sd, tt, te, [ep], "bf16")
    ref = torch.from_numpy(g["logits"])
    print("| %s | %.3f | %.2e | %.2e | %.4f | %d/%d |" % (name, float(ref.max() - ref.min()), maxdiff(l32[0], ref), df,
          maxdiff(l16[0], ref), int((l16[0].argmax(1) == ref.argmax(1)).sum()), ref.shape[0]))
