"""Dev tool (GPU box): max |logits - reference golden| for every golden case in the three numerics modes -> stdout table + JSON
(gpurun_out/parity_table.json: the per-case measured values tests/_cases.py's bounds are derived from)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from _cases import SMALL_CASES, OUTLIER_CASES, case_inputs, load_golden, maxdiff, run_engine
cases = SMALL_CASES + OUTLIER_CASES + ["cfg2_B16_5w1s_T8", "cfg3_B16_5w5s_T8_mb", "cfg4_L14_5w1s_T16", "rn50_5w1s_T2"]
if "--cases" in sys.argv:
    cases = sys.argv[sys.argv.index("--cases") + 1:]
print("| case | logits spread | fp32: max abs dlogits | fp32: max abs dfeats | fp16: max abs dlogits | bf16: max abs dlogits | bf16 argmax agrees |")
print("|---|---|---|---|---|---|---|")
table = {}
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
    lh, _ = run_engine(m, a, sd, tt, te, [ep], "fp16")
    d16h = maxdiff(lh[0], ref)
    s16h = "%.2e (%d/%d)" % (d16h, int((lh[0].argmax(1) == ref.argmax(1)).sum()), ref.shape[0])
    table[name] = {"fp32": maxdiff(l32[0], ref), "fp16": d16h, "bf16": maxdiff(l16[0], ref)}
    if name in OUTLIER_CASES:                          # the unfolded block (separate LayerNorm kernels) on the same outlier statistics
        os.environ["CFSAR_LN_FOLD"] = "0"
        lnf, _ = run_engine(m, a, sd, tt, te, [ep], "bf16")
        os.environ.pop("CFSAR_LN_FOLD")
        table[name]["bf16_unfolded"] = maxdiff(lnf[0], ref)
    print("| %s | %.3f | %.2e | %.2e | %s | %.4f | %d/%d |" % (name, float(ref.max() - ref.min()), maxdiff(l32[0], ref), df, s16h,
          maxdiff(l16[0], ref), int((l16[0].argmax(1) == ref.argmax(1)).sum()), ref.shape[0]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(table, open(os.path.join(ROOT, "gpurun_out", "parity_table.json"), "w"), indent=1)
