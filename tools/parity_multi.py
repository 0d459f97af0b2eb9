"""Dev tool (GPU box): every numerics mode against the MULTI-EPISODE reference goldens (tests/golden/multi_*.npz, oracle/make_golden.py --multi:
13 episodes = 65 logit rows per configuration, at the generator's standard contrast `mc_*` and at high contrast `hc_*`, logits spread 3-4).
Per (case, mode): rms / p99 / max of |logits - reference| over all elements, the largest per-episode deviation and how many episodes exceed 1e-3,
the same relative to the episode's logits spread -> gpurun_out/parity_multi.json (profiles/r05_parity_table.md is made from it).
usage: python tools/parity_multi.py [--modes fp32,fp16,bf16] [case ...]      a mode may carry environment switches, modes are then separated by "|":
                                                                                 --modes "fp16_strict|fp16_strict;CFSAR_FP16_SPLIT=out;CFSAR_FP16_MCORR=qkv,fc,pr" """
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from _cases import multi_case_stats, MULTI_CASES

modes = ["fp32", "fp16", "bf16"]
if "--modes" in sys.argv:
    i = sys.argv.index("--modes"); modes = sys.argv[i + 1].split("|" if "|" in sys.argv[i + 1] or ";" in sys.argv[i + 1] else ","); del sys.argv[i:i + 2]
cases = sys.argv[1:] or [c for c in MULTI_CASES if os.path.exists(os.path.join(ROOT, "tests", "golden", "multi_%s.npz" % c))]
table = {}
for name in cases:
    table[name] = {}
    for mode in modes:
        for k in ("CFSAR_FP16_SPLIT", "CFSAR_FP16_MCORR"):
            os.environ.pop(k, None)
        for kv in mode.split(";")[1:]:
            os.environ[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
        try:
            st = multi_case_stats(name, mode.split(";")[0])
        except ValueError as exc:                   # a tower that refuses the mode by name (RN50: fp16_strict)
            print("%-26s %-40s not served: %s" % (name, mode, exc), flush=True)
            continue
        table[name][mode] = st
        print("%-26s %-40s rows %3d spread %.2f | rms %.2e p99 %.2e max %.2e | worst episode %.2e, episodes > 1e-3: %d of %d | max / spread %.2e | argmax %d/%d" % (
            name, mode, st["rows"], st["mean_spread"], st["rms"], st["p99"], st["max"], st["worst_episode_max"], st["episodes_over_1e-3"], st["episodes"],
            st["max_rel_spread"], st["argmax_equal"], st["rows"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(table, open(os.path.join(ROOT, "gpurun_out", "parity_multi.json"), "w"), indent=1)
