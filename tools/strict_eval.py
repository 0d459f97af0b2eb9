"""Dev tool (GPU box): deviation of 16-bit numerics modes from the fp32 mode on N FRESH episodes per configuration (the fp32 mode is within 1e-5 of the
reference on every golden): rms / p99 / max |dlogits|, how many episodes have their largest deviation over 7e-4 / 8e-4 / 1e-3.  Round 6: the evidence
behind precision "fp16_strict" (profiles/r06_strict_*.{txt,json}).

usage: python tools/strict_eval.py [--episodes 64] [--configs cfg2,cfg3,cfg4] MODE [MODE ...]
  MODE = precision[;ENV=VALUE ...]      e.g.  fp16   fp16_strict   "fp16_strict;CFSAR_FP16_SPLIT=out;CFSAR_FP16_MCORR=qkv,fc,pr"
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import clip_fsar_amd.synth as synth
from _cases import case_inputs, load_golden, run_engine

CASES = {"cfg2": "cfg2_B16_5w1s_T8", "cfg3": "cfg3_B16_5w5s_T8_mb", "cfg4": "cfg4_L14_5w1s_T16"}
argv = sys.argv[1:]
NEP, cfgs, lowfreq = 64, ["cfg2", "cfg3", "cfg4"], None
while argv and argv[0].startswith("--"):
    if argv[0] == "--episodes":
        NEP = int(argv[1])
    elif argv[0] == "--configs":
        cfgs = argv[1].split(",")
    elif argv[0] == "--lowfreq":
        lowfreq = float(argv[1])
    else:
        raise SystemExit("unknown option %s" % argv[0])
    argv = argv[2:]
modes = argv or ["fp16", "fp16_strict"]
ENV_KEYS = ("CFSAR_FP16_SPLIT", "CFSAR_FP16_MCORR")
table = {}
for cfg in cfgs:
    g = load_golden(CASES[cfg])
    m = dict(g["meta"])
    if lowfreq is not None:
        m["lowfreq"] = lowfreq
    a, sd, tt, te, _ = case_inputs(m)
    eps = [{k: torch.from_numpy(v) for k, v in synth.make_episode(way=m["way"], shot=m["shot"], query_per_class=m["q"], frames=m["T"], res=a["res"],
                                                                  n_test_classes=m["n_test"], episode=1000 + e, seed=m["seed"],
                                                                  lowfreq=m.get("lowfreq", 0.0)).items()} for e in range(NEP)]
    chunk = 4 if m["arch"] == "ViT-L/14" else 8
    run_all = lambda prec: torch.cat([run_engine(m, a, sd, tt, te, eps[i:i + chunk], prec, vit_options={})[0] for i in range(0, NEP, chunk)])
    for k in ENV_KEYS:
        os.environ.pop(k, None)
    ref = run_all("fp32")
    spread = float((ref.reshape(NEP, -1).max(1).values - ref.reshape(NEP, -1).min(1).values).mean())
    table[cfg] = {"episodes": NEP, "mean_spread": spread, "modes": {}}
    for spec in modes:
        parts = spec.split(";")
        for k in ENV_KEYS:
            os.environ.pop(k, None)
        for kv in parts[1:]:
            k, v = kv.split("=", 1)
            os.environ[k] = v
        d = (run_all(parts[0]) - ref).abs()
        per_ep = d.reshape(NEP, -1).max(1).values
        flat = d.flatten().sort().values
        st = {"rms": float(d.pow(2).mean().sqrt()), "p99": float(flat[int(0.99 * (len(flat) - 1))]), "max": float(d.max()),
              "median_episode_max": float(per_ep.median()), "episodes_over_7e-4": int((per_ep > 7e-4).sum()),
              "episodes_over_8e-4": int((per_ep > 8e-4).sum()), "episodes_over_1e-3": int((per_ep > 1e-3).sum())}
        table[cfg]["modes"][spec] = st
        print("%-5s %-62s %d episodes (spread %.2f): rms %.2e p99 %.2e max %.2e | episode max: median %.2e, > 7e-4: %d, > 8e-4: %d, > 1e-3: %d" % (
            cfg, spec, NEP, spread, st["rms"], st["p99"], st["max"], st["median_episode_max"], st["episodes_over_7e-4"], st["episodes_over_8e-4"],
            st["episodes_over_1e-3"]), flush=True)
for k in ENV_KEYS:
    os.environ.pop(k, None)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(table, open(os.path.join(ROOT, "gpurun_out", "strict_eval_%dep.json" % NEP), "w"), indent=1)
