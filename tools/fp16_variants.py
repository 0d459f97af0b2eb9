"""Dev tool (GPU box): the fp16 numerics mode's options (wide residual add, two-word stream, split weights per GEMM) against the
reference goldens -> max / rms |logits - reference| per case and variant (gpurun_out/fp16_variants.json); profiles/r04_parity_table.md
is made from it.  usage: python tools/fp16_variants.py [case ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from _cases import case_inputs, load_golden, run_engine

W = dict(CFSAR_FP16_WIDE="1", CFSAR_FP16_LO="1")
VARIANTS = [("r3 (fp16 add)", dict(CFSAR_FP16_WIDE="0", CFSAR_FP16_SPLIT="", CFSAR_FP16_MCORR="")),
            ("wide+lo", dict(W, CFSAR_FP16_SPLIT="", CFSAR_FP16_MCORR="")),
            ("wide+lo+split qkv,out,pr", dict(W, CFSAR_FP16_SPLIT="qkv,out,pr", CFSAR_FP16_MCORR="")),
            ("wide+lo+split all", dict(W, CFSAR_FP16_SPLIT="qkv,out,fc,pr", CFSAR_FP16_MCORR="")),
            ("wide+lo+mcorr all", dict(W, CFSAR_FP16_SPLIT="", CFSAR_FP16_MCORR="qkv,out,fc,pr")),
            ("wide+lo+mcorr qkv,out,pr", dict(W, CFSAR_FP16_SPLIT="", CFSAR_FP16_MCORR="qkv,out,pr")),
            ("wide+lo+split out+mcorr qkv,fc,pr", dict(W, CFSAR_FP16_SPLIT="out", CFSAR_FP16_MCORR="qkv,fc,pr")),
            ("wide+lo+split out+mcorr qkv,pr", dict(W, CFSAR_FP16_SPLIT="out", CFSAR_FP16_MCORR="qkv,pr")),
            ("wide+lo+split out+mcorr qkv", dict(W, CFSAR_FP16_SPLIT="out", CFSAR_FP16_MCORR="qkv")),
            ("wide+lo+mcorr qkv,out,fc", dict(W, CFSAR_FP16_SPLIT="", CFSAR_FP16_MCORR="qkv,out,fc")),
            ("wide+lo+mcorr qkv,fc", dict(W, CFSAR_FP16_SPLIT="", CFSAR_FP16_MCORR="qkv,fc")),
            ("wide+lo+mcorr qkv,out", dict(W, CFSAR_FP16_SPLIT="", CFSAR_FP16_MCORR="qkv,out")),
            ("wide+mcorr all (one-word stream)", dict(CFSAR_FP16_WIDE="1", CFSAR_FP16_LO="0", CFSAR_FP16_SPLIT="", CFSAR_FP16_MCORR="qkv,out,fc,pr"))]
if os.environ.get("VARIANT_FILTER"):
    VARIANTS = [v for v in VARIANTS if any(t in v[0] for t in os.environ["VARIANT_FILTER"].split("|"))]
# `--episodes n`: instead of the one golden episode per case, n fresh episodes per case against the fp32 MODE of the same engine (itself within
# 1e-5 of the reference on every golden): 5 n ... 25 n logits per case -> rms / max / how many episodes exceed 1e-3 are steady statistics
NEP = 0
if "--episodes" in sys.argv:
    i = sys.argv.index("--episodes")
    NEP = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
cases = sys.argv[1:] or ["cfg2_B16_5w1s_T8", "cfg3_B16_5w5s_T8_mb", "cfg4_L14_5w1s_T16", "t_5w1s_T8", "t_5w3s_T16_mb_d2", "t197_5w1s_T2"]
table = {}
for name in cases:
    g = load_golden(name); m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    ref = torch.from_numpy(g["logits"])
    table[name] = {}
    if NEP:
        import clip_fsar_amd.synth as synth
        eps = [{k: torch.from_numpy(v) for k, v in synth.make_episode(way=m["way"], shot=m["shot"], query_per_class=m["q"], frames=m["T"],
                                                                      res=a["res"], n_test_classes=m["n_test"], episode=1000 + e,
                                                                      seed=m["seed"], lowfreq=m.get("lowfreq", 0.0)).items()} for e in range(NEP)]
        chunk = 4 if m["arch"] == "ViT-L/14" else 8
        run_all = lambda prec: torch.cat([run_engine(m, a, sd, tt, te, eps[i:i + chunk], prec)[0] for i in range(0, NEP, chunk)])
        ref32 = run_all("fp32")
        for vn, env in VARIANTS:
            os.environ.update(env)
            d = run_all("fp16") - ref32
            per_ep = d.abs().reshape(NEP, -1).max(1).values
            table[name][vn] = {"episodes": NEP, "max": float(d.abs().max()), "rms": float(d.pow(2).mean().sqrt()),
                               "median_episode_max": float(per_ep.median()), "episodes_over_1e-3": int((per_ep > 1e-3).sum()),
                               "episodes_over_7e-4": int((per_ep > 7e-4).sum())}
            print("%-22s %-36s %d episodes: max %.2e rms %.2e median episode max %.2e  > 7e-4: %d  > 1e-3: %d" % (
                name, vn, NEP, table[name][vn]["max"], table[name][vn]["rms"], table[name][vn]["median_episode_max"],
                table[name][vn]["episodes_over_7e-4"], table[name][vn]["episodes_over_1e-3"]), flush=True)
        continue
    for vn, env in VARIANTS:
        os.environ.update(env)
        lg, _ = run_engine(m, a, sd, tt, te, [ep], "fp16")
        d = (lg[0] - ref)
        table[name][vn] = {"max": float(d.abs().max()), "rms": float(d.pow(2).mean().sqrt()),
                           "argmax": int((lg[0].argmax(1) == ref.argmax(1)).sum()), "n": int(ref.shape[0])}
        print("%-24s %-22s max %.2e rms %.2e argmax %d/%d" % (name, vn, table[name][vn]["max"], table[name][vn]["rms"],
                                                            table[name][vn]["argmax"], ref.shape[0]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(table, open(os.path.join(ROOT, "gpurun_out", "fp16_variants%s.json" % ("_%dep" % NEP if NEP else "")), "w"), indent=1)
