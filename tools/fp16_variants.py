"""Dev tool (GPU box): the fp16 numerics mode's options (wide residual add, two-word stream, split weights per GEMM) against the
reference goldens -> max / rms |logits - reference| per case and variant (gpurun_out/fp16_variants.json); profiles/r04_parity_table.md
is made from it.  usage: python tools/fp16_variants.py [case ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from _cases import case_inputs, load_golden, run_engine

VARIANTS = [("r3 (fp16 add)", dict(CFSAR_FP16_WIDE="0", CFSAR_FP16_SPLIT="")),
            ("wide", dict(CFSAR_FP16_WIDE="1", CFSAR_FP16_LO="0", CFSAR_FP16_SPLIT="")),
            ("wide+lo", dict(CFSAR_FP16_WIDE="1", CFSAR_FP16_LO="1", CFSAR_FP16_SPLIT="")),
            ("wide+lo+out", dict(CFSAR_FP16_WIDE="1", CFSAR_FP16_LO="1", CFSAR_FP16_SPLIT="out")),
            ("wide+lo+qkv,out", dict(CFSAR_FP16_WIDE="1", CFSAR_FP16_LO="1", CFSAR_FP16_SPLIT="qkv,out")),
            ("wide+qkv,out", dict(CFSAR_FP16_WIDE="1", CFSAR_FP16_LO="0", CFSAR_FP16_SPLIT="qkv,out")),
            ("wide+lo+qkv,out,pr", dict(CFSAR_FP16_WIDE="1", CFSAR_FP16_LO="1", CFSAR_FP16_SPLIT="qkv,out,pr")),
            ("wide+lo+all", dict(CFSAR_FP16_WIDE="1", CFSAR_FP16_LO="1", CFSAR_FP16_SPLIT="qkv,out,fc,pr"))]
cases = sys.argv[1:] or ["cfg2_B16_5w1s_T8", "cfg3_B16_5w5s_T8_mb", "cfg4_L14_5w1s_T16", "t_5w1s_T8", "t_5w3s_T16_mb_d2", "t197_5w1s_T2"]
table = {}
for name in cases:
    g = load_golden(name); m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    ref = torch.from_numpy(g["logits"])
    table[name] = {}
    for vn, env in VARIANTS:
        os.environ.update(env)
        lg, _ = run_engine(m, a, sd, tt, te, [ep], "fp16")
        d = (lg[0] - ref)
        table[name][vn] = {"max": float(d.abs().max()), "rms": float(d.pow(2).mean().sqrt()),
                           "argmax": int((lg[0].argmax(1) == ref.argmax(1)).sum()), "n": int(ref.shape[0])}
        print("%-24s %-22s max %.2e rms %.2e argmax %d/%d" % (name, vn, table[name][vn]["max"], table[name][vn]["rms"],
                                                            table[name][vn]["argmax"], ref.shape[0]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(table, open(os.path.join(ROOT, "gpurun_out", "fp16_variants.json"), "w"), indent=1)
