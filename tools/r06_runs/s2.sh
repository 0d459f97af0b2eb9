#!/bin/bash
# r06 session 2: where fp16_strict's first form does not help (ViT-L/14 high contrast, outlier channels): which existing device does (split out_proj / all
# weights); RN50 tower: batch 32 against batch 1 and against the fp32 mode per episode.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_s2; mkdir -p $O
timeout 2400 python tools/parity_multi.py --modes "fp16_strict|fp16_strict;CFSAR_FP16_SPLIT=out;CFSAR_FP16_MCORR=qkv,fc,pr|fp16_strict;CFSAR_FP16_SPLIT=qkv,out,fc,pr;CFSAR_FP16_MCORR=|fp16_strict;CFSAR_FP16_SPLIT=out,pr;CFSAR_FP16_MCORR=qkv,fc" hc_cfg4_L14_5w1s_T16 oc_cfg2_B16_5w1s_T8 mc_cfg4_L14_5w1s_T16 hc_cfg2_B16_5w1s_T8 > $O/parity_multi.log 2>&1; grep -v amdgpu.ids $O/parity_multi.log | tail -18
timeout 900 python - > $O/rn50_batch.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import torch
from _cases import case_inputs, load_golden, run_engine, maxdiff
g = load_golden("rn50_5w1s_T2"); m = dict(g["meta"]); m["T"] = 8
B = 32
eps = [case_inputs(m, episode=300 + e)[4] for e in range(B)]
a, sd, tt, te, _ = case_inputs(m)
l32, _ = run_engine(m, a, sd, tt, te, eps, "fp32")
l32_1 = torch.cat([run_engine(m, a, sd, tt, te, [eps[i]], "fp32")[0] for i in (0, 15, 31)])
print("fp32: batch 32 vs alone (episodes 0, 15, 31):", [maxdiff(l32[i], l32_1[j]) for j, i in enumerate((0, 15, 31))])
for prec in ("bf16", "fp16"):
    for nb in (32, 16, 8, 2):
        lb, _ = run_engine(m, a, sd, tt, te, eps[:nb], prec)
        d = (lb - l32[:nb]).abs().reshape(nb, -1).max(1).values
        print(prec, "batch", nb, "vs fp32 per episode: max %.2e  first %.2e  last %.2e  rms %.2e" % (float(d.max()), float(d[0]), float(d[-1]), float((lb - l32[:nb]).pow(2).mean().sqrt())))
        l1, _ = run_engine(m, a, sd, tt, te, [eps[0]], prec)
        l1b, _ = run_engine(m, a, sd, tt, te, [eps[nb - 1]], prec)
        print("   batch", nb, "vs alone: episode 0 %.2e, last %.2e" % (maxdiff(lb[0], l1[0]), maxdiff(lb[nb - 1], l1b[0])))
PY
grep -v amdgpu.ids $O/rn50_batch.log | tail -22
