#!/bin/bash
# r06 session 1: the fp16_strict mode (exact patch-embed front end) -- kernel + e2e tests, batch-36 parity tests, 64 fresh episodes per configuration,
# multi-episode reference goldens, a first strict bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_s1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "strict or patch_embed" > $O/pytest_kernels.log 2>&1; tail -5 $O/pytest_kernels.log
timeout 2400 python -m pytest tests/test_gpu_e2e.py -q -m gpu -x -k "strict or b36 or harness_batch or b16" > $O/pytest_e2e.log 2>&1; tail -15 $O/pytest_e2e.log
timeout 1500 python tools/strict_eval.py --episodes 64 fp16 fp16_strict "fp16_strict;CFSAR_FP16_SPLIT=qkv,out,fc,pr;CFSAR_FP16_MCORR=" "fp16_strict;CFSAR_FP16_SPLIT=out;CFSAR_FP16_MCORR=qkv,fc,pr" > $O/strict_eval.log 2>&1; grep -v amdgpu.ids $O/strict_eval.log | tail -14; cp gpurun_out/strict_eval_64ep.json $O/
timeout 1500 python tools/parity_multi.py --modes fp16,fp16_strict mc_cfg2_B16_5w1s_T8 hc_cfg2_B16_5w1s_T8 hc_cfg3_B16_5w5s_T8_mb hc_cfg4_L14_5w1s_T16 mc_cfg4_L14_5w1s_T16 oc_cfg2_B16_5w1s_T8 > $O/parity_multi.log 2>&1; grep -v amdgpu.ids $O/parity_multi.log | tail -14; cp gpurun_out/parity_multi.json $O/
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config-legs"
timeout 900 $B > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
timeout 600 $B --precision fp16_strict --no-fp16-leg > $O/bench_strict.json 2> $O/bench_strict.err; tail -c 300 $O/bench_strict.json
python - <<PY
import json
for n in ("bench", "bench_strict"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), "fp16", (d.get("fp16_mode") or {}).get("value"), "strict", (d.get("strict_mode") or {}).get("value"), (d.get("strict_mode") or {}).get("parity", {}).get("max_abs_dlogits"), (d.get("parity") or {}).get("max_abs_dlogits"))
    except Exception as e:
        print(n, "failed", e)
PY
