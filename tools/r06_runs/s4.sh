#!/bin/bash
# r06 session 4: the fp16_strict defaults (front end + split QKV weights, corrected out_proj / c_fc / c_proj, raw-stream means) -- accuracy on the six reference
# sets and 64 fresh episodes, the default bench line with its fp16 and strict legs, the whole GPU suite.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_s4; mkdir -p $O
F0="fp16_strict"
F2="fp16_strict;CFSAR_FP16_SPLIT=qkv,fc;CFSAR_FP16_MCORR=out,pr"
timeout 2400 python tools/parity_multi.py --modes "fp16|$F0|$F2" mc_cfg2_B16_5w1s_T8 hc_cfg2_B16_5w1s_T8 hc_cfg3_B16_5w5s_T8_mb hc_cfg4_L14_5w1s_T16 mc_cfg4_L14_5w1s_T16 oc_cfg2_B16_5w1s_T8 > $O/parity_multi.log 2>&1; grep -v amdgpu.ids $O/parity_multi.log | cut -c1-250 | tail -20; cp gpurun_out/parity_multi.json $O/
timeout 2400 python tools/strict_eval.py --episodes 64 fp16 "$F0" "$F2" > $O/strict_eval.log 2>&1; grep -v amdgpu.ids $O/strict_eval.log | cut -c1-260 | tail -10; cp gpurun_out/strict_eval_64ep.json $O/
timeout 2400 python tools/strict_eval.py --episodes 64 --lowfreq 2.0 --configs cfg2,cfg4 fp16 "$F0" > $O/strict_eval_hc.log 2>&1; grep -v amdgpu.ids $O/strict_eval_hc.log | cut -c1-260 | tail -6
timeout 900 python bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('default', d['value'], d['roofline']['frac'], 'fp16', d['fp16_mode']['value'], d['fp16_mode']['relative_to_value'], 'strict', d['strict_mode']['value'], d['strict_mode']['relative_to_value'], d['strict_mode']['parity']['max_abs_dlogits'], d['strict_mode']['roofline']['frac'])"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config-legs --no-fp16-leg --precision fp16_strict > $O/bench_strict.json 2> $O/bench_strict.err; tail -c 200 $O/bench_strict.json
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "rc $?" >> $O/pytest_all.log; tail -12 $O/pytest_all.log
