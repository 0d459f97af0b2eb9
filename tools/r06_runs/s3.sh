#!/bin/bash
# r06 session 3: fp16_strict second form (two-word attention output, three-segment out_proj) -- kernel tests, candidates with split LN-folded weights:
# accuracy on the six multi-episode reference sets and 64 fresh episodes, speed of each candidate (product build and the by-ka policy build).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_s3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "pair or two_word or strict or residual_wide or attention_means" > $O/pytest_kernels.log 2>&1; tail -5 $O/pytest_kernels.log
timeout 1800 python -m pytest tests/test_gpu_e2e.py -q -m gpu -k "strict_mode_single or strict_mode_is or b36 or rn50_harness" > $O/pytest_e2e.log 2>&1; tail -8 $O/pytest_e2e.log
S0="fp16_strict"
S1="fp16_strict;CFSAR_FP16_SPLIT=qkv,fc;CFSAR_FP16_MCORR=out,pr"
S2="fp16_strict;CFSAR_FP16_SPLIT=qkv,fc,pr;CFSAR_FP16_MCORR=out"
S3="fp16_strict;CFSAR_FP16_SPLIT=qkv;CFSAR_FP16_MCORR=out,fc,pr"
S4="fp16_strict;CFSAR_FP16_SPLIT=fc;CFSAR_FP16_MCORR=qkv,out,pr"
timeout 3000 python tools/parity_multi.py --modes "$S0|$S1|$S2|$S3|$S4" mc_cfg2_B16_5w1s_T8 hc_cfg2_B16_5w1s_T8 hc_cfg3_B16_5w5s_T8_mb hc_cfg4_L14_5w1s_T16 mc_cfg4_L14_5w1s_T16 oc_cfg2_B16_5w1s_T8 > $O/parity_multi.log 2>&1; grep -v amdgpu.ids $O/parity_multi.log | cut -c1-250 | tail -32; cp gpurun_out/parity_multi.json $O/
timeout 2400 python tools/strict_eval.py --episodes 64 "$S0" "$S1" "$S2" > $O/strict_eval.log 2>&1; grep -v amdgpu.ids $O/strict_eval.log | cut -c1-260 | tail -10; cp gpurun_out/strict_eval_64ep.json $O/
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config-legs --no-fp16-leg --precision fp16_strict"
for lib in product byka; do
  for v in "0::" "1:qkv,fc:out,pr" "2:qkv,fc,pr:out" "3:qkv:out,fc,pr" "4:fc:qkv,out,pr" "5:qkv,out,fc,pr:"; do
    n=${v%%:*}; r=${v#*:}; sp=${r%%:*}; mc=${r#*:}
    if [ "$n" = "0" ]; then unset CFSAR_FP16_SPLIT CFSAR_FP16_MCORR; else export CFSAR_FP16_SPLIT="$sp" CFSAR_FP16_MCORR="$mc"; fi
    if [ "$lib" = "byka" ]; then export CFSAR_LIB_PATH=clip-fsar_amd/libclipfsar_hip_byka.so; else unset CFSAR_LIB_PATH; fi
    timeout 600 $B > $O/bench_${lib}_S$n.json 2> $O/bench_${lib}_S$n.err
    python -c "
import json,sys
try:
    d=json.loads(open('$O/bench_${lib}_S$n.json').read().strip().splitlines()[-1]); print('$lib S$n split=[$sp] mcorr=[$mc]', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'), (d.get('parity') or {}).get('max_abs_dlogits'))
except Exception as e: print('$lib S$n failed', e)"
  done
done
unset CFSAR_FP16_SPLIT CFSAR_FP16_MCORR CFSAR_LIB_PATH
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config-legs > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('default', d['value'], 'fp16', d['fp16_mode']['value'], 'strict', d['strict_mode']['value'], d['strict_mode']['parity']['max_abs_dlogits'])"
