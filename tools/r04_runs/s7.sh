#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s7; mkdir -p $O
VARIANT_FILTER="mcorr qkv,out,fc|mcorr qkv,fc|mcorr qkv,out|mcorr all" timeout 2400 python tools/fp16_variants.py --episodes 16 cfg2_B16_5w1s_T8 cfg3_B16_5w5s_T8_mb cfg4_L14_5w1s_T16 > $O/variants_16ep.txt 2>&1
cp gpurun_out/fp16_variants_16ep.json $O/ 2>/dev/null
grep -v amdgpu.ids $O/variants_16ep.txt | tail -40
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-leg --precision fp16"
for v in "qkv,out,fc" "qkv,fc" "qkv,out,fc,pr"; do CFSAR_FP16_SPLIT= CFSAR_FP16_MCORR=$v timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['parity']['max_abs_dlogits'])"; done
