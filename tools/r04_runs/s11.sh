#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "lnfold_hp or residual_wide or frame_col" > $O/pytest_new.log 2>&1; echo "rc $?" >> $O/pytest_new.log
tail -6 $O/pytest_new.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-leg --precision fp16"
for v in 0 1 0 1; do CFSAR_FUSED_XMEANS=$v timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_xmeans=$v', d['value'], d['ms_per_step'], d['parity']['max_abs_dlogits'])"; done
VARIANT_FILTER="wide+lo+mcorr all" timeout 2400 python tools/fp16_variants.py --episodes 16 cfg2_B16_5w1s_T8 cfg3_B16_5w5s_T8_mb cfg4_L14_5w1s_T16 2>&1 | grep -v amdgpu.ids | tail -4
VARIANT_FILTER="wide+lo+mcorr all" timeout 2400 python tools/fp16_variants.py cfg2_B16_5w1s_T8 cfg3_B16_5w5s_T8_mb cfg4_L14_5w1s_T16 2>&1 | grep -v amdgpu.ids | tail -4
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "rc $?" >> $O/pytest_all.log; tail -5 $O/pytest_all.log
