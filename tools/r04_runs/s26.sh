#!/bin/bash
# raw-stream form of the LN-folded GEMMs' correction (no frame_col_means passes): tests, A/B against CFSAR_FP16_RAWMEANS=0, 16-episode statistics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "lnfold or correction or means" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "fp16 or cfg2_full or cfg3_cfg4 or outlier or pruning" 2>&1 | tail -4
B="python bench.py --precision fp16 --no-cpu-baseline --steps 12 --warmup 3"
for r in 1 2 3; do
  for v in 1 0; do
    CFSAR_FP16_RAWMEANS=$v timeout 600 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rawmeans $v', d['value'], d['ms_per_step'], d['parity']['max_abs_dlogits'])"
  done
done
VARIANT_FILTER="wide+lo+mcorr all" timeout 2400 python tools/fp16_variants.py --episodes 16 cfg2_B16_5w1s_T8 cfg3_B16_5w5s_T8_mb cfg4_L14_5w1s_T16 2>&1 | grep -v amdgpu.ids | tail -4
