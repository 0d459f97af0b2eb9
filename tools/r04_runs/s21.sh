#!/bin/bash
# same-box A/B of the wide residual epilogue: v_fma_mix form (product) vs the plain C form (variant plainc), fp16 headline, interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --precision fp16 --no-cpu-baseline --steps 12 --warmup 3"
for r in 1 2 3; do
  for v in product plainc; do
    if [ $v = plainc ]; then export CFSAR_LIB_PATH=$GRAFT_REPO_ROOT/clip-fsar_amd/libclipfsar_hip_plainc.so; else unset CFSAR_LIB_PATH; fi
    timeout 600 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['parity']['max_abs_dlogits'])"
  done
done
