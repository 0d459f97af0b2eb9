#!/bin/bash
# same-box A/B of two builds of the library (product vs clip-fsar_amd/libclipfsar_hip_$1.so: `python clip-fsar_amd/build.py --variant NAME [-D...]`),
# fp16 headline, interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
V=${1:-plainc}
B="python bench.py --precision fp16 --no-cpu-baseline --steps 12 --warmup 3"
for r in 1 2 3; do
  for v in product $V; do
    if [ $v = product ]; then unset CFSAR_LIB_PATH; else export CFSAR_LIB_PATH=$GRAFT_REPO_ROOT/clip-fsar_amd/libclipfsar_hip_$v.so; fi
    timeout 600 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['parity']['max_abs_dlogits'])"
  done
done
