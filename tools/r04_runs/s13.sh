#!/bin/bash
# session 13: compiler-visible tail loads (CFSAR_VISIBLE_TAIL_LOADS) A/B: speed (bf16 / fp16), the packed build with them, bit stability
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-leg"
one() { env "$@" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['parity']['max_abs_dlogits'])"; }
for i in 1 2 3; do
  echo -n "bf16 product "; one $B 2>/dev/null
  echo -n "bf16 visible "; one CFSAR_LIB_PATH=clip-fsar_amd/libclipfsar_hip_vis.so $B 2>/dev/null
done
for i in 1 2; do
  echo -n "fp16 product "; one $B --precision fp16 2>/dev/null
  echo -n "fp16 visible "; one CFSAR_LIB_PATH=clip-fsar_amd/libclipfsar_hip_vis.so $B --precision fp16 2>/dev/null
done
echo "packed + visible loads, end to end:"; CFSAR_LIB_PATH=clip-fsar_amd/libclipfsar_hip_vispk.so python tools/batch_invariance_probe.py 2>&1 | grep -E "MCORR|fault"
CFSAR_LIB_PATH=clip-fsar_amd/libclipfsar_hip_vis.so python -m pytest tests/test_gpu_kernels.py -q -k "bit_stable or lnfold or residual" 2>&1 | tail -2
CFSAR_LIB_PATH=clip-fsar_amd/libclipfsar_hip_vis.so python -m pytest tests/test_gpu_e2e.py -q 2>&1 | tail -2
echo -n "B=1 product "; one $B --episodes-per-step 1 2>/dev/null; echo -n "B=1 visible "; one CFSAR_LIB_PATH=clip-fsar_amd/libclipfsar_hip_vis.so $B --episodes-per-step 1 2>/dev/null
