#!/bin/bash
# r05 session 40: operand path of the LN-folded launches (QKV, c_fc) in situ, developer library.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s40; mkdir -p $O
for a in "fp16 36 cfg2" "bf16 16 cfg2" "bf16 11 cfg4" "fp16 11 cfg4" "bf16 12 cfg3" "bf16 2 cfg2"; do
  set -- $a
  CFSAR_DEV_LIB=1 timeout 900 python tools/lnfold_path_ab.py $1 $2 $3 > $O/$3_$1_$2.log 2>&1; grep "episodes per step" $O/$3_$1_$2.log
done
