#!/bin/bash
# r05 session 25: store policy of the LN-folded launches x walk direction (does a consumer find its producer's last rows in the Infinity Cache?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s25; mkdir -p $O
CFSAR_DEV_LIB=1 timeout 1500 python tools/store_walk_ab.py 18 bf16 > $O/store_walk_18.log 2>&1; grep "episodes  store" $O/store_walk_18.log
CFSAR_DEV_LIB=1 timeout 1500 python tools/store_walk_ab.py 36 bf16 > $O/store_walk_36.log 2>&1; grep "episodes  store" $O/store_walk_36.log
