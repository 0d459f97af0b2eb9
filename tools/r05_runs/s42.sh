#!/bin/bash
# r05 session 42: tile-walk groups for the RN50 tower's launches on the persistent kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s42; mkdir -p $O
CFSAR_DEV_LIB=1 timeout 1500 python tools/rn_walk_ab.py bf16 > $O/rn_walk.log 2>&1; grep "band groups" $O/rn_walk.log; tail -2 $O/rn_walk.log | grep -i "error"
