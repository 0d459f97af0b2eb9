#!/bin/bash
# r05 session 22: walk direction of consecutive tower launches (Infinity Cache reuse of the producer's last rows): bench legs, developer library.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s22; mkdir -p $O
CFSAR_DEV_LIB=1 timeout 1200 python tools/walk_direction_ab.py bf16 16 18 36 > $O/walk_bf16.log 2>&1; grep "episodes per step" $O/walk_bf16.log
CFSAR_DEV_LIB=1 timeout 1200 python tools/walk_direction_ab.py fp16 18 36 > $O/walk_fp16.log 2>&1; grep "episodes per step" $O/walk_fp16.log
