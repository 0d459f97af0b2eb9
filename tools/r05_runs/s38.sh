#!/bin/bash
# r05 session 38: out_proj's operand path in situ (bench legs, developer library).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s38; mkdir -p $O
CFSAR_DEV_LIB=1 timeout 1500 python tools/outproj_path_ab.py bf16 > $O/bf16.log 2>&1; grep "episodes per step" $O/bf16.log
CFSAR_DEV_LIB=1 timeout 1500 python tools/outproj_path_ab.py fp16 > $O/fp16.log 2>&1; grep "episodes per step" $O/fp16.log
