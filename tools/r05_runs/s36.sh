#!/bin/bash
# r05 session 36: operand path of the residual launches under the new tile walks (c_proj: column-fastest groups of 16; out_proj: groups of 6), 36 episodes.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s36; mkdir -p $O
AB_SHAPES=proj AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py 36 20:1280 24:1280 28:1280 20:0 > $O/proj_paths.log 2>&1; grep variant $O/proj_paths.log
AB_SHAPES=out AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py 36 28:3584 20:3584 24:3584 28:0 > $O/out_paths.log 2>&1; grep variant $O/out_paths.log
