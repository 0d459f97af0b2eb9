#!/bin/bash
# r05 session 19: c_proj / out_proj (fp16-stream residual + statistics instances of the 8-wave kernel) by operand path, with the epilogue / store ablations:
# where the 17-20 % to the vendor's plain GEMM on the K = 3 072 shape goes.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s19; mkdir -p $O
AB_STREAM=fp16 AB_SHAPES=out,proj timeout 900 python tools/gemm_ab.py 16 0:0 20:0 22:0 24:0 26:0 28:0 30:0 22:4 30:4 22:16 30:16 22:8 30:8 > $O/gemm_ab_res.log 2>&1; cat $O/gemm_ab_res.log | tail -30
