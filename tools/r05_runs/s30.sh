#!/bin/bash
# r05 session 30: row bands interleaved over the XCDs (one contiguous front) against XCD-contiguous ranges, stand-alone, M = 2 216 bands (a multiple of 8).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s30; mkdir -p $O
AB_FPE=256 AB_TOK=277 AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py 8 0:0 0:67108864 0:256 0:67109120 > $O/xcd_interleave.log 2>&1; grep "variant" $O/xcd_interleave.log
