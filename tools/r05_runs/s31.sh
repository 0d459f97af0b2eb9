#!/bin/bash
# r05 session 31: column-fast tile walk on the N = 768 launches (out_proj, c_proj; fp16-stream residual instances), by band group, 36 and 16 episodes.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s31; mkdir -p $O
# dbg 256 = column-fast; bits 9-11 select the group: 1 -> 4, 2 -> 16, 3 -> 2, 4 -> 32, 5 -> 1, 6 -> 3, 7 -> 6 (0 = 8)
AB_SHAPES=out,proj AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py 36 0:0 0:256 0:768 0:1280 0:1792 0:2304 0:2816 0:3328 0:3840 0:512 0:1536 > $O/colfast_36.log 2>&1; grep "variant" $O/colfast_36.log
AB_SHAPES=out,proj AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py 16 0:0 0:256 0:768 0:1280 0:2816 0:3840 > $O/colfast_16.log 2>&1; grep "variant" $O/colfast_16.log
