#!/bin/bash
# r05 session 33: band-fastest tile walk by group size on the short-K launches at 36 episodes (QKV, c_fc: plain instances; out_proj fp16-stream residual).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s33; mkdir -p $O
# bits 9-11: 1 -> 4 bands, 2 -> 16, 3 -> 2, 4 -> 32, 5 -> 1, 6 -> 3, 7 -> 6 (0 = 8)
AB_SHAPES=qkv,fc,out AB_STREAM=fp16 timeout 1200 python tools/gemm_ab.py 36 0:0 0:512 0:1024 0:1536 0:2048 0:2560 0:3072 0:3584 > $O/groups_36.log 2>&1; grep "variant" $O/groups_36.log
