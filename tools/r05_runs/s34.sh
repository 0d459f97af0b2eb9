#!/bin/bash
# r05 session 34: band groups 6 / 3 / 2 / 4 against 8 on the short-K launches at 16 episodes (ViT-B/16) and on the ViT-L/14 shapes (11 episodes of 160 frames).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s34; mkdir -p $O
AB_SHAPES=qkv,fc,out AB_STREAM=fp16 timeout 1200 python tools/gemm_ab.py 16 0:0 0:3584 0:3072 0:1536 0:512 > $O/groups_16.log 2>&1; grep "variant" $O/groups_16.log
AB_D=1024 AB_TOK=257 AB_FPE=160 AB_SHAPES=qkv,fc,out AB_STREAM=fp16 timeout 1200 python tools/gemm_ab.py 11 0:0 0:3584 0:3072 0:1536 0:512 > $O/groups_L14.log 2>&1; grep "variant" $O/groups_L14.log
AB_SHAPES=qkv,fc,out AB_STREAM=fp16 timeout 1200 python tools/gemm_ab.py 12 0:0 0:3584 0:3072 0:1536 0:512 > $O/groups_12x3.log 2>&1
