#!/bin/bash
# r05 session 37: out_proj (fp16-stream residual instance): register-staged (20) against LDS-DMA (28) operand path, groups of 8 / 6, at 36, 16 and 1 episode(s); ViT-L/14 shape.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s37; mkdir -p $O
for b in 36 16 18 1; do
AB_SHAPES=out AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py $b 28:0 20:0 28:3584 20:3584 > $O/out_b$b.log 2>&1; grep variant $O/out_b$b.log
done
AB_D=1024 AB_TOK=257 AB_FPE=160 AB_SHAPES=out AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py 11 28:0 20:0 28:3584 20:3584 > $O/out_L14.log 2>&1; grep variant $O/out_L14.log
