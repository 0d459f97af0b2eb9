#!/bin/bash
# r05 session 20: SURVEY K1 - the patch gather inside the patch-embed GEMM (cfsar_patch_embed): bit-equality tests, timing against the three-launch
# form, the bench leg with and without it.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -m gpu -k "patch_embed or fused_patch or im2col" > $O/pytest_k1.log 2>&1; tail -5 $O/pytest_k1.log
timeout 900 python tools/patch_embed_time.py > $O/patch_embed_time.log 2>&1; tail -12 $O/patch_embed_time.log
