#!/bin/bash
# r05 session 23: the fp16 mode of the FINAL build (frame_gemm, patch_embed) on 64 fresh episodes per configuration against the fp32 mode
# (itself within 1e-5 of the reference on every golden): the steady statistic next to the 13-episode reference goldens of profiles/r05_parity_table.md.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s23; mkdir -p $O
VARIANT_FILTER="wide+lo+mcorr all" timeout 1500 python tools/fp16_variants.py --episodes 64 cfg2_B16_5w1s_T8 cfg3_B16_5w5s_T8_mb cfg4_L14_5w1s_T16 > $O/fresh64.log 2>&1; tail -12 $O/fresh64.log
cp gpurun_out/fp16_variants.json $O/fresh64.json 2>/dev/null
