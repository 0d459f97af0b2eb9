#!/bin/bash
# session 6: steady parity statistics of the fp16-mode variants: 16 fresh episodes per configuration against the fp32 mode
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s6; mkdir -p $O
timeout 2400 python tools/fp16_variants.py --episodes 16 cfg2_B16_5w1s_T8 cfg3_B16_5w5s_T8_mb cfg4_L14_5w1s_T16 > $O/variants_16ep.txt 2>&1
cp gpurun_out/fp16_variants_16ep.json $O/ 2>/dev/null
grep -v amdgpu.ids $O/variants_16ep.txt | tail -40
