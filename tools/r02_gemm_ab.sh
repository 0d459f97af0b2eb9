#!/bin/bash
# GPU box: round-2 GEMM A/B -- p12 persistent (13) vs the gemm_vit.hip kernel (20 + 4*opath + store), tile walks, and the
# FETCH_SIZE of the interesting ones.  Output: gpurun_out/r02_ab/*.txt
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02_ab; mkdir -p $OUT
B=${B:-16}
AB_STREAM=fp16 python tools/gemm_ab.py $B 13:0 20:0 24:0 22:0 26:0 21:0 > $OUT/ab_main.txt 2>&1
AB_STREAM=fp16 python tools/gemm_ab.py $B 20:0 20:256 20:512 20:768 20:1024 20:1280 20:4 20:8 24:4 > $OUT/ab_walk.txt 2>&1
cat $OUT/ab_main.txt $OUT/ab_walk.txt
M=$((80*197*B))
for v in 13:0 20:0 22:0 20:768 24:0; do
  for shape in "2304 768 plain" "768 768 res16"; do
    tag=$(echo "${v}_${shape}" | tr ' :' '__')
    AB_VARIANT=$v PMC_MATCH=gemm PMC_GROUPS="4" bash tools/pmc_gemm.sh r02ab_$tag $M $shape > /dev/null 2>&1
    echo "== $v $shape"; grep -E "TCC_EA0_RDREQ|TCC_HIT|TCC_MISS|GRBM" gpurun_out/pmc_r02ab_$tag/summary.txt
  done
done > $OUT/pmc_l2.txt 2>&1
cat $OUT/pmc_l2.txt
