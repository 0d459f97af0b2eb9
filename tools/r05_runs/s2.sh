#!/bin/bash
# r05 session 2: evidence for the two-workgroups-per-CU GEMM's A/B (tools/r05_runs/s1.sh): are both workgroups resident (per-tile
# time stamps of all 512), what do the counters say (MFMA busy, waits, L2 hit) next to the 8-wave kernel, does a priority split help.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp CFSAR_DEV_LIB=1
O=gpurun_out/r05_s2
mkdir -p $O
for sh in qkv fc; do
  timeout 300 python tools/vit_trace.py 16 $sh 30 0 > $O/trace_${sh}_w8.log 2>&1
  TRACE_GRID=512 timeout 300 python tools/vit_trace.py 16 $sh 38 0 > $O/trace_${sh}_w4.log 2>&1
  tail -2 $O/trace_${sh}_w8.log; tail -2 $O/trace_${sh}_w4.log
done
AB_SHAPES=qkv,fc timeout 600 python tools/gemm_ab.py 16 30:0 38:0 38:524288 > $O/gemm_ab_prio.log 2>&1
cat $O/gemm_ab_prio.log
for v in 30 38; do
  AB_VARIANT=$v PMC_MATCH=vit_gemm PMC_GROUPS='1 4' timeout 600 bash tools/pmc_gemm.sh r05_qkv_v$v 252160 2304 768 plain > $O/pmc_qkv_v$v.log 2>&1
  AB_VARIANT=$v PMC_MATCH=vit_gemm PMC_GROUPS='1 4' timeout 600 bash tools/pmc_gemm.sh r05_fc_v$v 252160 3072 768 gelu > $O/pmc_fc_v$v.log 2>&1
  cat gpurun_out/pmc_r05_qkv_v$v/summary.txt gpurun_out/pmc_r05_fc_v$v/summary.txt
done
