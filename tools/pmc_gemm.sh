#!/bin/bash
# GPU box dev tool: PMC passes over ONE bf16 GEMM shape (tools/gemm_only.py) -> gpurun_out/pmc_<tag>/summary.txt
# usage: tools/pmc_gemm.sh <tag> M N K [f32]    (env CFSAR_GEMM_VARIANT / CFSAR_GEMM_DEBUG pass through)
#        PMC_CMD='python tools/attn_only.py' PMC_MATCH=vit_attn tools/pmc_gemm.sh <tag> <args of the command>
#        PMC_GROUPS='1 2' limits the counter groups (each group is one full run of the command)
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_TOTAL_ACCESSES TA_BUSY TCP_TCR_TCP_STALL_CYCLES TCP_RFIFO_STALL_CYCLES TCP_LFIFO_STALL_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  case " ${PMC_GROUPS:-1 2 3 4} " in *" $i "*) ;; *) continue;; esac
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o p -- ${PMC_CMD:-python tools/gemm_only.py} "$@" > $OUT/g$i.log 2>&1
done
python - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/g*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "${PMC_MATCH:-gemm_kernel}" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-34s %16.0f  (avg of %d launches)" % (k, sum(v) / len(v), len(v)))
PY
cat $OUT/summary.txt
