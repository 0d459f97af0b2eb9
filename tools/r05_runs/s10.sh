#!/bin/bash
# r05 session 10: counters for the attention kernel next to the copy bandwidth of the same box; counters of the fp16 mode's kernels.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s10; mkdir -p $O
python - > $O/copy_bw.log 2>&1 <<'PY'
import torch, statistics
def t(fn, n=20):
    for _ in range(3): fn()
    ts=[]
    for _ in range(5):
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e)/n*1e3)
    return statistics.median(ts)
a=torch.empty(1280*197*2304, device="cuda", dtype=torch.bfloat16).normal_()      # 1.16 GB, the qkv matrix
b=torch.empty(1280*197*768, device="cuda", dtype=torch.bfloat16)                  # 0.39 GB, the attention output
c=torch.empty_like(a)
us=t(lambda: c.copy_(a)); print("copy 1.16 GB -> 1.16 GB: %.1f us, %.2f TB/s (read + write)" % (us, 2*a.numel()*2/us/1e6))
us=t(lambda: b.copy_(a[:b.numel()])); print("copy 0.39 GB -> 0.39 GB: %.1f us, %.2f TB/s" % (us, 2*b.numel()*2/us/1e6))
us=t(lambda: torch.sum(a.view(torch.int16)[:a.numel()//2*2].view(torch.int32), dtype=torch.int64)); print("read-only 1.16 GB (int32 sum): %.1f us, %.2f TB/s" % (us, a.numel()*2/us/1e6))
PY
cat $O/copy_bw.log | grep -v amdgpu.ids
timeout 300 python tools/attn_time.py 1280 2>&1 | grep -v amdgpu.ids | tee $O/attn_time.log
ATTN_VARIANT=256 timeout 300 python tools/attn_time.py 1280 2>&1 | grep -v amdgpu.ids | sed 's/^/memory pipeline only (no tile computation): /' | tee -a $O/attn_time.log
OUT=$O/pmc; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  cd /tmp; timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/g$i -o p -- bash -c "cd $GRAFT_REPO_ROOT && python tools/attn_only.py 1280" > $GRAFT_REPO_ROOT/$OUT/g$i.log 2>&1; cd $GRAFT_REPO_ROOT
done
python - <<PY | tee $O/attn_pmc_summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/g*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "vit_attn" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-30s %16.0f  (avg of %d launches)" % (k, sum(v) / len(v), len(v)))
PY
rm -rf $OUT/g*/p_agent_info.csv
CMD_EXTRA="--precision fp16" COMMIT=$1 bash tools/collect_profiles.sh r05_fp16pmc > gpurun_out/collect_r05_fp16.log 2>&1; tail -2 gpurun_out/collect_r05_fp16.log | cut -c1-200
ls gpurun_out/prof_r05_fp16pmc/
