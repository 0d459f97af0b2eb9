#!/bin/bash
# session 4: round-4 profiles of the bench command (bf16 default): kernel trace + stats, FETCH / WRITE, MFMA / LDS / TCC PMC passes;
# kernel trace of the fp16 mode
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
COMMIT=$1 bash tools/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1
tail -30 gpurun_out/collect_r04.log
O=gpurun_out/prof_r04_fp16; mkdir -p $O
CMD="python bench.py --steps 4 --warmup 2 --episodes-per-step 16 --no-cpu-baseline --no-kernel-events --no-fp16-leg --precision fp16"
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- bash -c "cd $GRAFT_REPO_ROOT && $CMD" > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_summary.py $O/trace/t_kernel_trace.csv 0 > $O/kernel_summary.txt; rm -rf $O/trace
head -20 $O/kernel_summary.txt
