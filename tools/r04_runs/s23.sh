#!/bin/bash
# c_fc (fp16 mode) with and without the fused per-frame output sums: per-kernel times
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for um in 1 0; do
  P=gpurun_out/prof_um$um; mkdir -p $P
  cd /tmp; CFSAR_FUSED_UMEANS=$um rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$P/trace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 4 --warmup 2 --episodes-per-step 16 --no-cpu-baseline --no-kernel-events --no-fp16-leg --precision fp16" > $GRAFT_REPO_ROOT/$P/trace.log 2>&1
  cd $GRAFT_REPO_ROOT; python tools/trace_summary.py $P/trace/t_kernel_trace.csv 0 > $P/kernel_summary.txt; rm -rf $P/trace; echo "== fused u means $um"; head -9 $P/kernel_summary.txt | cut -c1-140
done
