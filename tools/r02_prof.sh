#!/bin/bash
# GPU box: rocprofv3 kernel trace of the bench command -> gpurun_out/prof_<tag>/kernel_summary.txt   usage: tools/r02_prof.sh <tag> [env assignments / bench flags via CMD_EXTRA]
set -u
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-fp16-leg ${CMD_EXTRA:-} > $OUT/trace.log 2>&1
python tools/trace_summary.py $OUT/trace/t_kernel_trace.csv 0 > $OUT/kernel_summary.txt
cp $OUT/trace/t_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
cat $OUT/kernel_summary.txt
