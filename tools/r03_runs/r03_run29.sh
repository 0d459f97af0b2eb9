cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/rn50prof6
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python bench.py --config rn50 --no-cpu-baseline --no-kernel-events --steps 4 --warmup 2 > $OUT/trace.log 2>&1
python tools/trace_summary.py $OUT/trace/t_kernel_trace.csv 0 > $OUT/kernel_summary.txt
rm -rf $OUT/trace
head -40 $OUT/kernel_summary.txt
