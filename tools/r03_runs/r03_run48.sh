cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cfg4prof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/cfg4prof/t -o t -- python bench.py --config cfg4 --no-cpu-baseline --no-kernel-events --no-fp16-leg --steps 3 --warmup 1 > gpurun_out/cfg4prof/log 2>&1
python tools/trace_summary.py gpurun_out/cfg4prof/t/t_kernel_trace.csv 0 | head -22 | cut -c1-210
rm -rf gpurun_out/cfg4prof/t
