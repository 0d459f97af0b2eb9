#!/bin/bash
# session 10: round-4 evidence set on one box: GPU suite, smoke, default bench line (+ cpu baseline), fp16 line, profiles of the bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s10; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "rc $?" >> $O/pytest_all.log; tail -4 $O/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
timeout 600 python bench.py --precision fp16 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err; tail -c 300 $O/bench_fp16.json
timeout 600 python bench.py --inputs host --no-cpu-baseline --no-fp16-leg > $O/bench_host.json 2> $O/bench_host.err
COMMIT=$1 bash tools/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1; tail -12 gpurun_out/collect_r04.log | cut -c1-200
P=gpurun_out/prof_r04_fp16; mkdir -p $P
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$P/trace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 4 --warmup 2 --episodes-per-step 16 --no-cpu-baseline --no-kernel-events --no-fp16-leg --precision fp16" > $GRAFT_REPO_ROOT/$P/trace.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/trace_summary.py $P/trace/t_kernel_trace.csv 0 > $P/kernel_summary.txt; rm -rf $P/trace; head -14 $P/kernel_summary.txt
