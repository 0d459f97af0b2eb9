#!/bin/bash
# session 12: final round-4 evidence on one box: GPU suite, smoke, bench lines, 16-episode parity statistics of the product setting, profiles
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s12; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "rc $?" >> $O/pytest_all.log; tail -4 $O/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
timeout 600 python bench.py --precision fp16 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err; tail -c 200 $O/bench_fp16.json
VARIANT_FILTER="wide+lo+mcorr all|r3" timeout 2400 python tools/fp16_variants.py --episodes 16 cfg2_B16_5w1s_T8 cfg3_B16_5w5s_T8_mb cfg4_L14_5w1s_T16 2>&1 | grep -v amdgpu.ids | tail -7; cp gpurun_out/fp16_variants_16ep.json $O/
timeout 1500 python tools/parity_report.py > $O/parity_table.md 2> $O/parity.err; cp gpurun_out/parity_table.json $O/; cat $O/parity_table.md
COMMIT=$1 bash tools/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1; tail -3 gpurun_out/collect_r04.log | cut -c1-200
P=gpurun_out/prof_r04_fp16; mkdir -p $P
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$P/trace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 4 --warmup 2 --episodes-per-step 16 --no-cpu-baseline --no-kernel-events --no-fp16-leg --precision fp16" > $GRAFT_REPO_ROOT/$P/trace.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/trace_summary.py $P/trace/t_kernel_trace.csv 0 > $P/kernel_summary.txt; rm -rf $P/trace; head -12 $P/kernel_summary.txt
