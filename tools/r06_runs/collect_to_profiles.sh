#!/bin/bash
# dev container: copy what a tools/r06_runs/s6.sh (or s8.sh) session merged into gpurun_out/ to the tracked profiles/r06_* names.   usage: bash tools/r06_runs/collect_to_profiles.sh <session dir, e.g. gpurun_out/r06_s6>
S=${1:-gpurun_out/r06_s6}
last() { tail -n 1 "$1" | grep -q '^{' && tail -n 1 "$1" > "$2"; }
last $S/bench_with_traffic.json profiles/r06_bench.json
last $S/bench.json profiles/r06_bench_with_cpu_baseline.json
last $S/bench_fp16.json profiles/r06_bench_fp16.json
last $S/bench_strict.json profiles/r06_bench_fp16_strict.json
last $S/bench_fp32.json profiles/r06_bench_fp32.json
last $S/bench_rn50.json profiles/r06_bench_rn50.json
last $S/bench_rn50_fp16.json profiles/r06_bench_rn50_fp16.json
last $S/bench_b1.json profiles/r06_bench_b1.json
grep -v amdgpu.ids $S/parity_multi.log > profiles/r06_parity_multi.log; [ -f $S/parity_multi.json ] && cp $S/parity_multi.json profiles/r06_parity_multi.json
grep -v amdgpu.ids $S/strict_eval.log > profiles/r06_strict_fresh64.log; cp $S/strict_eval_64ep.json profiles/r06_strict_fresh64.json
grep -v amdgpu.ids $S/strict_eval_hc.log > profiles/r06_strict_fresh64_high_contrast.log; cp $S/strict_eval_hc_64ep.json profiles/r06_strict_fresh64_high_contrast.json
tail -6 $S/pytest_all.log | grep -v "^$\|Docs:" > profiles/r06_gpu_suite_tail.txt; tail -3 $S/smoke.log >> profiles/r06_gpu_suite_tail.txt
P=gpurun_out/prof_r06
cp $P/kernel_summary.txt profiles/r06_bench_kernel_summary.txt; cp $P/kernel_stats.csv profiles/r06_bench_kernel_stats.csv
cp $P/gemm_traffic.json profiles/r06_gemm_traffic.json; cp $P/pmc_by_kernel.json profiles/r06_pmc_by_kernel.json
cp gpurun_out/prof_r06_fp16/kernel_summary.txt profiles/r06_bench_fp16_kernel_summary.txt
cp gpurun_out/prof_r06_fp16_strict/kernel_summary.txt profiles/r06_bench_fp16_strict_kernel_summary.txt
ls -la profiles/r06_* | wc -l
