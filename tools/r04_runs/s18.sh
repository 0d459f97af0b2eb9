#!/bin/bash
# session 18: round-4 evidence on one box after the RN50 fp16 mode (gemm.hip changed -> traffic fingerprint): GPU suite, smoke, bench lines,
# parity statistics, profiles, RN50 lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s18; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "rc $?" >> $O/pytest_all.log; tail -4 $O/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
timeout 600 python bench.py --precision fp16 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err; tail -c 120 $O/bench_fp16.json
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
timeout 600 $B --config rn50 > $O/bench_rn50.json 2>$O/bench_rn50.err
timeout 600 $B --config rn50 --precision fp16 --no-fp16-leg > $O/bench_rn50_fp16.json 2>$O/bench_rn50_fp16.err
python - <<PY
import json
for n in ("bench", "bench_fp16", "bench_rn50", "bench_rn50_fp16"):
    d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic"), (d.get("fp16_mode") or {}).get("value"), (d.get("parity") or {}).get("max_abs_dlogits"))
PY
VARIANT_FILTER="wide+lo+mcorr all|r3" timeout 2400 python tools/fp16_variants.py --episodes 16 cfg2_B16_5w1s_T8 cfg3_B16_5w5s_T8_mb cfg4_L14_5w1s_T16 2>&1 | grep -v amdgpu.ids | tail -7; cp gpurun_out/fp16_variants_16ep.json $O/
timeout 1500 python tools/parity_report.py > $O/parity_table.md 2> $O/parity.err; cp gpurun_out/parity_table.json $O/; cat $O/parity_table.md
COMMIT=$1 bash tools/collect_profiles.sh r04 > gpurun_out/collect_r04.log 2>&1; tail -3 gpurun_out/collect_r04.log | cut -c1-200
for cfg in "fp16:--precision fp16" "rn50:--config rn50" "rn50_fp16:--config rn50 --precision fp16"; do
  tag=${cfg%%:*}; extra=${cfg#*:}
  P=gpurun_out/prof_r04_$tag; mkdir -p $P
  cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$P/trace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 4 --warmup 2 --episodes-per-step 16 --no-cpu-baseline --no-kernel-events --no-fp16-leg $extra" > $GRAFT_REPO_ROOT/$P/trace.log 2>&1
  cd $GRAFT_REPO_ROOT; python tools/trace_summary.py $P/trace/t_kernel_trace.csv 0 > $P/kernel_summary.txt; rm -rf $P/trace; echo "== $tag"; head -8 $P/kernel_summary.txt | cut -c1-150
done
