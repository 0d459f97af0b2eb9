#!/bin/bash
# r06 session 6: round-6 evidence on one box: GPU suite, smoke, the default bench line (headline + fp16 + strict legs + cfg3 / cfg4 legs), the other bench lines,
# every 16-bit mode against the six multi-episode reference sets and on 64 fresh episodes (standard and high contrast), the profiles the roofline object cites.
# usage: bash tools/r06_runs/s6.sh <commit>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_s6; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "rc $?" >> $O/pytest_all.log; tail -4 $O/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config-legs"
timeout 600 $B --precision fp16 --no-fp16-leg > $O/bench_fp16.json 2> $O/bench_fp16.err
timeout 600 $B --precision fp16_strict --no-fp16-leg > $O/bench_strict.json 2> $O/bench_strict.err
timeout 600 $B --precision fp32 --no-fp16-leg > $O/bench_fp32.json 2> $O/bench_fp32.err
timeout 600 $B --config rn50 > $O/bench_rn50.json 2>$O/bench_rn50.err
timeout 600 $B --config rn50 --precision fp16 --no-fp16-leg > $O/bench_rn50_fp16.json 2>$O/bench_rn50_fp16.err
timeout 600 $B --episodes-per-step 1 --steps 200 --warmup 20 > $O/bench_b1.json 2>$O/bench_b1.err
python - <<PY
import json
for n in ("bench", "bench_fp16", "bench_strict", "bench_fp32", "bench_rn50", "bench_rn50_fp16", "bench_b1"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic"), (d.get("fp16_mode") or {}).get("value"), (d.get("strict_mode") or {}).get("value"), (d.get("parity") or {}).get("max_abs_dlogits"))
    except Exception as e:
        print(n, "failed", e)
PY
timeout 2400 python tools/parity_multi.py --modes "fp32|fp16_strict|fp16|bf16" > $O/parity_multi.log 2>&1; grep -v amdgpu.ids $O/parity_multi.log | cut -c1-240; cp gpurun_out/parity_multi.json $O/
timeout 2400 python tools/strict_eval.py --episodes 64 fp16 fp16_strict > $O/strict_eval.log 2>&1; grep -v amdgpu.ids $O/strict_eval.log | cut -c1-240; cp gpurun_out/strict_eval_64ep.json $O/strict_eval_64ep.json
timeout 2400 python tools/strict_eval.py --episodes 64 --lowfreq 2.0 fp16 fp16_strict > $O/strict_eval_hc.log 2>&1; grep -v amdgpu.ids $O/strict_eval_hc.log | cut -c1-240; cp gpurun_out/strict_eval_64ep.json $O/strict_eval_hc_64ep.json
COMMIT=$1 bash tools/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1; tail -3 gpurun_out/collect_r06.log | cut -c1-200
for prec in fp16 fp16_strict; do
  P=gpurun_out/prof_r06_$prec; mkdir -p $P
  cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$P/trace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-fp16-leg --no-config-legs --precision $prec" > $GRAFT_REPO_ROOT/$P/trace.log 2>&1
  cd $GRAFT_REPO_ROOT; python tools/trace_summary.py $P/trace/t_kernel_trace.csv 0 > $P/kernel_summary.txt; rm -rf $P/trace; head -8 $P/kernel_summary.txt | cut -c1-150
done
cp gpurun_out/prof_r06/gemm_traffic.json profiles/r06_gemm_traffic.json      # (on the box: the line below then quotes the traffic of THIS build)
timeout 900 python bench.py --steps 20 --no-cpu-baseline > $O/bench_with_traffic.json 2> $O/bench_with_traffic.err; tail -c 200 $O/bench_with_traffic.json
