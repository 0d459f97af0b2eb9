#!/bin/bash
# r05 session 27: RN50 tower at the harness's new batch (32 episodes per call): bench lines (bf16 / fp16), harness and RN50 GPU tests.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s27; mkdir -p $O
timeout 1200 python -m pytest tests/test_harness.py tests/test_gpu_e2e.py tests/test_bench_contract.py -q -m gpu -k "rn50 or rn or reference_shaped or config" > $O/pytest_rn.log 2>&1; tail -3 $O/pytest_rn.log
timeout 600 python bench.py --config rn50 --no-cpu-baseline > $O/bench_rn50.json 2> $O/bench_rn50.err
timeout 600 python bench.py --config rn50 --no-cpu-baseline --precision fp16 > $O/bench_rn50_fp16.json 2> $O/bench_rn50_fp16.err
timeout 600 python bench.py --config rn50 --no-cpu-baseline --episodes-per-step 36 --steps 10 > $O/bench_rn50_b36.json 2> $O/bench_rn50_b36.err
python - <<PY
import json
for n in ("bench_rn50", "bench_rn50_fp16", "bench_rn50_b36"):
    d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["config"]["episodes_per_step_per_gpu"], d["roofline"]["frac"], d["roofline"].get("frac_end_to_end"), d.get("fp16_mode", {}).get("value"))
PY
