#!/bin/bash
# r05 session 9: wall time of the default bench run (headline + fp16 leg + cfg3 / cfg4 legs + CPU baseline); episodes-per-step sweep for the record.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s9; mkdir -p $O
t0=$(date +%s); timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; t1=$(date +%s); echo "default bench.py wall time: $((t1 - t0)) s"
for b in 16 24 32; do
  timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-config-legs --no-fp16-leg --episodes-per-step $b > $O/bench_b$b.json 2> $O/bench_b$b.err
done
python - <<PY
import json
for n in ("default", "b16", "b24", "b32"):
    d = json.loads(open("$O/bench_%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["episodes_per_step_per_gpu"])
PY
