#!/bin/bash
# r05 session 17: the default bench line at the harness's batch (36 episodes per step), twice, with wall time.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s17; mkdir -p $O
for i in 1 2; do
t0=$SECONDS
timeout 1200 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err; python - <<PY
import json
d = json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["episodes_per_step_per_gpu"], d["roofline"]["frac"], d["roofline"]["traffic"], d["fp16_mode"]["value"], {k: v["value"] for k, v in d["configs"].items()})
PY
echo "bench wall time $((SECONDS - t0)) s" | tee -a $O/bench_$i.err
done
