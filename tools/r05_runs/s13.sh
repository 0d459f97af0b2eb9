#!/bin/bash
# r05 session 13: run-to-run spread of the headline on ONE box (the round-over-round signal has to be read against it): 6 default-length runs.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s13; mkdir -p $O
for i in 1 2 3 4 5 6; do
  timeout 600 python bench.py --no-cpu-baseline --no-fp16-leg --no-config-legs > $O/bench_$i.json 2> $O/bench_$i.err
done
python - <<PY | tee $O/spread.txt
import json, statistics
v = [json.loads(open("$O/bench_%d.json" % i).read().strip().splitlines()[-1]) for i in range(1, 7)]
vals = [d["value"] for d in v]; fr = [d["roofline"]["frac"] for d in v]
print("episodes/s (40 timed steps each):", vals)
print("mean %.2f  min %.2f  max %.2f  spread %.2f %%  GEMM frac %.4f-%.4f" % (statistics.mean(vals), min(vals), max(vals), 100 * (max(vals) - min(vals)) / statistics.mean(vals), min(fr), max(fr)))
PY
