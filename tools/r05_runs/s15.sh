#!/bin/bash
# r05 session 15: the harness's batch rule (utils/batching.py: 36 cfg2 episodes per call) on the GPU: suite, smoke, the bench command's profiles at
# the new default batch (bf16 and fp16), then the default bench line twice.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s15; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "rc $?" >> $O/pytest_all.log; tail -4 $O/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
COMMIT=$1 timeout 1500 bash tools/collect_profiles.sh r05 > gpurun_out/collect_r05.log 2>&1; tail -2 gpurun_out/collect_r05.log | cut -c1-200
CMD_EXTRA="--precision fp16" COMMIT=$1 timeout 1500 bash tools/collect_profiles.sh r05_fp16pmc > gpurun_out/collect_r05_fp16.log 2>&1; tail -2 gpurun_out/collect_r05_fp16.log | cut -c1-200
for i in 1 2; do
/usr/bin/time -v -o $O/bench_$i.time timeout 1200 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err; python - <<PY
import json
d = json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["episodes_per_step_per_gpu"], d["roofline"]["frac"], d["roofline"]["traffic"], d["fp16_mode"]["value"], {k: v["value"] for k, v in d["configs"].items()})
PY
grep "Elapsed" $O/bench_$i.time
done
