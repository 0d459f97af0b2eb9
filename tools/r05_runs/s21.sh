#!/bin/bash
# r05 session 21: the tree with cfsar_patch_embed (ABI 8): GPU suite, smoke, the bench command's profiles (bf16 and fp16) for this build, the default bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s21; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "rc $?" >> $O/pytest_all.log; tail -4 $O/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
COMMIT=$1 timeout 1500 bash tools/collect_profiles.sh r05 > gpurun_out/collect_r05.log 2>&1
CMD_EXTRA="--precision fp16" COMMIT=$1 timeout 1500 bash tools/collect_profiles.sh r05_fp16pmc > gpurun_out/collect_r05_fp16.log 2>&1
cp gpurun_out/prof_r05/gemm_traffic.json profiles/r05_gemm_traffic.json      # (on the box: the line below then quotes the traffic of THIS build)
t0=$SECONDS
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["episodes_per_step_per_gpu"], d["roofline"]["frac"], d["roofline"]["traffic"], d["fp16_mode"]["value"], {k: v["value"] for k, v in d["configs"].items()})
PY
echo "bench wall time $((SECONDS - t0)) s" | tee -a $O/bench.err
