#!/bin/bash
# r05 session 26: RN50 tower, episodes per step sweep (the harness's default for this tower is 16).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s26; mkdir -p $O
for b in 16 24 16 32 36 24; do
  timeout 600 python bench.py --config rn50 --steps 12 --warmup 3 --no-cpu-baseline --no-config-legs --no-fp16-leg --episodes-per-step $b > $O/rn50_b${b}_$RANDOM.json 2>> $O/err.log
done
python - <<PY | tee $O/sweep.txt
import json, glob
for n in sorted(glob.glob("$O/rn50_*.json")):
    d = json.loads(open(n).read().strip().splitlines()[-1])
    print(d["config"]["episodes_per_step_per_gpu"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["parity"].get("max_abs_dlogits"))
PY
