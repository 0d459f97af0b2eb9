#!/bin/bash
# r05 session 14: episodes per step chosen so that the persistent GEMM grid's rounds come out integral (18 / 36 episodes: 1 109 / 2 217 row bands x
# {3, 9, 12} column tiles = 13.0 / 39.0 / 52.0 rounds on 256 CUs, 99.9 % full, against 97.6 % at 16): same-box sweep, bf16 and fp16.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s14; mkdir -p $O
for prec in bf16 fp16; do
for b in 16 18 16 18 36 29; do
  timeout 600 python bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-config-legs --no-fp16-leg --precision $prec --episodes-per-step $b > $O/bench_${prec}_b${b}_$RANDOM.json 2>> $O/err.log
done
done
python - <<PY | tee $O/sweep.txt
import json, glob
for n in sorted(glob.glob("$O/bench_*.json")):
    d = json.loads(open(n).read().strip().splitlines()[-1])
    print(d["config"]["precision"], d["config"]["episodes_per_step_per_gpu"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["parity"]["max_abs_dlogits"])
PY
