#!/bin/bash
# r05 session 3: the one-wave-per-SIMD GEMM (csrc/gemm_vit1w.hip, forced variants 40 / 42): correctness, stand-alone and in-situ A/B;
# the numerics modes against the multi-episode reference goldens; the default bench run with its new cfg3 / cfg4 / fp16 legs.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s3
mkdir -p $O
export CFSAR_DEV_LIB=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "forced_variants_dev and (40 or 42)" > $O/pytest_variants.log 2>&1
tail -3 $O/pytest_variants.log
timeout 900 python tools/gemm_ab.py 16 30:0 42:0 40:0 30:4 42:4 30:16 42:16 > $O/gemm_ab.log 2>&1
cat $O/gemm_ab.log
for arm in base:-1,-1 w1short:15,-1 w1all:5,-1; do
  name=${arm%%:*}; paths=${arm#*:}
  CFSAR_DEV_VIT_PATHS=$paths timeout 600 python bench.py --steps 10 --no-cpu-baseline --no-fp16-leg --no-config-legs > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["parity"]["max_abs_dlogits"])
except Exception as e:
    print("$name failed", e); print(open("$O/bench_$name.err").read()[-2000:])
PY
done
unset CFSAR_DEV_LIB
timeout 1200 python tools/parity_multi.py > $O/parity_multi.log 2>&1
cat $O/parity_multi.log
cp gpurun_out/parity_multi.json $O/ 2>/dev/null
timeout 900 python bench.py --steps 10 > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json
