#!/bin/bash
# r05 session 1: the two-workgroups-per-CU GEMM (csrc/gemm_vit4.hip) -- correctness of the forced variants (36 / 38), stand-alone A/B
# against the 8-wave kernel on the four ViT-B shapes, ablations, in-situ bench A/B (developer library on both arms).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp CFSAR_DEV_LIB=1
O=gpurun_out/r05_s1
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "forced_variants_dev and (36 or 38)" > $O/pytest_variants.log 2>&1
tail -5 $O/pytest_variants.log
timeout 900 python tools/gemm_ab.py 16 30:0 28:0 20:0 36:0 38:0 > $O/gemm_ab.log 2>&1
cat $O/gemm_ab.log
timeout 900 python tools/gemm_ab.py 16 30:4 38:4 30:8 38:8 30:16 38:16 > $O/gemm_ab_abl.log 2>&1
cat $O/gemm_ab_abl.log
for arm in base:-1,-1 wg4short:14,-1 wg4all:4,-1; do
  name=${arm%%:*}; paths=${arm#*:}
  CFSAR_DEV_VIT_PATHS=$paths timeout 600 python bench.py --steps 10 --no-cpu-baseline --no-fp16-leg > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["parity"]["max_abs_dlogits"])
except Exception as e:
    print("$name failed", e); print(open("$O/bench_$name.err").read()[-2000:])
PY
done
