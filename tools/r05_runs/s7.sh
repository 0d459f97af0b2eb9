#!/bin/bash
# r05 session 7: the fp16 mode's wide c_proj (K = 3 072) on the LDS-DMA path against the register-staged instance (45 spilled registers): bench A/B,
# four alternations, developer library; the hybrid policy of the per-frame GEMMs (product library) at 16 and at 1 episode per call.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s7; mkdir -p $O
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-config-legs --no-fp16-leg --precision fp16"
timeout 600 $B > $O/bench_fp16_prod.json 2> $O/bench_fp16_prod.err
timeout 600 $B --episodes-per-step 1 --steps 200 --warmup 20 > $O/bench_fp16_b1.json 2> $O/bench_fp16_b1.err
export CFSAR_DEV_LIB=1
for rep in 1 2 3 4; do
for arm in base:-1,-1 dma:2,-1; do
  name=${arm%%:*}; paths=${arm#*:}
  CFSAR_DEV_VIT_PATHS=$paths timeout 600 $B > $O/bench_fp16_${name}_$rep.json 2> $O/bench_fp16_${name}_$rep.err
done
done
python - <<PY
import json, glob
for n in sorted(glob.glob("$O/bench_fp16_*.json")):
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        print(n.split("/")[-1], d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("parity") or {}).get("max_abs_dlogits"))
    except Exception as e:
        print(n, "failed", e)
PY
