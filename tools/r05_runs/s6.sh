#!/bin/bash
# r05 session 6: cfsar_frame_gemm stand-alone; the fp16 mode at one episode per call; the wide c_proj on the LDS-DMA path (its register-staged
# instance carries 45 spills).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s6; mkdir -p $O
timeout 600 python tools/frame_gemm_time.py 2>&1 | grep -v amdgpu.ids | tee $O/frame_gemm_time.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "frame_gemm" 2>&1 | tail -2
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config-legs --no-fp16-leg"
timeout 600 $B --precision fp16 --episodes-per-step 1 --steps 200 --warmup 20 > $O/bench_fp16_b1.json 2> $O/bench_fp16_b1.err
export CFSAR_DEV_LIB=1
for arm in base:-1,-1 dma:2,-1; do
  name=${arm%%:*}; paths=${arm#*:}
  CFSAR_DEV_VIT_PATHS=$paths timeout 600 $B --precision fp16 > $O/bench_fp16_$name.json 2> $O/bench_fp16_$name.err
done
python - <<PY
import json
for n in ("bench_fp16_b1", "bench_fp16_base", "bench_fp16_dma"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("parity") or {}).get("max_abs_dlogits"))
    except Exception as e:
        print(n, "failed", e); print(open("$O/%s.err" % n).read()[-1500:])
PY
