#!/bin/bash
# second stream word as one e5m2 byte: kernel tests, fp16 e2e tests, bench (bf16 with fp16 leg, fp16 headline), 16-episode statistics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s20; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "wide or pair or lnfold or correction" 2>&1 | tail -6
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "fp16 or cfg2_full or cfg3_cfg4 or outlier or pruning" 2>&1 | tail -6
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --precision fp16 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err
python - <<PY
import json
for n in ("bench", "bench_fp16"):
    d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("fp16_mode") or {}).get("value"), (d.get("fp16_mode") or {}).get("parity"), (d.get("parity") or {}).get("max_abs_dlogits"))
PY
VARIANT_FILTER="wide+lo+mcorr all" timeout 2400 python tools/fp16_variants.py --episodes 16 cfg2_B16_5w1s_T8 cfg3_B16_5w5s_T8_mb cfg4_L14_5w1s_T16 2>&1 | grep -v amdgpu.ids | tail -4; cp gpurun_out/fp16_variants_16ep.json $O/
