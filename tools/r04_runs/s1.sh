#!/bin/bash
# session 1: baseline checks + the band-chunked layer schedule A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s1; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-leg"
timeout 300 $B > $O/bench_base.json 2> $O/bench_base.err
for mode in block mlp pairs; do for cf in 80 160 320 640; do
  CFSAR_CHUNK_MODE=$mode CFSAR_CHUNK_FRAMES=$cf timeout 300 $B > $O/bench_${mode}_$cf.json 2> $O/bench_${mode}_$cf.err
done; done
timeout 300 $B > $O/bench_base2.json 2>> $O/bench_base.err
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("parity",{}).get("max_abs_dlogits"))
except Exception as e: print("ERR",e)
PY
done
tail -3 $O/pytest.log
