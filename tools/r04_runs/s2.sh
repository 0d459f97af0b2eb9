#!/bin/bash
# session 2: the fp16 numerics mode's new kernels (wide residual, split weights): kernel tests, parity per variant, cost per variant
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "residual_wide or lnfold_split or copy_rows" > $O/pytest_new.log 2>&1; echo "rc $?" >> $O/pytest_new.log
tail -5 $O/pytest_new.log
timeout 1200 python tools/fp16_variants.py > $O/variants.txt 2>&1; cp gpurun_out/fp16_variants.json $O/ 2>/dev/null
tail -60 $O/variants.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-leg --precision fp16"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("parity",{}).get("max_abs_dlogits"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
run r3 CFSAR_FP16_WIDE=0 CFSAR_FP16_SPLIT=
run wide CFSAR_FP16_WIDE=1 CFSAR_FP16_LO=0 CFSAR_FP16_SPLIT=
run widelo CFSAR_FP16_WIDE=1 CFSAR_FP16_LO=1 CFSAR_FP16_SPLIT=
run widelo_out CFSAR_FP16_WIDE=1 CFSAR_FP16_LO=1 CFSAR_FP16_SPLIT=out
run widelo_qkvout CFSAR_FP16_WIDE=1 CFSAR_FP16_LO=1 CFSAR_FP16_SPLIT=qkv,out
run wide_qkvout CFSAR_FP16_WIDE=1 CFSAR_FP16_LO=0 CFSAR_FP16_SPLIT=qkv,out
run widelo_all CFSAR_FP16_WIDE=1 CFSAR_FP16_LO=1 CFSAR_FP16_SPLIT=qkv,out,fc,pr
timeout 600 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "rc $?" >> $O/pytest_all.log; tail -5 $O/pytest_all.log
