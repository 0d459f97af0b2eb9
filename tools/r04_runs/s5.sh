#!/bin/bash
# session 5: per-frame low-word correction (mcorr): kernel tests, parity per variant, cost per variant, operand path of the wide kernels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "residual_wide or lnfold_split or lnfold_hp or copy_rows or frame_col" > $O/pytest_new.log 2>&1; echo "rc $?" >> $O/pytest_new.log
tail -6 $O/pytest_new.log
VARIANT_FILTER="mcorr|split qkv,out,pr|wide+lo" timeout 1500 python tools/fp16_variants.py cfg2_B16_5w1s_T8 cfg3_B16_5w5s_T8_mb cfg4_L14_5w1s_T16 t197_5w1s_T2 > $O/variants.txt 2>&1; cp gpurun_out/fp16_variants.json $O/ 2>/dev/null
tail -40 $O/variants.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-leg --precision fp16"
run() { tag=$1; shift; env "$@" timeout 300 $B > $O/bench_$tag.json 2> $O/bench_$tag.err; python - $O/bench_$tag.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("parity",{}).get("max_abs_dlogits"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
run split3 CFSAR_FP16_SPLIT=qkv,out,pr CFSAR_FP16_MCORR=
run widelo CFSAR_FP16_SPLIT= CFSAR_FP16_MCORR=
run mcorr_all CFSAR_FP16_SPLIT= CFSAR_FP16_MCORR=qkv,out,fc,pr
run mcorr3 CFSAR_FP16_SPLIT= CFSAR_FP16_MCORR=qkv,out,pr
run out_mcorr3 CFSAR_FP16_SPLIT=out CFSAR_FP16_MCORR=qkv,fc,pr
run out_mcorr2 CFSAR_FP16_SPLIT=out CFSAR_FP16_MCORR=qkv,pr
# operand path of the fp16-mode kernels (developer library): policy vs LDS-DMA everywhere vs register-staged everywhere
run dev_policy CFSAR_DEV_LIB=1 CFSAR_FP16_SPLIT=out CFSAR_FP16_MCORR=qkv,fc,pr
run dev_opath2 CFSAR_DEV_LIB=1 CFSAR_DEV_VIT_PATHS=2,-1 CFSAR_FP16_SPLIT=out CFSAR_FP16_MCORR=qkv,fc,pr
run dev_opath0 CFSAR_DEV_LIB=1 CFSAR_DEV_VIT_PATHS=0,-1 CFSAR_FP16_SPLIT=out CFSAR_FP16_MCORR=qkv,fc,pr
# where the time goes in the candidate
cd /tmp; CFSAR_FP16_SPLIT=out CFSAR_FP16_MCORR=qkv,fc,pr rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-fp16-leg --precision fp16" > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/trace_summary.py $O/trace/t_kernel_trace.csv 0 > $O/kernel_summary_out_mcorr3.txt; rm -rf $O/trace; head -16 $O/kernel_summary_out_mcorr3.txt
