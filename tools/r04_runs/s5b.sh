#!/bin/bash
# session 5b: cost per fp16-mode variant, operand path of the wide kernels, kernel summary of the candidate
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s5; mkdir -p $O
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
