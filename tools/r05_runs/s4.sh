#!/bin/bash
# r05 session 4: the two-workgroups-per-CU form on the RESIDUAL launches alone (their epilogue is memory-bound: stream read + write,
# two words in the fp16 mode) -- bench A/B in the fp16 and the bf16 mode; multi-episode parity of the remaining goldens.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s4
mkdir -p $O
export CFSAR_DEV_LIB=1
for prec in fp16 bf16; do
for arm in base:-1,-1:0 g2out:14,-1:0x200000 g2res:4,-1:0x200000 g3res:5,-1:0x200000; do
  name=${arm%%:*}; rest=${arm#*:}; paths=${rest%%:*}; dbg=${rest#*:}
  CFSAR_DEV_VIT_PATHS=$paths CFSAR_DEV_VIT_DBG=$dbg timeout 600 python bench.py --precision $prec --steps 10 --no-cpu-baseline --no-fp16-leg --no-config-legs > $O/bench_${prec}_$name.json 2> $O/bench_${prec}_$name.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_${prec}_$name.json").read().strip().splitlines()[-1])
    print("$prec $name", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["parity"]["max_abs_dlogits"])
except Exception as e:
    print("$prec $name failed", e); print(open("$O/bench_${prec}_$name.err").read()[-1500:])
PY
done
done
unset CFSAR_DEV_LIB
timeout 1500 python tools/parity_multi.py hc_cfg4_L14_5w1s_T16 > $O/parity_multi_cfg4.log 2>&1
cat $O/parity_multi_cfg4.log
cp gpurun_out/parity_multi.json $O/parity_multi_cfg4.json 2>/dev/null
