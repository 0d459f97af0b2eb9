#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s9; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-leg"
for eps in 16 24 32 16 32; do timeout 300 $B --episodes-per-step $eps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 B=$eps', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
for c in cfg3 cfg4; do for p in bf16 fp16; do timeout 600 $B --config $c --precision $p --episodes-per-step 8 > $O/bench_${c}_$p.json 2>/dev/null; python -c "
import json
d=json.loads(open('$O/bench_${c}_$p.json').read().strip().splitlines()[-1]); print('$c $p', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"; done; done
timeout 300 $B --precision fp32 > $O/bench_fp32.json 2>/dev/null; python -c "
import json
d=json.loads(open('$O/bench_fp32.json').read().strip().splitlines()[-1]); print('fp32', d['value'], d['parity']['max_abs_dlogits'])"
timeout 300 $B --episodes-per-step 1 > $O/bench_b1.json 2>/dev/null; python -c "
import json
d=json.loads(open('$O/bench_b1.json').read().strip().splitlines()[-1]); print('B=1', d['value'])"
timeout 300 $B --config rn50 > $O/bench_rn50.json 2>/dev/null; python -c "
import json
d=json.loads(open('$O/bench_rn50.json').read().strip().splitlines()[-1]); print('rn50', d['value'])"
