#!/bin/bash
# r06 session 5: ViT-L/14 at high contrast (the one place the strict defaults still pass 1e-3 on fresh episodes): what more splitting buys there and costs.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_s5; mkdir -p $O
F2="fp16_strict;CFSAR_FP16_SPLIT=qkv,fc;CFSAR_FP16_MCORR=out,pr"
F3="fp16_strict;CFSAR_FP16_SPLIT=qkv,fc,pr;CFSAR_FP16_MCORR=out"
F4="fp16_strict;CFSAR_FP16_SPLIT=qkv,out,fc,pr;CFSAR_FP16_MCORR="
timeout 2400 python tools/strict_eval.py --episodes 64 --lowfreq 2.0 --configs cfg4,cfg3 fp16_strict "$F2" "$F3" "$F4" > $O/strict_eval_hc.log 2>&1; grep -v amdgpu.ids $O/strict_eval_hc.log | cut -c1-260 | tail -10
for v in "0::" "2:qkv,fc:out,pr" "3:qkv,fc,pr:out"; do
  n=${v%%:*}; r=${v#*:}; sp=${r%%:*}; mc=${r#*:}
  if [ "$n" = "0" ]; then unset CFSAR_FP16_SPLIT CFSAR_FP16_MCORR; else export CFSAR_FP16_SPLIT="$sp" CFSAR_FP16_MCORR="$mc"; fi
  timeout 600 python bench.py --config cfg4 --steps 4 --warmup 2 --no-cpu-baseline --no-config-legs --no-fp16-leg --precision fp16_strict > $O/bench_cfg4_S$n.json 2> $O/bench_cfg4_S$n.err
  python -c "
import json
d=json.loads(open('$O/bench_cfg4_S$n.json').read().strip().splitlines()[-1]); print('cfg4 strict split=[$sp] mcorr=[$mc]', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"
done
unset CFSAR_FP16_SPLIT CFSAR_FP16_MCORR
for p in bf16 fp16; do timeout 600 python bench.py --config cfg4 --steps 4 --warmup 2 --no-cpu-baseline --no-config-legs --no-fp16-leg --precision $p > $O/bench_cfg4_$p.json 2> $O/bench_cfg4_$p.err; python -c "
import json
d=json.loads(open('$O/bench_cfg4_$p.json').read().strip().splitlines()[-1]); print('cfg4 $p', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"; done
