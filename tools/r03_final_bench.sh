# GPU box: the bench lines committed under profiles/r03_*.json
mkdir -p gpurun_out/r3final
O=gpurun_out/r3final
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --no-cpu-baseline --inputs host --no-fp16-leg > $O/bench_host.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --precision fp16 > $O/bench_fp16.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --precision fp32 --steps 4 --warmup 1 > $O/bench_fp32.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --episodes-per-step 1 --steps 80 --warmup 10 > $O/bench_b1.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --config cfg3 --steps 8 > $O/bench_cfg3.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --config cfg4 --steps 6 > $O/bench_cfg4.json 2>> $O/bench.err
python bench.py --no-cpu-baseline --config rn50 > $O/bench_rn50.json 2>> $O/bench.err
for f in bench bench_host bench_fp16 bench_fp32 bench_b1 bench_cfg3 bench_cfg4 bench_rn50; do python -c "
import json,sys
d=json.load(open('$O/$f.json')); r=d.get('roofline') or {}
print('$f', d['value'], d['ms_per_step'], r.get('frac'), r.get('frac_end_to_end'), (d.get('parity') or {}).get('max_abs_dlogits'), (d.get('fp16_mode') or {}).get('value'), (d.get('inputs_host') or {}).get('overlapped_episodes_per_s'))"; done
