timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --episodes-per-step 1 --steps 80 --warmup 10 > gpurun_out/r3final/bench_b1.json 2>/dev/null
python -c "import json; d=json.load(open('gpurun_out/r3final/bench_b1.json')); print('b1', d['value'], d['ms_per_step'], d['roofline'].get('frac_end_to_end'))"
python bench.py --no-cpu-baseline --no-fp16-leg --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=16', d['value'], d['ms_per_step'])"
