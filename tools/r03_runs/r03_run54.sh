timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for i in 1 2; do python bench.py --episodes-per-step 1 --no-cpu-baseline --no-fp16-leg --steps 80 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=1', d['value'], d['ms_per_step'], d['parity']['max_abs_dlogits'])"; done
python bench.py --episodes-per-step 2 --no-cpu-baseline --no-fp16-leg --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=2', d['value'], d['ms_per_step'])"
python bench.py --no-cpu-baseline --no-fp16-leg --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=16', d['value'], d['ms_per_step'], d['roofline']['frac'])"
