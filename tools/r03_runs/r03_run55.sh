timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -2
for i in 1 2; do python bench.py --no-cpu-baseline --no-fp16-leg --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=16', d['value'], d['ms_per_step'], d['parity']['max_abs_dlogits'])"; done
python bench.py --no-cpu-baseline --precision fp32 --steps 4 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32', d['value'], d['parity']['max_abs_dlogits'])"
