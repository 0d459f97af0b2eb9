timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -5
timeout 600 python tools/lnfold_ab.py 16 2>&1 | grep "^fc"
for i in 1 2; do python bench.py --no-cpu-baseline --no-fp16-leg --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['parity']['max_abs_dlogits'])"; done
