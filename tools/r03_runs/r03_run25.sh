for i in 1 2; do python bench.py --config rn50 --no-cpu-baseline --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RN50', d['value'], d['ms_per_step'], d['parity'].get('max_abs_dlogits'), d['roofline'].get('frac_end_to_end'))"; done
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
