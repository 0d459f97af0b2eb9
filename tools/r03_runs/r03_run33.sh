for i in 1 2; do python bench.py --config rn50 --no-cpu-baseline --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RN50', d['value'], d['ms_per_step'], d['parity'], d['roofline'].get('frac_end_to_end'))"; done
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -k "rn" 2>&1 | tail -3
