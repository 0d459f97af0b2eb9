timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -k "lnfold or residual or stream or head" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -x 2>&1 | tail -3
for i in 1 2; do python bench.py --episodes-per-step 1 --no-cpu-baseline --no-fp16-leg --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=1', d['value'], d['ms_per_step'], d['parity']['max_abs_dlogits'])"; done
python bench.py --episodes-per-step 2 --no-cpu-baseline --no-fp16-leg --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=2', d['value'], d['ms_per_step'])"
python bench.py --no-cpu-baseline --no-fp16-leg --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=16', d['value'], d['ms_per_step'])"
