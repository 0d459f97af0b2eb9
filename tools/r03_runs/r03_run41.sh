timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -k "lnfold or residual or stream or head" 2>&1 | tail -6
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -x 2>&1 | tail -3
for fs in 1 0 1 0; do CFSAR_FUSE_STATS=$fs python bench.py --episodes-per-step 1 --no-cpu-baseline --no-fp16-leg --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=1 fuse=$fs', d['value'], d['ms_per_step'], d['parity']['max_abs_dlogits'])"; done
for fs in 1 0; do CFSAR_FUSE_STATS=$fs python bench.py --episodes-per-step 2 --no-cpu-baseline --no-fp16-leg --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=2 fuse=$fs', d['value'], d['ms_per_step'])"; done
for fs in 1 0; do CFSAR_FUSE_STATS=$fs python bench.py --no-cpu-baseline --no-fp16-leg --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=16 fuse=$fs', d['value'], d['ms_per_step'])"; done
