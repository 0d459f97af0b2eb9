timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "skinny or small or temporal or fp32 or f32" 2>&1 | tail -3
timeout 300 python tools/skinny_time.py 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_tail.py -q -x 2>&1 | tail -2
for i in 1 2; do python bench.py --episodes-per-step 1 --no-cpu-baseline --no-fp16-leg --steps 80 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=1', d['value'], d['ms_per_step'], d['parity']['max_abs_dlogits'])"; done
