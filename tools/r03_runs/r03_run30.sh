timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3x3" 2>&1 | tail -3
timeout 600 python tools/rn_conv_ab.py 16 2>&1 | grep -v amdgpu
for i in 1 2; do python bench.py --config rn50 --no-cpu-baseline --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RN50', d['value'], d['ms_per_step'], d['parity'].get('max_abs_dlogits'), d['roofline'].get('frac_end_to_end'))"; done
