timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "rn50_conv3 or relu_and_bf16 or epilogue" 2>&1 | tail -3
timeout 900 python tools/rn_gemm_ab.py 0:0 13:0 2>&1 | grep -v amdgpu | grep "c1" | cut -c1-200
for i in 1 2; do python bench.py --config rn50 --no-cpu-baseline --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RN50', d['value'], d['ms_per_step'])"; done
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -k "rn" 2>&1 | tail -2
