timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv3x3" 2>&1 | tail -8
python clip-fsar_amd/build.py --dev > /dev/null 2>&1
timeout 600 python tools/rn_conv_ab.py 16 2>&1 | grep -v amdgpu
