AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py 16 30:0 30:32768 30:65536 2>&1 | grep -v amdgpu
