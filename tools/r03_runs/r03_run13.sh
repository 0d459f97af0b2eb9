mkdir -p gpurun_out/r3m
AB_SHAPES=qkv,fc AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py 16 30:0 28:0 31:0 32:0 33:0 34:0 2>&1 | grep -v amdgpu > gpurun_out/r3m/ab.txt
cat gpurun_out/r3m/ab.txt
