mkdir -p gpurun_out/r3r
AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py 16 30:0 30:4096 30:16384 30:8192 2>&1 | grep -v amdgpu > gpurun_out/r3r/ab.txt
AB_STREAM=fp16 AB_SHAPES=out,proj timeout 900 python tools/gemm_ab.py 16 28:0 28:4096 28:16384 2>&1 | grep -v amdgpu >> gpurun_out/r3r/ab.txt
cat gpurun_out/r3r/ab.txt
