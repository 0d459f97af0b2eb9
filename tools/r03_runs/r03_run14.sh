mkdir -p gpurun_out/r3n
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" 2>&1 | tail -2
for F in 1280 80; do python tools/attn_time.py $F 2>&1 | grep -v amdgpu >> gpurun_out/r3n/attn.txt; done
python tools/attn_time.py 640 257 2>&1 | grep -v amdgpu >> gpurun_out/r3n/attn.txt
ATTN_VARIANT=30 python tools/attn_time.py 1280 2>&1 | grep -v amdgpu >> gpurun_out/r3n/attn.txt
cat gpurun_out/r3n/attn.txt
