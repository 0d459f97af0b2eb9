mkdir -p gpurun_out/r3l
for v in 31 30 7; do for F in 1280 80; do ATTN_VARIANT=$v python tools/attn_time.py $F 2>&1 | grep -v amdgpu >> gpurun_out/r3l/attn.txt; done; done
cat gpurun_out/r3l/attn.txt
