mkdir -p gpurun_out/r3b
for s in qkv fc out; do timeout 300 python tools/vit_trace.py 16 $s 26 0 12 23 46 >> gpurun_out/r3b/trace.txt 2>&1; done
timeout 300 python tools/vit_trace.py 16 qkv 24 0 23 >> gpurun_out/r3b/trace.txt 2>&1
cat gpurun_out/r3b/trace.txt
