mkdir -p gpurun_out/r3c
timeout 600 python tools/vit_variant_check.py 4 26 30 28 24 > gpurun_out/r3c/check.txt 2>&1
AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py 16 26:0 30:0 26:4 30:4 26:16 30:16 > gpurun_out/r3c/gemm_ab.txt 2>&1
AB_STREAM=fp16 AB_SHAPES=proj timeout 600 python tools/gemm_ab.py 16 20:0 30:0 >> gpurun_out/r3c/gemm_ab.txt 2>&1
timeout 300 python tools/vit_trace.py 16 qkv 30 0 >> gpurun_out/r3c/trace.txt 2>&1
cat gpurun_out/r3c/check.txt gpurun_out/r3c/gemm_ab.txt gpurun_out/r3c/trace.txt
