mkdir -p gpurun_out/r3p; rm -f gpurun_out/r3p/attn.txt
for rep in 1 2; do for v in 31 10 40 41 42 43 44 22; do echo "variant $v" >> gpurun_out/r3p/attn.txt; ATTN_VARIANT=$v python tools/attn_time.py 1280 2>&1 | grep "us " >> gpurun_out/r3p/attn.txt; done; done
for v in 0 45 46 47 24 25; do echo "variant $v (257)" >> gpurun_out/r3p/attn.txt; ATTN_VARIANT=$v python tools/attn_time.py 640 257 2>&1 | grep "us " >> gpurun_out/r3p/attn.txt; done
for v in 31 10 40 22; do echo "variant $v (80 frames)" >> gpurun_out/r3p/attn.txt; ATTN_VARIANT=$v python tools/attn_time.py 80 2>&1 | grep "us " >> gpurun_out/r3p/attn.txt; done
cat gpurun_out/r3p/attn.txt
