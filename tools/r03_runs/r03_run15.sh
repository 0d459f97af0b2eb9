mkdir -p gpurun_out/r3o; rm -f gpurun_out/r3o/attn.txt
for v in 31 2 6 10 11 12 18 19 21 22 4; do echo "variant $v" >> gpurun_out/r3o/attn.txt; ATTN_VARIANT=$v python tools/attn_time.py 1280 2>&1 | grep "us " >> gpurun_out/r3o/attn.txt; done
cat gpurun_out/r3o/attn.txt
