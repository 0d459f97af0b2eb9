mkdir -p gpurun_out/r3a
( tools/ubench/bin/store_burst > gpurun_out/r3a/store_burst.txt 2>&1 )
( timeout 300 tools/ubench/bin/pk_trans_waw > gpurun_out/r3a/pk_trans_waw.txt 2>&1 )
( AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py 16 26:0 26:4 26:16 24:0 24:16 > gpurun_out/r3a/gemm_ab_dma.txt 2>&1 )
( AB_STREAM=fp16 AB_SHAPES=proj timeout 600 python tools/gemm_ab.py 16 20:0 20:4 20:16 > gpurun_out/r3a/gemm_ab_reg.txt 2>&1 )
( timeout 600 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err )
tail -5 gpurun_out/r3a/*.txt; cat gpurun_out/r3a/bench.json
