mkdir -p gpurun_out/r3g
python clip-fsar_amd/build.py --dev --force > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or lnfold or residual or vit" > gpurun_out/r3g/pytest.txt 2>&1
tail -3 gpurun_out/r3g/pytest.txt
CFSAR_DEV_LIB=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "forced_variants" > gpurun_out/r3g/pytest_dev.txt 2>&1
tail -3 gpurun_out/r3g/pytest_dev.txt
AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py 16 24:0 26:0 28:0 30:0 20:0 > gpurun_out/r3g/ab.txt 2>&1
timeout 300 python tools/vit_trace.py 16 qkv 30 0 >> gpurun_out/r3g/ab.txt 2>&1
timeout 300 python tools/vit_trace.py 16 qkv 26 0 >> gpurun_out/r3g/ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3g/ab.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3g/bench.json 2> gpurun_out/r3g/bench.err; python -c "import json;d=json.load(open('gpurun_out/r3g/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['parity'])"
