mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or lnfold or residual or vit" > gpurun_out/r3d/pytest.txt 2>&1
tail -3 gpurun_out/r3d/pytest.txt
for defs in "-DCFSAR_EPI_PIPE=0" "-DCFSAR_EPI_PIPE=1"; do
  CFSAR_BUILD_DEFS="$defs" python clip-fsar_amd/build.py --dev --force > /dev/null 2>&1
  echo "== $defs" >> gpurun_out/r3d/ab.txt
  AB_STREAM=fp16 timeout 900 python tools/gemm_ab.py 16 26:0 30:0 20:0 >> gpurun_out/r3d/ab.txt 2>&1
  timeout 600 python tools/vit_variant_check.py 4 26 30 20 >> gpurun_out/r3d/ab.txt 2>&1
done
cat gpurun_out/r3d/ab.txt
