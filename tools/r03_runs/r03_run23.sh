timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "lnfold or gelu" 2>&1 | tail -3
for defs in "-DCFSAR_GELU_UNFUSED" "-DCFSAR_NOOP" "-DCFSAR_GELU_UNFUSED" "-DCFSAR_NOOP"; do
  CFSAR_BUILD_DEFS="$defs" python clip-fsar_amd/build.py --dev --force > /dev/null 2>&1
  echo "== $defs"
  timeout 600 python tools/lnfold_ab.py 16 2>&1 | grep "^fc"
  for i in 1 2; do CFSAR_DEV_LIB=1 python bench.py --no-cpu-baseline --no-fp16-leg --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['parity'])"; done
done
