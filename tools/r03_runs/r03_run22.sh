for defs in "-DCFSAR_PRE_OPATH1_ONLY" "-DCFSAR_NOOP"; do
  CFSAR_BUILD_DEFS="$defs" python clip-fsar_amd/build.py --dev --force > /dev/null 2>&1
  echo "== $defs"
  AB_STREAM=fp16 AB_SHAPES=out,proj timeout 900 python tools/gemm_ab.py 16 28:0 24:0 2>&1 | grep -v amdgpu
  for i in 1 2; do CFSAR_DEV_LIB=1 python bench.py --no-cpu-baseline --no-fp16-leg --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"; done
done
