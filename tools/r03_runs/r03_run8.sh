mkdir -p gpurun_out/r3h
B() { python bench.py --no-cpu-baseline --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['max_abs_dlogits'])"; }
python clip-fsar_amd/build.py --dev --force > /dev/null 2>&1
for i in 1 2; do
  CFSAR_DEV_LIB=1 CFSAR_DEV_VIT_PATHS=11,-1 B "nopk opath1      " >> gpurun_out/r3h/ab.txt
  CFSAR_DEV_LIB=1 B "nopk opath2(new) " >> gpurun_out/r3h/ab.txt
  CFSAR_DEV_LIB=1 CFSAR_DEV_VIT_PATHS=2,-1 B "nopk opath2 all K " >> gpurun_out/r3h/ab.txt
  CFSAR_DEV_LIB=1 CFSAR_DEV_VIT_PATHS=0,-1 B "nopk opath0 all K " >> gpurun_out/r3h/ab.txt
done
CFSAR_BUILD_DEFS="-DCFSAR_PACKED_FP32" python clip-fsar_amd/build.py --dev --force > /dev/null 2>&1
for i in 1 2; do
  CFSAR_DEV_LIB=1 B "packed opath2(new)" >> gpurun_out/r3h/ab.txt
done
B "product" >> gpurun_out/r3h/ab.txt
cat gpurun_out/r3h/ab.txt
