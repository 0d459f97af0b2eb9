mkdir -p gpurun_out/r3j
B() { python bench.py --no-cpu-baseline --steps 12 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['frac'], d['parity']['max_abs_dlogits'], d['parity']['meets_north_star'])"; }
for i in 1 2; do
B "bf16" "--precision bf16" >> gpurun_out/r3j/ab.txt
B "fp16" "--precision fp16" >> gpurun_out/r3j/ab.txt
done
B "bf16 cfg3" "--precision bf16 --config cfg3" >> gpurun_out/r3j/ab.txt
B "fp16 cfg3" "--precision fp16 --config cfg3" >> gpurun_out/r3j/ab.txt
B "bf16 cfg4" "--precision bf16 --config cfg4" >> gpurun_out/r3j/ab.txt
B "fp16 cfg4" "--precision fp16 --config cfg4" >> gpurun_out/r3j/ab.txt
B "bf16 b1" "--precision bf16 --episodes-per-step 1 --steps 60" >> gpurun_out/r3j/ab.txt
B "fp16 b1" "--precision fp16 --episodes-per-step 1 --steps 60" >> gpurun_out/r3j/ab.txt
cat gpurun_out/r3j/ab.txt
