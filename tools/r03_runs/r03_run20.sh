mkdir -p gpurun_out/r3t
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r3t/pytest.txt; cat gpurun_out/r3t/pytest.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 12 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());print('pruned', d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['frac_end_to_end'],d['parity']['max_abs_dlogits'],d['fp16_mode']['value'],d['fp16_mode']['parity']['max_abs_dlogits'])"
done
python bench.py --no-cpu-baseline --episodes-per-step 1 --steps 80 --warmup 10 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());print('b1', d['value'],d['ms_per_step'])"
