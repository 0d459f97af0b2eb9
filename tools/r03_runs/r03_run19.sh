mkdir -p gpurun_out/r3s
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r3s/pytest.txt; cat gpurun_out/r3s/pytest.txt
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 12 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());print('pruned', d['value'],d['ms_per_step'],d['roofline']['frac'],d['parity']['max_abs_dlogits'],d['fp16_mode']['value'],d['fp16_mode']['parity']['max_abs_dlogits'])"
CFSAR_FULL_LAST_BLOCK=1 python bench.py --no-cpu-baseline --steps 12 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());print('full  ', d['value'],d['ms_per_step'],d['roofline']['frac'],d['parity']['max_abs_dlogits'],d['fp16_mode']['value'],d['fp16_mode']['parity']['max_abs_dlogits'])"
done
