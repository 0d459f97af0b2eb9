mkdir -p gpurun_out/r3q
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r3q/pytest.txt; cat gpurun_out/r3q/pytest.txt
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --steps 12 2>/dev/null | python -c "
import sys,json;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['frac_end_to_end'],d['parity']['max_abs_dlogits'],d['fp16_mode']['value'],d['fp16_mode']['parity']['max_abs_dlogits'])"; done
