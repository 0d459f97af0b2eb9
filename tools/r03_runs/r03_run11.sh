mkdir -p gpurun_out/r3k
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r3k/pytest.txt; cat gpurun_out/r3k/pytest.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3k/bench.json 2> gpurun_out/r3k/bench.err; python -c "
import json;d=json.load(open('gpurun_out/r3k/bench.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['parity']['max_abs_dlogits']);print(d['fp16_mode'])"
