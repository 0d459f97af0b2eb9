mkdir -p gpurun_out/r3i
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r3i/pytest.txt; cat gpurun_out/r3i/pytest.txt
timeout 600 python bench.py --no-cpu-baseline --inputs host > gpurun_out/r3i/bench_host.json 2> gpurun_out/r3i/bench_host.err; python -c "
import json;d=json.load(open('gpurun_out/r3i/bench_host.json'));print(d['value'],d['ms_per_step'],d['roofline']['frac'],d['parity']);print(d['inputs_host'])"
