python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3
timeout 900 python tools/stream_stress.py 2>&1 | grep -v amdgpu | tail -6
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_end_to_end'], d['parity']['max_abs_dlogits'], d['fp16_mode']['value'], d['fp16_mode']['parity']['max_abs_dlogits'])"
