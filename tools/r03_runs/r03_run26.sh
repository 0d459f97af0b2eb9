timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -k "cfg3_four" 2>&1 | grep -E "assert|^E" | head -10
timeout 1500 python tools/parity_report.py 2>&1 | grep -v amdgpu | tail -30
