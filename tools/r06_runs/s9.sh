#!/bin/bash
# r06 session 9: fp16_strict on the tiny architectures (split fallback below 128 tokens, 14 x 14 patches), smoke with the strict mode.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_s9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -k "small_cases_fp16_strict or strict_mode_is or strict_option" > $O/pytest.log 2>&1; tail -12 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -5 $O/smoke.log
