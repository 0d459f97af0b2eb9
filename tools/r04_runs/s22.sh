#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "wide or pair or lnfold or correction or means" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -k "fp16 or cfg2_full or cfg3_cfg4 or outlier or pruning" 2>&1 | tail -3
bash tools/r04_runs/s21.sh ${1:-plainc}
