#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s17; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -s -k "refuses or rn50 or cfg3_cfg4_full_size or small_cases_fp16" 2>&1 | grep -v "^$" | tail -14
timeout 900 python tools/parity_report.py --cases rn50_5w1s_T2 rn_t_5w2s_T4 2>&1 | tail -8
