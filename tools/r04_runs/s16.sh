#!/bin/bash
# RN50 tower's fp16 mode: kernel tests, goldens, steady statistic, bench (bf16 with the fp16 leg; fp16 as the headline), kernel summary
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s16; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemm_ex or conv3x3 or avgpool or stem or rn50 or fp16_residual or policy or bit_stable" 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_gpu_e2e.py -q -x -m gpu -s -k "small_cases or refuses or rn50 or cfg3_cfg4_full_size" 2>&1 | grep -v "^$" | tail -12
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
timeout 600 $B --config rn50 > $O/bench_rn50.json 2>$O/bench_rn50.err; python -c "
import json
d=json.loads(open('$O/bench_rn50.json').read().strip().splitlines()[-1]); print('rn50 bf16', d['value'], d.get('fp16_mode'))"
timeout 600 $B --config rn50 --precision fp16 > $O/bench_rn50_fp16.json 2>$O/bench_rn50_fp16.err; python -c "
import json
d=json.loads(open('$O/bench_rn50_fp16.json').read().strip().splitlines()[-1]); print('rn50 fp16', d['value'], d['ms_per_step'], d['roofline'])"
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o rn50f16 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp16-leg --config rn50 --precision fp16 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-170
