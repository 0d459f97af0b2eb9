#!/bin/bash
# session 3: default fp16 mode = wide + two-word stream + split qkv,out,pr: kernel tests, parity table, bench, the whole GPU suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "residual_wide or lnfold_split or copy_rows" > $O/pytest_new.log 2>&1; echo "rc $?" >> $O/pytest_new.log
tail -4 $O/pytest_new.log
timeout 1500 python tools/parity_report.py > $O/parity_table.md 2> $O/parity.err; cp gpurun_out/parity_table.json $O/
cat $O/parity_table.md
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision fp16 > $O/bench_fp16.json 2> $O/bench_fp16.err; tail -c 1500 $O/bench_fp16.json
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "rc $?" >> $O/pytest_all.log; tail -15 $O/pytest_all.log
