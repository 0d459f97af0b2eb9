for ds in 0 1 0; do CFSAR_DUAL_STREAM=$ds python bench.py --episodes-per-step 1 --no-cpu-baseline --no-fp16-leg --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=1 dual=$ds', d['value'], d['ms_per_step'])"; done
for ds in 0 1; do CFSAR_DUAL_STREAM=$ds python bench.py --episodes-per-step 2 --no-cpu-baseline --no-fp16-leg --steps 40 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=2 dual=$ds', d['value'], d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/b1prof -o t -- python bench.py --episodes-per-step 1 --no-cpu-baseline --no-kernel-events --no-fp16-leg --steps 30 --warmup 5 > gpurun_out/b1prof.log 2>&1
python tools/trace_summary.py gpurun_out/b1prof/t_kernel_trace.csv 0 | head -24 | cut -c1-200
rm -rf gpurun_out/b1prof
