#!/bin/bash
# r05 session 28: what c_fc's fixed-point column sums cost in the fp16 mode: the same bench command with the c_proj correction off (no column sums).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_s28; mkdir -p $O
CMD="python bench.py --steps 4 --warmup 2 --precision fp16 --no-cpu-baseline --no-kernel-events --no-fp16-leg --no-config-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_all -o t -- $CMD > $O/all.log 2>&1
python tools/trace_summary.py $O/t_all/t_kernel_trace.csv 0 > $O/summary_mcorr_all.txt
CFSAR_FP16_MCORR=qkv,out,fc rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_nopr -o t -- $CMD > $O/nopr.log 2>&1
python tools/trace_summary.py $O/t_nopr/t_kernel_trace.csv 0 > $O/summary_mcorr_no_pr.txt
rm -rf $O/t_all $O/t_nopr
head -7 $O/summary_mcorr_all.txt | cut -c1-150; head -7 $O/summary_mcorr_no_pr.txt | cut -c1-150
