timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tools/r03_refresh.sh > gpurun_out/r3final_refresh.log 2>&1
bash tools/collect_profiles.sh r03 > gpurun_out/prof_r03.log 2>&1
CMD_EXTRA="--config rn50" bash tools/collect_profiles.sh r03_rn50 > gpurun_out/prof_r03_rn50.log 2>&1
tail -9 gpurun_out/r3final_refresh.log
head -8 gpurun_out/prof_r03/kernel_summary.txt | cut -c1-200
head -16 gpurun_out/prof_r03_rn50/kernel_summary.txt | cut -c1-200
