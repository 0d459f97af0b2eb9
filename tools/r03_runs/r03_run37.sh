bash tools/r03_refresh.sh > gpurun_out/r3final_refresh.log 2>&1
bash tools/collect_profiles.sh r03 > gpurun_out/prof_r03.log 2>&1
CMD_EXTRA="--config rn50" bash tools/collect_profiles.sh r03_rn50 > gpurun_out/prof_r03_rn50.log 2>&1
tail -12 gpurun_out/r3final_refresh.log
tail -14 gpurun_out/prof_r03.log | cut -c1-220
