# GPU box: parity table + the committed bench lines of the final build
mkdir -p gpurun_out/r3final
timeout 1500 python tools/parity_report.py 2>&1 | grep -v amdgpu.ids > gpurun_out/parity_table.md
bash tools/r03_final_bench.sh
