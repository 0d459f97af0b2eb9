cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r05_s8
timeout 900 python tools/parity_multi.py hc_rn50_5w1s_T8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s8/parity_rn50.log
cp gpurun_out/parity_multi.json gpurun_out/r05_s8/parity_multi_rn50.json
timeout 900 python -m pytest tests -q -m gpu -k "developer_options or prefetcher_collate or multi_episode_reference_goldens and rn50 or frame_gemm" 2>&1 | tail -4
