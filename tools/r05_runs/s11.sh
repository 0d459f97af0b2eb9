cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out/r05_s11
timeout 900 python tools/parity_multi.py oc_cfg2_B16_5w1s_T8 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s11/parity_oc.log
cp gpurun_out/parity_multi.json gpurun_out/r05_s11/parity_multi_oc.json
timeout 900 python -m pytest tests -q -m gpu -k "multi_episode_reference_goldens and oc_cfg2" 2>&1 | tail -3
