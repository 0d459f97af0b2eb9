#!/bin/bash
# r06 session 7: cfsar_patch_embed for 14 x 14 patches (ViT-L/14): kernel + e2e tests, cfg4 with and without it (same process, alternated).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_s7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "patch_embed" > $O/pytest_kernels.log 2>&1; tail -6 $O/pytest_kernels.log
timeout 1800 python -m pytest tests/test_gpu_e2e.py -q -m gpu -k "fused_patch or cfg3_cfg4_full_size or harness_batch_cfg3_cfg4 or strict_mode_single" > $O/pytest_e2e.log 2>&1; tail -6 $O/pytest_e2e.log
timeout 1200 python - > $O/patch14_ab.log 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
B = bench.default_episodes_per_step("cfg4")
for prec in ("bf16", "fp16"):
    for rep in range(3):
        for fused in (True, False):
            leg = bench.timed_leg("cfg4", prec, B, 4, dev, None, vit_options={"fused_patch": fused})
            print("cfg4 %s fused_patch=%s: %.3f episodes/s (%d per step), golden %.2e" % (prec, fused, leg["value"], B, leg["parity"]["max_abs_dlogits"]), flush=True)
PY
grep -v amdgpu.ids $O/patch14_ab.log | tail -14
