#!/bin/bash
# r05 session 24: per-row cost of the four block operations against the batch (stand-alone, product library; vendor plain GEMMs beside them), and the
# kernel tests of the patch-embed binding's argument checks.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s24; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "patch_embed" > $O/pytest_pe.log 2>&1; tail -2 $O/pytest_pe.log
timeout 1500 python tools/vendor_point.py 8 16 18 24 30 36 40 > $O/vendor_point_batches.log 2>&1; grep -v "^/opt" $O/vendor_point_batches.log | tail -45
