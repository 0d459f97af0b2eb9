#!/bin/bash
# r05 session 18: the block's four GEMM operations against the vendor library on the same box (plain hipBLASLt GEMMs, and the reference's own op sequence).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s18; mkdir -p $O
timeout 900 python tools/vendor_point.py 16 36 > $O/vendor_point.log 2>&1; tail -14 $O/vendor_point.log
