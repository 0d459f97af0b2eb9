#!/bin/bash
# GPU box: rebuild the developer library with one extra compile flag / define at a time (e.g. "-DCFSAR_PACKED_FP32" = with packed-fp32 ops) and run the two-stream stress on the LN-folded QKV GEMM
for defs in "$@"; do
  d="$defs"; [ "$d" = "-" ] && d=""
  CFSAR_BUILD_DEFS="$d" python clip-fsar_amd/build.py --dev --force > /dev/null 2>&1
  echo "== $defs"
  CFSAR_DEV_LIB=1 ONLY=lnfold_qkv ITERS=${ITERS:-150} timeout 600 python tools/stream_stress.py 40 2>&1 | grep -v amdgpu | tail -3
done
