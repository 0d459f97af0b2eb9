#!/bin/bash
# GPU box: same-box A/B of compile-time variants of csrc/gemm_vit.hip, all as developer builds (same -DCFSAR_DEV overhead).
# usage: tools/r02_ab_builds.sh "<defs A>" "<defs B>" ...   ("-" = no extra define)
for defs in "$@"; do
  d="$defs"; [ "$d" = "-" ] && d=""
  CFSAR_BUILD_DEFS="$d" python clip-fsar_amd/build.py --dev --force > /dev/null 2>&1
  for i in 1 2; do
    CFSAR_DEV_LIB=1 python bench.py --no-cpu-baseline --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s' % '$defs', d['value'], d['ms_per_step'])"
  done
done
python bench.py --no-cpu-baseline --steps 12 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s' % 'product build', d['value'], d['ms_per_step'])"
