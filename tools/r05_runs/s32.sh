#!/bin/bash
# r05 session 32 (reused for every walk-policy step): the default bench line with the previous library and with this one, alternated.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${OUTDIR:-r05_s32}; mkdir -p $O
for i in 1 2 3; do
  CFSAR_LIB_PATH=clip-fsar_amd/libclipfsar_hip_prev.so timeout 900 python bench.py --no-cpu-baseline > $O/prev_$i.json 2> $O/prev_$i.err
  timeout 900 python bench.py --no-cpu-baseline > $O/new_$i.json 2> $O/new_$i.err
done
python - <<PY | tee $O/ab.txt
import json
for arm in ("prev", "new"):
    for i in (1, 2, 3):
        d = json.loads(open("$O/%s_%d.json" % (arm, i)).read().strip().splitlines()[-1])
        print(arm, d["value"], d["roofline"]["frac"], d["fp16_mode"]["value"], {k: v["value"] for k, v in d["configs"].items()}, d["parity"]["max_abs_dlogits"], d["fp16_mode"]["parity"]["max_abs_dlogits"])
PY
