#!/bin/bash
# r05 session 29: GPU suite + smoke on the round's last tree (RN50 batch / launch bound changes are Python-side).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s29; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_all.log 2>&1; echo "rc $?" >> $O/pytest_all.log; tail -4 $O/pytest_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
