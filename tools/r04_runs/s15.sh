#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4s15; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-leg"
for c in cfg3 cfg4; do for p in bf16 fp16; do timeout 600 $B --config $c --precision $p --episodes-per-step 8 > $O/bench_${c}_$p.json 2>/dev/null; python -c "
import json
d=json.loads(open('$O/bench_${c}_$p.json').read().strip().splitlines()[-1]); print('$c $p', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'))"; done; done
timeout 300 $B --precision fp32 > $O/bench_fp32.json 2>/dev/null; python -c "
import json
d=json.loads(open('$O/bench_fp32.json').read().strip().splitlines()[-1]); print('fp32', d['value'], d['parity']['max_abs_dlogits'])"
timeout 300 $B --episodes-per-step 1 > $O/bench_b1.json 2>/dev/null; python -c "
import json
d=json.loads(open('$O/bench_b1.json').read().strip().splitlines()[-1]); print('B=1', d['value'])"
timeout 300 $B --config rn50 > $O/bench_rn50.json 2>/dev/null; python -c "
import json
d=json.loads(open('$O/bench_rn50.json').read().strip().splitlines()[-1]); print('rn50', d['value'])"
timeout 600 python bench.py --inputs host --no-cpu-baseline --no-fp16-leg > $O/bench_host.json 2>/dev/null; python -c "
import json
d=json.loads(open('$O/bench_host.json').read().strip().splitlines()[-1]); print('host', d['value'], d['inputs_host'])"
# the driver's multi-GPU launch line at N = 1: torchrun -> RANK / LOCAL_RANK / WORLD_SIZE from the launcher, RCCL group of one rank
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-fp16-leg > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err; python -c "
import json
d=json.loads(open('$O/bench_torchrun1.json').read().strip().splitlines()[-1]); print('torchrun N=1', d['value'], d['collective'], d['per_rank_episodes_per_s'], d['config']['launcher'])"
