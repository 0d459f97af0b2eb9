#!/bin/bash
# r05 session 16: what the fp16 mode's second stream word
# costs in time: two-word (product) / one-word (developer option) / two-word with the second word moved as ONE byte per element (developer library,
# dbg bit 23: a TIMING experiment, the numbers of that arm are not the mode's) / bf16 mode, alternated.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_s16; mkdir -p $O
for r in 1 2; do
CFSAR_DEV_LIB=1 timeout 900 python tools/fp16_stream_time.py 36 1 > $O/stream_plain_$r.log 2>&1; tail -3 $O/stream_plain_$r.log
CFSAR_DEV_LIB=1 CFSAR_DEV_VIT_DBG=8388608 timeout 900 python tools/fp16_stream_time.py 36 1 > $O/stream_byte_$r.log 2>&1; tail -3 $O/stream_byte_$r.log
done
