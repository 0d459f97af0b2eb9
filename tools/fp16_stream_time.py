"""What the fp16 mode's second stream word costs: bench.py's timed leg (cfg2, fp16) with the two-word residual stream (product) against the one-word
stream (developer option fp16_lo=False; NOT a product setting: profiles/r04_parity_table.md, its deviation is outside the mode's statistic), alternated
on one box.  Half of the difference bounds what a 3-byte stream (fp16 + an 8-bit second word) could return.
usage: python tools/fp16_stream_time.py [episodes_per_step] [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 36
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
arms = (("two-word (product)", None), ("one-word", {"fp16_lo": False}), ("bf16 mode", "bf16"))
res = {n: [] for n, _ in arms}
for r in range(rounds):
    for name, opt in arms:
        prec, vo = ("bf16", None) if opt == "bf16" else ("fp16", opt)
        leg = bench.timed_leg("cfg2", prec, B, 8, dev, None, vit_options=vo)
        res[name].append(leg["value"])
        print(r, name, leg["value"], leg["ms_per_step"], leg["parity"].get("max_abs_dlogits"), flush=True)
for n, v in res.items():
    print("%-20s median %.1f episodes/s  %s" % (n, sorted(v)[len(v) // 2], v))
