"""Experiment (developer library): consecutive tower launches walk their rows in OPPOSITE directions, so a consumer starts with the rows its
producer wrote last -- the ones still in the 256 MB Infinity Cache -- instead of streaming 0.4-3.5 GB activation matrices through an LRU in the
order that evicts every line just before it is needed.  Bench leg (cfg2) with and without, alternated in one process.
usage: CFSAR_DEV_LIB=1 python tools/walk_direction_ab.py [precision] [episodes ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["CFSAR_DEV_LIB"] = "1"
import bench  # noqa: E402
from clip_fsar_amd import hip  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
L = hip.lib()
dev = torch.device("cuda:0")
for B in [int(a) for a in sys.argv[2:]] or [16, 18, 36]:
    res = {0: [], 1: []}
    for r in range(3):
        for arm in (0, 1):
            L.cfsar_debug_set_vit_dbg(arm << 25)
            leg = bench.timed_leg("cfg2", prec, B, max(4, 160 // B), dev, None)
            res[arm].append(leg["value"])
            print(B, "alternating" if arm else "same direction", leg["value"], leg["parity"].get("max_abs_dlogits"), flush=True)
    L.cfsar_debug_set_vit_dbg(0)
    m = {a: sorted(v)[len(v) // 2] for a, v in res.items()}
    print("%s, %d episodes per step: same direction %.1f  alternating %.1f  (%+.2f %%)" % (prec, B, m[0], m[1], 100 * (m[1] / m[0] - 1)))
