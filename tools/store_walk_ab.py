"""Experiment (developer library): does a consumer launch profit from the rows its producer wrote LAST when (a) it walks its rows in the opposite
direction and (b) the producer's stores use another cache policy?  LN-folded launches (QKV, c_fc: their outputs are read by the attention kernel and
by c_proj) take store policy s in {2 = sc1 nt (product), 0 = plain, 3 = sc1, 4 = sc0 sc1, 6 = sc0}; the residual launches keep the product policy.
usage: CFSAR_DEV_LIB=1 python tools/store_walk_ab.py [episodes] [precision]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["CFSAR_DEV_LIB"] = "1"
import bench  # noqa: E402
from clip_fsar_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 18
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
L = hip.lib()
dev = torch.device("cuda:0")
arms = [(s, w) for s in (2, 0, 3, 4, 6) for w in (0, 1)]
res = {a: [] for a in arms}
for r in range(3):
    for s, w in arms:
        L.cfsar_debug_set_vit_paths(12 if s != 2 else -1, s if s != 2 else -1)
        L.cfsar_debug_set_vit_dbg((1 << 22) | (w << 25))
        leg = bench.timed_leg("cfg2", prec, B, max(4, 160 // B), dev, None)
        res[(s, w)].append(leg["value"])
        print(r, "store", s, "alternating" if w else "same", leg["value"], leg["parity"].get("max_abs_dlogits"), flush=True)
L.cfsar_debug_set_vit_paths(-1, -1)
L.cfsar_debug_set_vit_dbg(0)
base = sorted(res[(2, 0)])[1]
for (s, w), v in res.items():
    m = sorted(v)[1]
    print("%s %2d episodes  store %d  %-11s median %.1f  (%+.2f %%)  %s" % (prec, B, s, "alternating" if w else "same", m, 100 * (m / base - 1), v))
