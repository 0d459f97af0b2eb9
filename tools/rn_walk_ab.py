"""Developer library: band-group size of the tile walk for the RN50 tower's 1 x 1 convs that run on the persistent ViT-block kernel (cfsar_gemm_ex
dispatch, gemm.hip kVitGroup), bench leg of the rn50 configuration, alternated.  A walk A/B compares the same kernel instance, so it carries over to the
product build.  usage: CFSAR_DEV_LIB=1 python tools/rn_walk_ab.py [precision]"""
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
arms = (("8 (product)", 0), ("6", 3584), ("4", 512), ("3", 3072), ("16", 1024), ("2", 1536), ("8 column-fastest", 256))
res = {n: [] for n, _ in arms}
for r in range(3):
    for name, dbg in arms:
        L.cfsar_debug_set_gemm_variant(0, dbg)
        leg = bench.timed_leg("rn50", prec, 32, 6, dev, None)
        res[name].append(leg["value"])
        print(r, name, leg["value"], flush=True)
L.cfsar_debug_set_gemm_variant(0, 0)
base = sorted(res[arms[0][0]])[1]
for n, v in res.items():
    print("rn50 %s, 32 episodes per step, band groups of %-18s median %.1f  (%+.2f %%)  %s" % (prec, n, sorted(v)[1], 100 * (sorted(v)[1] / base - 1), v))
