"""Developer library: out_proj (the short-K residual launch) on the register-staged operand path against the LDS-DMA path of the product policy, in the
bench leg (cfg2; cfg4 for the ViT-L/14 shapes), alternated.  usage: CFSAR_DEV_LIB=1 python tools/outproj_path_ab.py [precision]"""
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
for cfg, B, steps in (("cfg2", 36, 6), ("cfg2", 16, 10), ("cfg4", 11, 3)):
    res = {0: [], 1: []}
    for r in range(3):
        for arm in (0, 1):
            # 10 + path: short-K launches only; bit 21: the LN-folded launches keep the product policy -> only out_proj changes
            L.cfsar_debug_set_vit_paths(10 if arm else -1, -1)
            L.cfsar_debug_set_vit_dbg((1 << 21) if arm else 0)
            leg = bench.timed_leg(cfg, prec, B, steps, dev, None)
            res[arm].append(leg["value"])
            print(cfg, B, "register-staged" if arm else "LDS-DMA (product)", leg["value"], leg["parity"].get("max_abs_dlogits"), flush=True)
    L.cfsar_debug_set_vit_paths(-1, -1)
    L.cfsar_debug_set_vit_dbg(0)
    m = {a: sorted(v)[1] for a, v in res.items()}
    print("%s %s, %d episodes per step: out_proj LDS-DMA %.2f  register-staged %.2f  (%+.2f %%)" % (prec, cfg, B, m[0], m[1], 100 * (m[1] / m[0] - 1)))
