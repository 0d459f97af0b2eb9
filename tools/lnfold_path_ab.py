"""Developer library: the LN-folded short-K launches (QKV, c_fc) on the register-staged (0) / late-DMA (1) operand paths against the early-DMA path (2)
of the product policy, in the bench leg, alternated.  usage: CFSAR_DEV_LIB=1 python tools/lnfold_path_ab.py [precision] [episodes] [config]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["CFSAR_DEV_LIB"] = "1"
import bench  # noqa: E402
from clip_fsar_amd import hip  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 36
cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
L = hip.lib()
dev = torch.device("cuda:0")
arms = (("early-DMA (product)", -1), ("register-staged", 10), ("late-DMA", 11))
res = {n: [] for n, _ in arms}
for r in range(3):
    for name, path in arms:
        L.cfsar_debug_set_vit_paths(path, -1)                         # 10 + path: short-K launches only
        L.cfsar_debug_set_vit_dbg((1 << 22) if path >= 0 else 0)      # bit 22: the residual launches keep the product policy
        leg = bench.timed_leg(cfg, prec, B, max(3, 200 // B) if cfg == "cfg2" else 3, dev, None)
        res[name].append(leg["value"])
        print(r, name, leg["value"], leg["parity"].get("max_abs_dlogits"), flush=True)
L.cfsar_debug_set_vit_paths(-1, -1)
L.cfsar_debug_set_vit_dbg(0)
base = sorted(res[arms[0][0]])[1]
for n, v in res.items():
    print(cfg, "%s %d episodes per step  QKV / c_fc %-20s median %.2f  (%+.2f %%)  %s" % (prec, B, n, sorted(v)[1], 100 * (sorted(v)[1] / base - 1), v))
