"""Dev tool: run-to-run determinism of the engine (same inputs, fresh engine each time) and B=16 vs B=1 agreement."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch
from _cases import load_golden, case_inputs, maxdiff
from clip_fsar_amd.engine import ClipFsarEngine
g = load_golden("cfg2_B16_5w1s_T8"); m = g["meta"]
a, sd, tt, te, ep0 = case_inputs(m)
eps = [ep0] + [case_inputs(m, episode=m["episode"] + e)[4] for e in range(1, 4)]
dev = torch.device("cuda")
def run(eng, es, taps=None):
    sup = torch.stack([e["support_set"] for e in es]).to(dev); tgt = torch.stack([e["target_set"] for e in es]).to(dev)
    sl = torch.stack([e["support_labels"] for e in es]).to(dev); rl = torch.stack([e["real_support_labels"] for e in es]).to(dev)
    lo, cl = eng.forward(sup, tgt, sl, rl, way=m["way"], T=m["T"], taps=taps)
    torch.cuda.synchronize()
    return lo.cpu(), cl.cpu()
for mode in (os.environ.get("MODES", "single,nofold").split(",")):
    os.environ["CFSAR_LN_FOLD"] = "0" if mode == "nofold" else "1"
    eng = ClipFsarEngine(a, sd, tt, te, precision="bf16", device="cuda")
    ref = None
    nd = 0
    for it in range(12):
        taps = {} if mode != "dual" else None
        lo, cl = run(eng, [eps[it % 2]], taps)
        key = it % 2
        if ref is None: ref = {}
        if key not in ref:
            ref[key] = (lo, cl, {k: v.clone().cpu() if isinstance(v, torch.Tensor) else v for k, v in (taps or {}).items()})
        else:
            d = maxdiff(lo, ref[key][0])
            if d != 0:
                nd += 1
                msg = "%s iter %d: logits differ by %.3e" % (mode, it, d)
                if taps:
                    for k in ("ln_pre", "block0", "block1", "block5", "block11", "feats"):
                        if k in taps and k in ref[key][2]:
                            msg += " | %s %.3e" % (k, maxdiff(taps[k].float().cpu(), ref[key][2][k].float()))
                print(msg)
    print(mode, "non-identical repeats:", nd, "of 10")
