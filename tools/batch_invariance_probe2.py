"""Dev tool (GPU box): where does the fp16 mode's batch dependence come from?  Records the correction GEMM's inputs / outputs and the LN-folded GEMM's
output for the first calls of a 1-episode and a 16-episode forward of the same episode 0 and compares the rows of episode 0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from _cases import case_inputs, load_golden, run_engine
from clip_fsar_amd import hip
g = load_golden("cfg2_B16_5w1s_T8"); m = g["meta"]
a, sd, tt, te, ep0 = case_inputs(m)
eps = [ep0] + [case_inputs(m, episode=m["episode"] + e)[4] for e in range(1, 16)]
rec = {}
orig_corr, orig_hp, orig_wide = hip.corr_gemm, hip.gemm_lnfold_hp, hip.gemm_residual_wide
def corr_gemm(mA, wlo, out):
    orig_corr(mA, wlo, out)
    rec.setdefault(cur, []).append(("corr", mA.clone(), out.clone()))
def hp(x, Wg, out, *a_, **k):
    orig_hp(x, Wg, out, *a_, **k)
    rec.setdefault(cur, []).append(("hp", x[:k.get("M") or x.shape[0]].clone(), out[:k.get("M") or x.shape[0]].clone()))
def wide(A, W, x, xlo, *a_, **k):
    orig_wide(A, W, x, xlo, *a_, **k)
    M = k.get("M") or A.shape[0]
    rec.setdefault(cur, []).append(("wide", A[:M].clone(), x[:M].clone()))
hip.corr_gemm, hip.gemm_lnfold_hp, hip.gemm_residual_wide = corr_gemm, hp, wide
cur = "b1"; run_engine(m, a, sd, tt, te, [eps[0]], "fp16")
cur = "b16"; run_engine(m, a, sd, tt, te, eps, "fp16")
# episode 0's frames: B = 1: support frames 0..39, query 40..79; B = 16: supports of all episodes first (16 x 40), then queries
N = 197
def rows_of_ep0(t, kind, B):
    per = t.shape[0] // (80 * B) if kind != "corr" else 1
    if B == 1:
        return t
    S = 40 * per
    return torch.cat([t[:S], t[16 * S:16 * S + S]])
for i, (r1, r16) in enumerate(zip(rec["b1"][:14], rec["b16"][:14])):
    k = r1[0]
    a_in, a_out = rows_of_ep0(r1[1], k, 1), rows_of_ep0(r1[2], k, 1)
    b_in, b_out = rows_of_ep0(r16[1], k, 16), rows_of_ep0(r16[2], k, 16)
    print("%2d %-5s in: %s max|d| %.3e   out: %s max|d| %.3e (|out| max %.3e)" % (i, k, tuple(a_in.shape), float((a_in.float() - b_in.float()).abs().max()),
          tuple(a_out.shape), float((a_out.float() - b_out.float()).abs().max()), float(a_out.float().abs().max())), flush=True)
