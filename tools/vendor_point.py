"""One ViT-B/16 block's four GEMM operations three ways on one box, interleaved (GPU dev tool; product library):
  vendor-plain : torch.matmul -> hipBLASLt, no epilogue at all (the point profiles/r01_gemm_ablation.md recorded at M = 63 040)
  reference    : what the reference runs on a GPU for the same four operations, op by op, fp16 (few_shot.py:623-640: ln_1 -> in_proj with bias,
                 out_proj with bias -> residual add, ln_2 -> c_fc with bias -> x * sigmoid(1.702 x) (:614-616), c_proj with bias -> residual add):
                 vendor GEMMs with their bias epilogue + the framework's LayerNorm / elementwise kernels
  ours         : cfsar_gemm_lnfold (QKV), cfsar_gemm_residual_stats (out_proj) + cfsar_ln_stats_finalize, cfsar_gemm_lnfold + QuickGELU (c_fc),
                 cfsar_gemm_residual_stats (c_proj) + cfsar_ln_stats_finalize -- the bf16 mode's launches for the same operations
usage: python tools/vendor_point.py [episodes ...]      (M = 80 * 197 * episodes; default 16 36)"""
import os
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import clip_fsar_amd  # noqa: E402,F401
from clip_fsar_amd import hip  # noqa: E402

D, dev = 768, "cuda"
ROUNDS, ITERS = 5, 4


def timed(fn):
    fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(ITERS):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / ITERS * 1e3


for B in [int(a) for a in sys.argv[1:]] or [16, 36]:
    M = 80 * 197 * B
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    x = rnd(M, D).half()
    o = rnd(M, D).to(torch.bfloat16)
    qkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
    u = torch.empty(M, 4 * D, device=dev, dtype=torch.float16)
    w = {"qkv": rnd(3 * D, D) * D ** -0.5, "out": rnd(D, D) * D ** -0.5, "fc": rnd(4 * D, D) * D ** -0.5, "pr": rnd(D, 4 * D) * (4 * D) ** -0.5}
    bias = {k: rnd(v.shape[0]) * 0.1 for k, v in w.items()}
    lnw, lnb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    # ours: folded weights (gamma = 1, beta = 0: Wg = W, c = row sums, d = bias)
    wg = {k: w[k].half().contiguous() for k in ("qkv", "fc")}
    cv = {k: wg[k].float().sum(1).contiguous() for k in wg}
    wb = {"out": w["out"].to(torch.bfloat16).contiguous(), "pr": w["pr"].half().contiguous()}
    part = torch.empty(M, D // 64, 2, device=dev, dtype=torch.float32)
    rstat = torch.empty(M, 4, device=dev, dtype=torch.float32)
    hip.row_stats(x, rstat, M, D)
    uu = rnd(M, 4 * D).half() * 0.3
    ours = {
        "qkv": lambda: hip.gemm_lnfold(x, wg["qkv"], qkv, cv["qkv"], bias["qkv"], rstat),
        "out": lambda: (hip.gemm_residual_stats(o, wb["out"], x, bias["out"], part), hip.ln_stats_finalize(part, rstat, M, D // 64, D)),
        "fc": lambda: hip.gemm_lnfold(x, wg["fc"], u, cv["fc"], bias["fc"], rstat, act=hip.ACT_QUICKGELU),
        "pr": lambda: (hip.gemm_residual_stats(uu, wb["pr"], x, bias["pr"], part), hip.ln_stats_finalize(part, rstat, M, D // 64, D)),
    }
    # the reference's op sequence in fp16 on the vendor library (CLIP's own GPU dtype)
    xr = x.clone()
    oh = o.half()
    wh = {k: v.half() for k, v in w.items()}
    bh = {k: v.half() for k, v in bias.items()}

    def ref_fc():
        t = F.linear(F.layer_norm(xr.float(), (D,), lnw, lnb).half(), wh["fc"], bh["fc"])         # LayerNorm subclass computes in fp32 (few_shot.py:604-610)
        return t * torch.sigmoid(1.702 * t)
    hidden = ref_fc()
    ref = {
        "qkv": lambda: F.linear(F.layer_norm(xr.float(), (D,), lnw, lnb).half(), wh["qkv"], bh["qkv"]),
        "out": lambda: xr.add_(F.linear(oh, wh["out"], bh["out"]).mul_(1e-3)),                      # scaled: keeps the in-place stream finite over many repeats
        "fc": ref_fc,
        "pr": lambda: xr.add_(F.linear(hidden, wh["pr"], bh["pr"]).mul_(1e-3)),
    }
    ab = {k: (torch.randn(M, v.shape[1], device=dev).to(torch.bfloat16), v.to(torch.bfloat16)) for k, v in w.items()}
    plain = {k: (lambda k=k: torch.matmul(ab[k][0], ab[k][1].t())) for k in w}
    t = {(arm, k): [] for arm in ("plain", "ref", "ours") for k in w}
    for r in range(ROUNDS):
        for k in w:
            t["plain", k].append(timed(plain[k]))
            t["ref", k].append(timed(ref[k]))
            t["ours", k].append(timed(ours[k]))
    print("episodes %d  M = %d" % (B, M))
    tot = {"plain": 0.0, "ref": 0.0, "ours": 0.0}
    for k, name in (("qkv", "ln_1 + QKV"), ("out", "out_proj + residual"), ("fc", "ln_2 + c_fc + QuickGELU"), ("pr", "c_proj + residual")):
        fl = 2.0 * M * w[k].shape[0] * w[k].shape[1]
        row = []
        for arm in ("plain", "ref", "ours"):
            med = statistics.median(t[arm, k])
            tot[arm] += med
            row.append("%s %8.1f us (%6.0f TF/s)" % (arm, med, fl / med / 1e6))
        print("  %-26s %s" % (name, "   ".join(row)))
    print("  %-26s plain %8.1f us   ref %8.1f us   ours %8.1f us   ours / ref = %.3f, ours / plain = %.3f" % (
        "four operations", tot["plain"], tot["ref"], tot["ours"], tot["ours"] / tot["ref"], tot["ours"] / tot["plain"]))
    del x, o, qkv, u, part, rstat, uu, xr, oh, hidden, ab
    torch.cuda.empty_cache()
