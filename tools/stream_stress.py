"""Dev tool (GPU box): does a kernel give bit-identical results when another copy of it (or of another kernel) runs concurrently on
a second HIP stream?  Each op gets two independent input / output sets; the reference result comes from a solo run.
usage: python tools/stream_stress.py [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from clip_fsar_amd import hip

F_ = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N, D, H = 197, 768, 12
M = F_ * N
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(1)


def rnd(*s, scale=1.0, dtype=torch.float32):
    return (torch.randn(*s, generator=g) * scale).to(dtype).to(dev)


def make_ops(tag):
    ops = {}
    x16 = rnd(M, D, dtype=torch.float16)
    Wq = rnd(3 * D, D, scale=D ** -0.5, dtype=torch.float16)
    cq, dq = rnd(3 * D), rnd(3 * D)
    rstat = torch.empty(M, 4, device=dev)
    hip.row_stats(x16, rstat, M, D)
    qkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
    ops["lnfold_qkv"] = (lambda: hip.gemm_lnfold(x16, Wq, qkv, cq, dq, rstat, M=M), lambda: qkv)
    Wf = rnd(4 * D, D, scale=D ** -0.5, dtype=torch.float16)
    cf, df = rnd(4 * D), rnd(4 * D)
    u = torch.empty(M, 4 * D, device=dev, dtype=torch.bfloat16)
    ops["lnfold_fc_gelu"] = (lambda: hip.gemm_lnfold(x16, Wf, u, cf, df, rstat, act=hip.ACT_QUICKGELU, M=M), lambda: u)
    qkv_in = rnd(M, 3 * D, dtype=torch.bfloat16)
    o = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    ops["attention"] = (lambda: hip.vit_attention(qkv_in, o, F_, N, D, H), lambda: o)
    Wo = rnd(D, D, scale=D ** -0.5, dtype=torch.bfloat16)
    bo = rnd(D)
    oin = rnd(M, D, dtype=torch.bfloat16)
    x0 = rnd(M, D, dtype=torch.float16)
    xr = x0.clone()
    part = torch.empty(M, D // 64, 2, device=dev)

    def res_stats():
        xr.copy_(x0)
        hip.gemm_residual_stats(oin, Wo, xr, bo, part, M=M)
    ops["out_residual_stats"] = (res_stats, lambda: torch.cat([xr.float().reshape(-1), part.reshape(-1)]))
    Wp = rnd(D, 4 * D, scale=(4 * D) ** -0.5, dtype=torch.bfloat16)
    uin = rnd(M, 4 * D, dtype=torch.bfloat16)
    xr2 = x0.clone()
    part2 = torch.empty(M, D // 64, 2, device=dev)

    def proj_stats():
        xr2.copy_(x0)
        hip.gemm_residual_stats(uin, Wp, xr2, bo, part2, M=M)
    ops["proj_residual_stats"] = (proj_stats, lambda: torch.cat([xr2.float().reshape(-1), part2.reshape(-1)]))
    rs2 = torch.empty(M, 4, device=dev)
    ops["finalize"] = (lambda: hip.ln_stats_finalize(part, rs2, M, D // 64, D), lambda: rs2)
    frames = rnd(F_, 3, 224, 224)
    patches = torch.empty(F_ * 196, 768, device=dev, dtype=torch.bfloat16)
    ops["im2col"] = (lambda: hip.im2col_patches(frames, patches, 16), lambda: patches)
    wpatch = rnd(D, 768, scale=768 ** -0.5, dtype=torch.bfloat16)
    pos = rnd(N, D)
    xe = torch.zeros(M, D, device=dev, dtype=torch.float16)
    pin = rnd(F_ * 196, 768, dtype=torch.bfloat16)
    ops["patch_gemm"] = (lambda: hip.gemm(pin, wpatch, xe, residual=pos, M=F_ * 196, N=D, K=768, ldo=D, ldr=D, row_group=196, row_gap=1,
                                          row_off=1, res_mod=196, res_off=1), lambda: xe)
    lw, lb = rnd(D) * 0.1 + 1, rnd(D) * 0.1
    xl = rnd(M, D, dtype=torch.float16)
    xo = torch.empty(M, D, device=dev, dtype=torch.float16)
    ops["layernorm"] = (lambda: hip.layernorm(xl, xo, lw, lb, M, D), lambda: xo)
    Wpl = rnd(3 * D, D, scale=D ** -0.5, dtype=torch.bfloat16)
    hin = rnd(M, D, dtype=torch.bfloat16)
    qkv2 = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
    ops["plain_qkv"] = (lambda: hip.gemm(hin, Wpl, qkv2, bias=cq, M=M), lambda: qkv2)
    a32, w32 = rnd(F_, D), rnd(512, D, scale=D ** -0.5)
    f32o = torch.zeros(2 * F_, 512, device=dev)
    ops["skinny"] = (lambda: hip.gemm(a32, w32, f32o, M=F_, N=512, K=D, ldo=512, row_group=F_, row_gap=F_, row_off=0), lambda: f32o)
    return ops


A, B = make_ops("a"), make_ops("b")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
names = list(A)
refs = {}
for n in names:
    for tag, ops in (("a", A), ("b", B)):
        ops[n][0]()
        torch.cuda.synchronize()
        refs[(n, tag)] = ops[n][1]().clone()
only = os.environ.get("ONLY")
pairs = [(n, n) for n in names if not only or n in only.split(",")] + ([] if only else [("lnfold_qkv", "attention"), ("attention", "out_residual_stats"), ("lnfold_fc_gelu", "proj_residual_stats"),
                                   ("finalize", "lnfold_qkv"), ("layernorm", "attention")])
ITERS = int(os.environ.get("ITERS", "15"))
for na, nb in pairs:
    bad = 0
    for it in range(ITERS):
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            for _ in range(3):
                A[na][0]()
        with torch.cuda.stream(s2):
            for _ in range(3):
                B[nb][0]()
        torch.cuda.synchronize()
        da = float((A[na][1]().float() - refs[(na, "a")].float()).abs().max())
        db = float((B[nb][1]().float() - refs[(nb, "b")].float()).abs().max())
        if da != 0 or db != 0:
            bad += 1
            if bad <= 3:
                print("   %s | %s iter %d: diff a %.3e  b %.3e" % (na, nb, it, da, db))
                for tag, ops_, nm in (("a", A, na), ("b", B, nb)):
                    cur, ref = ops_[nm][1]().float(), refs[(nm, tag)].float()
                    if cur.dim() == 2 and float((cur - ref).abs().max()) != 0:
                        r, c = torch.nonzero(cur != ref, as_tuple=True)
                        print("      %s: %d elements differ; rows %d..%d (%d distinct), cols %d..%d (%d distinct); row%%256 %d..%d"
                              % (tag, r.numel(), int(r.min()), int(r.max()), r.unique().numel(), int(c.min()), int(c.max()), c.unique().numel(),
                                 int((r % 256).min()), int((r % 256).max())))
    print("%-22s || %-22s : %d / %d runs differ" % (na, nb, bad, ITERS))
