"""GPU parity tests, kernel level: every C-ABI entry point of libclipfsar_hip.so against the CPU oracle
(oracle/clipfsar_oracle.py) or a plain torch fp32 reference of the same op, on seeded inputs.

Tolerances (written here, per the north star "within 1e-3 fp32"):
  fp32 kernels: 2e-4 absolute on O(1)-O(30) values (accumulation-order differences only);
  bf16 kernels: compared against the fp32 op applied to bf16-ROUNDED inputs, 2e-2 relative to the output scale.
"""
import math

import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import clipfsar_oracle as orc  # noqa: E402
from _cases import maxdiff  # noqa: E402


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from clip_fsar_amd import hip as h
    h.lib()
    return h


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


GEMM_SHAPES = [(77, 64, 64), (300, 192, 128), (1000, 768, 768), (197 * 3, 2304, 768), (130, 512, 3072), (45, 2048, 512)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_gemm_plain_and_epilogues(hip, M, N, K, dtype):
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    A = _rand(M, K, seed=1).to(td)
    W = _rand(N, K, seed=2, scale=K ** -0.5).to(td)
    bias = _rand(N, seed=3)
    res = _rand(M, N, seed=4)
    ref0 = A.float() @ W.float().t()
    tol = 2e-4 if dtype == "f32" else 2e-2
    Ad, Wd, bd, rd = A.cuda(), W.cuda(), bias.cuda(), res.cuda()
    # plain, asymmetric operands (transpose-detecting)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    hip.gemm(Ad, Wd, out)
    assert maxdiff(out.cpu(), ref0) < tol * max(1.0, float(ref0.abs().max()))
    # bias + QuickGELU, output in the input dtype
    out2 = torch.empty(M, N, device="cuda", dtype=td)
    hip.gemm(Ad, Wd, out2, bias=bd, act=hip.ACT_QUICKGELU)
    ref = orc.quick_gelu(ref0 + bias)
    assert maxdiff(out2.float().cpu(), ref) < (tol if dtype == "f32" else 4e-2) * max(1.0, float(ref.abs().max()))
    # bias + GELU(erf)
    hip.gemm(Ad, Wd, out, bias=bd, act=hip.ACT_GELU_ERF)
    assert maxdiff(out.cpu(), orc.gelu_erf(ref0 + bias)) < tol * max(1.0, float(ref0.abs().max()))
    # bias + residual, in place on the residual buffer (the residual-stream update)
    x = rd.clone()
    hip.gemm(Ad, Wd, x, bias=bd, residual=x)
    assert maxdiff(x.cpu(), ref0 + bias + res) < tol * max(1.0, float(ref0.abs().max()))


@pytest.mark.parametrize("M,N,K", [(1, 8, 64), (40, 512, 768), (85, 1536, 512), (85, 512, 2048), (96, 2048, 512), (97, 100, 96),
                                   (128, 516, 64), (170, 1536, 512), (192, 36, 160), (33, 24, 32), (7, 36, 128), (192, 516, 256), (130, 2052, 384)])
def test_gemm_skinny_f32(hip, M, N, K):
    """The skinny fp32 kernels (M <= 192: temporal head / final projection of one or two episodes).  K % 128 == 0: register tiles with K
    split over the workgroup's four waves (16- and 32-row tiles, one- and two-chunk quarters, N not a multiple of 32); otherwise the
    LDS-ring form: every rows-per-thread instance and both chunk widths (K % 64 == 0 with M <= 128, else 32-float chunks), N not a
    multiple of 8, one-chunk and two-chunk K.  All epilogues, and the output-row remap of the final projection (rows of a set scattered
    into [B, S+Q, T])."""
    A = _rand(M, K, seed=31)
    W = _rand(N, K, seed=32, scale=K ** -0.5)
    bias, res = _rand(N, seed=33), _rand(M, N, seed=34)
    ref0 = A.double() @ W.double().t()
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    tol = 2e-5 * max(1.0, float(ref0.abs().max()))
    out = torch.full((M, N), 7.0, device="cuda")
    hip.gemm(Ad, Wd, out)
    assert maxdiff(out.cpu().double(), ref0) < tol
    hip.gemm(Ad, Wd, out, bias=bd, act=hip.ACT_GELU_ERF)
    assert maxdiff(out.cpu(), orc.gelu_erf((ref0 + bias).float())) < 2 * tol
    x = res.cuda().clone()
    hip.gemm(Ad, Wd, x, bias=bd, residual=x)
    assert maxdiff(x.cpu().double(), ref0 + bias + res) < tol
    # row remap: row m -> m + (m // group) * gap + off inside a larger buffer; the other rows stay untouched
    group, gap, off = 8, 5, 3
    rows_out = M + ((M - 1) // group) * gap + off + 2
    y = torch.full((rows_out, N), -3.0, device="cuda")
    hip.gemm(Ad, Wd, y, bias=bd, M=M, N=N, K=K, ldo=N, row_group=group, row_gap=gap, row_off=off)
    idx = torch.tensor([m + (m // group) * gap + off for m in range(M)])
    yc = y.cpu()
    assert maxdiff(yc[idx].double(), ref0 + bias) < tol
    mask = torch.ones(rows_out, dtype=torch.bool)
    mask[idx] = False
    assert torch.all(yc[mask] == -3.0)


def test_gemm_row_remap(hip):
    """patch-embed epilogue: rows scattered behind the class token + positional rows added."""
    F_, npatch, D, K = 3, 16, 128, 64
    A = _rand(F_ * npatch, K, seed=5)
    W = _rand(D, K, seed=6, scale=K ** -0.5)
    pos = _rand(npatch + 1, D, seed=7)
    x = torch.full((F_ * (npatch + 1), D), -7.0, device="cuda")
    hip.gemm(A.cuda(), W.cuda(), x, residual=pos.cuda(), M=F_ * npatch, N=D, K=K, ldo=D, ldr=D, row_group=npatch,
             row_gap=1, row_off=1, res_mod=npatch, res_off=1)
    ref = (A @ W.t()).reshape(F_, npatch, D) + pos[1:]
    got = x.cpu().reshape(F_, npatch + 1, D)
    assert maxdiff(got[:, 1:], ref) < 2e-4
    assert torch.all(got[:, 0] == -7.0)          # class-token rows untouched


@pytest.mark.parametrize("rows,D", [(5, 64), (197 * 4, 768), (33, 1024), (85, 512)])
def test_layernorm(hip, rows, D):
    x = _rand(rows, D, seed=8, scale=3.0) + 0.5
    w, b = _rand(D, seed=9) * 0.1 + 1.0, _rand(D, seed=10) * 0.1
    ref = orc.layer_norm(x, w, b)
    o32 = torch.empty(rows, D, device="cuda")
    hip.layernorm(x.cuda(), o32, w.cuda(), b.cuda(), rows, D)
    assert maxdiff(o32.cpu(), ref) < 2e-5
    o16 = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    hip.layernorm(x.cuda(), o16, w.cuda(), b.cuda(), rows, D)
    assert maxdiff(o16.float().cpu(), ref) < 4e-2
    # strided rows (ln_post on the class-token rows) and in place
    xs = _rand(rows, 3 * D, seed=11)
    os_ = torch.empty(rows, D, device="cuda")
    hip.layernorm(xs.cuda(), os_, w.cuda(), b.cuda(), rows, D, in_stride=3 * D, out_stride=D)
    assert maxdiff(os_.cpu(), orc.layer_norm(xs[:, :D], w, b)) < 2e-5
    xi = x.cuda()
    hip.layernorm(xi, xi, w.cuda(), b.cuda(), rows, D)
    assert maxdiff(xi.cpu(), ref) < 2e-5


@pytest.mark.parametrize("P,res", [(16, 64), (14, 56), (16, 224)])
def test_im2col_and_cls_rows(hip, P, res):
    F_, D = 3, 128
    frames = _rand(F_, 3, res, res, seed=12)
    g = res // P
    kreal = 3 * P * P
    kpad = (kreal + 63) // 64 * 64
    ref = frames.reshape(F_, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(F_ * g * g, kreal)
    for td in (torch.float32, torch.bfloat16, torch.float16):
        out = torch.full((F_ * g * g, kpad), 9.0, device="cuda", dtype=td)
        hip.im2col_patches(frames.cuda(), out, P)
        got = out.float().cpu()
        assert maxdiff(got[:, :kreal], ref.to(td).float()) == 0.0
        assert torch.all(got[:, kreal:] == 0)
    ntok = g * g + 1
    x = torch.zeros(F_ * ntok, D, device="cuda")
    cls, pos = _rand(D, seed=13), _rand(ntok, D, seed=14)
    hip.cls_rows(x, cls.cuda(), pos.cuda(), F_, ntok, D)
    got = x.cpu().reshape(F_, ntok, D)
    assert maxdiff(got[:, 0], (cls + pos[0]).expand(F_, D)) == 0.0
    assert torch.all(got[:, 1:] == 0)


@pytest.mark.parametrize("wd,xd", [("bf16", "f16"), ("f16", "f16")])
@pytest.mark.parametrize("F_,res,D", [(7, 224, 768), (3, 32, 128), (5, 64, 192), (80, 224, 768)])
def test_patch_embed_fused_equals_the_three_launch_form_bitwise(hip, wd, xd, F_, res, D):
    """SURVEY K1 (few_shot.py:672-676): cfsar_patch_embed gathers the GEMM's rows from the fp32 frames itself.  Same rounding of the frames, same
    K order of the MFMA chain: every element equals cfsar_im2col_patches + cfsar_gemm (row remap, pos residual) + cfsar_cls_rows bit for bit
    (ragged row tiles, a partial column tile (D = 192), 1 ... 196 patches per frame); and the whole thing against an fp32 reference."""
    tw, tx = {"bf16": torch.bfloat16, "f16": torch.float16}[wd], {"bf16": torch.bfloat16, "f16": torch.float16}[xd]
    g = res // 16
    npatch, ntok = g * g, g * g + 1
    frames = _rand(F_, 3, res, res, seed=31).cuda()
    w = (_rand(D, 768, seed=32) * 768 ** -0.5).to(tw).cuda()
    pos, cls = (_rand(ntok, D, seed=33) * 0.3).cuda(), _rand(D, seed=34).cuda()
    assert hip.patch_embed_ok(16, w, torch.empty(1, D, device="cuda", dtype=tx))
    x = torch.full((F_ * ntok, D), 7.0, device="cuda", dtype=tx)
    hip.patch_embed(frames, w, pos, cls, x)
    patches = torch.empty(F_ * npatch, 768, device="cuda", dtype=tw)
    hip.im2col_patches(frames, patches, 16)
    y = torch.full((F_ * ntok, D), 7.0, device="cuda", dtype=tx)
    hip.gemm(patches, w, y, residual=pos, M=F_ * npatch, N=D, K=768, ldo=D, ldr=D, row_group=npatch, row_gap=1, row_off=1, res_mod=npatch, res_off=1)
    hip.cls_rows(y, cls, pos, F_, ntok, D)
    torch.cuda.synchronize()
    assert torch.equal(x.view(torch.int16), y.view(torch.int16)), maxdiff(x.float().cpu(), y.float().cpu())
    ref = (patches.float() @ w.float().t()).reshape(F_, npatch, D) + pos[1:]
    ref = torch.cat([(cls + pos[0]).expand(F_, 1, D), ref], 1).reshape(F_ * ntok, D)
    assert maxdiff(x.float().cpu(), ref.cpu()) < 6e-3


@pytest.mark.parametrize("F_,res,P,D", [(3, 224, 16, 768), (2, 224, 14, 1024), (5, 32, 16, 192), (2, 28, 14, 64)])
def test_strict_patch_embedding_front_end(hip, F_, res, P, D):
    """fp16_strict (round 6; few_shot.py:672-677): cfsar_im2col_patches_split keeps every fp32 pixel as [hi | lo | hi] fp16 words, ONE fp16 GEMM
    against [W_hi | W_hi | W_lo] gives the patch tokens in fp32, cfsar_embed_finish_pair adds class token / pos, applies ln_pre and writes the
    two-word stream.  Against the fp32 op on the UNROUNDED operands: 22-bit operands leave ~1e-6 relative (the one-pass fp16 GEMM: 3e-4)."""
    g = res // P
    npatch, ntok = g * g, g * g + 1
    kreal = 3 * P * P
    kpad = (kreal + 63) // 64 * 64
    frames = _rand(F_, 3, res, res, seed=61)
    w32 = torch.zeros(D, kpad)
    w32[:, :kreal] = _rand(D, kreal, seed=62) * kreal ** -0.5
    pos, cls = _rand(ntok, D, seed=63) * 0.3, _rand(D, seed=64)
    ln_w, ln_b = 1.0 + 0.1 * _rand(D, seed=65), 0.1 * _rand(D, seed=66)
    patches = torch.empty(F_ * npatch, 3 * kpad, device="cuda", dtype=torch.float16)
    hip.im2col_patches_split(frames.cuda(), patches, P)
    pt = frames.reshape(F_, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(F_ * npatch, kreal)
    hi = pt.half()
    lo = (pt - hi.float()).half()
    pc = patches.cpu()
    assert torch.equal(pc[:, :kreal], hi) and torch.equal(pc[:, 2 * kpad:2 * kpad + kreal], hi) and torch.equal(pc[:, kpad:kpad + kreal], lo)
    if kpad > kreal:
        assert float(pc[:, kreal:kpad].abs().max()) == 0.0 and float(pc[:, kpad + kreal:2 * kpad].abs().max()) == 0.0
    w_hi = w32.half()
    w3 = torch.cat([w_hi, w_hi, (w32 - w_hi.float()).half()], 1).contiguous().cuda()
    tok = torch.empty(F_ * npatch, D, device="cuda", dtype=torch.float32)
    hip.gemm(patches, w3, tok, M=F_ * npatch, N=D, K=3 * kpad, ldo=D)
    ref_tok = pt.double() @ w32[:, :kreal].double().t()
    scale = float(ref_tok.abs().max())
    assert maxdiff(tok.cpu(), ref_tok.float()) < 4e-6 * max(1.0, scale), (maxdiff(tok.cpu(), ref_tok.float()), scale)
    one_pass = hi.double() @ w_hi[:, :kreal].double().t()
    assert maxdiff(one_pass.float(), ref_tok.float()) > 20 * maxdiff(tok.cpu(), ref_tok.float())           # what the extra passes buy
    x_hi = torch.full((F_ * ntok, D), 7.0, device="cuda", dtype=torch.float16)
    x_lo = torch.full((F_ * ntok, D), 7.0, device="cuda", dtype=torch.float16)
    hip.embed_finish_pair(tok, cls.cuda(), pos.cuda(), ln_w.cuda(), ln_b.cuda(), x_hi, x_lo, F_, ntok, D)
    rows = torch.cat([cls.expand(F_, 1, D), tok.cpu().reshape(F_, npatch, D)], 1) + pos
    ref = orc.layer_norm(rows, ln_w, ln_b).reshape(F_ * ntok, D)
    got = x_hi.float().cpu() + x_lo.float().cpu()
    assert maxdiff(got, ref) < 3e-6 * max(1.0, float(ref.abs().max()))
    assert maxdiff(x_hi.float().cpu(), ref) <= 1.01 * 2.0 ** -11 * float(ref.abs().max())       # hi alone is the fp16 rounding of the row
    assert float(x_lo.float().abs().max()) <= 2.0 ** -11 * float(ref.abs().max())


@pytest.mark.parametrize("wd", ["bf16", "f16"])
@pytest.mark.parametrize("F_,res,D", [(6, 224, 1024), (5, 224, 192), (81, 224, 1024), (40, 56, 128)])
def test_patch_embed_fused_14x14_equals_the_gemm_on_the_padded_row_matrix_bitwise(hip, wd, F_, res, D):
    """cfsar_patch_embed for 14 x 14 patches (ViT-L/14): the K axis is laid out in padded rows, k' = (c 14 + dy) 16 + dx, 704 slots;
    the launch equals cfsar_gemm (row remap, pos residual) on the patch matrix in that layout + cfsar_cls_rows bit for bit (same roundings, same MFMA
    order), and the fp32 conv1 of the unrounded operands to the operands' 11 / 8 bits."""
    tw = {"bf16": torch.bfloat16, "f16": torch.float16}[wd]
    g = res // 14
    npatch, ntok = g * g, g * g + 1
    frames = _rand(F_, 3, res, res, seed=41)
    conv_w = _rand(D, 3, 14, 14, seed=42) * 588 ** -0.5
    pos, cls = (_rand(ntok, D, seed=43) * 0.3).cuda(), _rand(D, seed=44).cuda()
    w = hip.patch_embed_weight(conv_w.cuda(), 14, tw)
    assert tuple(w.shape) == (D, 704) and float(w.float().reshape(D, 44, 16)[:, :42, 14:].abs().max()) == 0.0 and float(w[:, 672:].float().abs().max()) == 0.0
    x = torch.full((F_ * ntok, D), 7.0, device="cuda", dtype=torch.float16)
    assert hip.patch_embed_ok(14, w, x)
    hip.patch_embed(frames.cuda(), w, pos, cls, x, patch=14)
    # the patch matrix in the kernel's column layout (pure data movement + the one rounding to the operand type)
    pt = frames.reshape(F_, 3, g, 14, g, 14).permute(0, 2, 4, 1, 3, 5)                                   # [F, gy, gx, c, dy, dx]
    pm = torch.zeros(F_, g, g, 3, 14, 16)
    pm[..., :14] = pt
    patches = torch.zeros(F_ * npatch, 704)
    patches[:, :672] = pm.reshape(F_ * npatch, 672)
    patches = patches.to(tw).cuda()
    y = torch.full((F_ * ntok, D), 7.0, device="cuda", dtype=torch.float16)
    hip.gemm(patches, w, y, residual=pos, M=F_ * npatch, N=D, K=704, ldo=D, ldr=D, row_group=npatch, row_gap=1, row_off=1, res_mod=npatch, res_off=1)
    hip.cls_rows(y, cls, pos, F_, ntok, D)
    torch.cuda.synchronize()
    if F_ * npatch >= 1024:                        # (the generic small-M GEMM sums K in another order)
        assert torch.equal(x.view(torch.int16), y.view(torch.int16)), maxdiff(x.float().cpu(), y.float().cpu())
    else:
        assert maxdiff(x.float().cpu(), y.float().cpu()) < 2e-3
    ref = torch.nn.functional.conv2d(frames, conv_w, stride=14).permute(0, 2, 3, 1).reshape(F_, npatch, D) + pos[1:].cpu()
    ref = torch.cat([(cls + pos[0]).cpu().expand(F_, 1, D), ref], 1).reshape(F_ * ntok, D)
    assert maxdiff(x.float().cpu(), ref) < (6e-3 if wd == "f16" else 3e-2)


def test_patch_embed_rejects_what_it_does_not_serve(hip):
    frames = _rand(2, 3, 28, 28, seed=35).cuda()
    w = _rand(64, 768, seed=36).to(torch.bfloat16).cuda()
    pos, cls = _rand(5, 64, seed=37).cuda(), _rand(64, seed=38).cuda()
    x = torch.empty(2 * 5, 64, device="cuda", dtype=torch.float16)
    with pytest.raises(RuntimeError, match="patch size 7"):
        hip.patch_embed(frames, w, pos.repeat(4, 1)[:17], cls, x.repeat(4, 1)[:34], patch=7)
    assert not hip.patch_embed_ok(7, w, x) and not hip.patch_embed_ok(16, w.float(), x) and not hip.patch_embed_ok(16, w, x.bfloat16())
    assert hip.patch_embed_ok(14, w, x) and not hip.patch_embed_ok(14, w[:, :640].contiguous(), x)          # 14 x 14: 704 padded-row slots
    with pytest.raises(RuntimeError, match="at least 10 rows"):                     # a stream too short for the frames: refused before the launch
        hip.patch_embed(_rand(2, 3, 32, 32, seed=39).cuda(), w, pos, cls, x[:9])
    with pytest.raises(RuntimeError, match="pos must be"):
        hip.patch_embed(_rand(2, 3, 64, 64, seed=39).cuda(), w, pos, cls, x)
    with pytest.raises(RuntimeError, match="fp16 residual stream"):
        hip.patch_embed(_rand(2, 3, 32, 32, seed=39).cuda(), w, pos, cls, x.bfloat16())


def _ref_attention(qkv, F_, ntok, D, heads):
    q, k, v = qkv.float().reshape(F_, ntok, 3, heads, 64).permute(2, 0, 3, 1, 4)
    att = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    return (att @ v).permute(0, 2, 1, 3).reshape(F_ * ntok, D)


@pytest.mark.parametrize("ntok", [5, 17, 197, 257])
@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_vit_attention(hip, ntok, dtype):
    F_, heads = 3, 2
    D = heads * 64
    td = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dtype]
    qkv = (_rand(F_ * ntok, 3 * D, seed=15) * 1.5).to(td)
    # spike one key against one query so the softmax is far from uniform somewhere
    ref = _ref_attention(qkv, F_, ntok, D, heads)
    out = torch.empty(F_ * ntok, D, device="cuda", dtype=td)
    hip.vit_attention(qkv.cuda(), out, F_, ntok, D, heads)
    tol = {"f32": 2e-5, "bf16": 3e-2, "f16": 4e-3}[dtype]                # 16-bit: P and the output are rounded to the operand type
    assert maxdiff(out.float().cpu(), ref) < tol


@pytest.mark.parametrize("ntok", [5, 17, 197, 257])
@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_vit_attention_cls(hip, ntok, dtype):
    """cfsar_vit_attention_cls == row 0 of every frame of the full attention (few_shot.py:623 for the one query the last block needs)."""
    F_, heads = 5, 3
    D = heads * 64
    td = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dtype]
    qkv = (_rand(F_ * ntok, 3 * D, seed=25) * 1.5).to(td)
    ref = _ref_attention(qkv, F_, ntok, D, heads).reshape(F_, ntok, D)[:, 0, :]
    out = torch.full((F_, D), float("nan"), device="cuda", dtype=td)
    hip.vit_attention_cls(qkv.cuda(), out, F_, ntok, D, heads)
    tol = {"f32": 2e-5, "bf16": 1.5e-2, "f16": 2e-3}[dtype]                # fp32 arithmetic on the stored operands; output rounded to td
    assert maxdiff(out.float().cpu(), ref) < tol


def test_class_text_logits(hip):
    nv, T, E, ncls = 10, 8, 512, 64
    feats, text, scale = _rand(nv, T, E, seed=16), _rand(ncls, E, seed=17), torch.tensor([1.7])
    out = torch.empty(nv, ncls, device="cuda")
    hip.class_text_logits(feats.cuda(), text.cuda(), scale.cuda(), out, nv, T, E)
    assert maxdiff(out.cpu(), orc.cos_sim(feats.mean(1), text) * scale) < 1e-5


@pytest.mark.parametrize("merge_before", [False, True])
def test_build_sequences_and_prototypes(hip, merge_before):
    B, way, shot, Q, T, E, ntest = 2, 5, 3, 5, 4, 64, 24
    S = way * shot
    feats = _rand(B, S + Q, T, E, seed=18)
    text = _rand(ntest, E, seed=19)
    lab = torch.stack([torch.arange(way).repeat_interleave(shot)[torch.randperm(S, generator=torch.Generator().manual_seed(b))]
                       for b in range(B)]).float()
    real = (lab * 3 + 2)                                                   # real class id per support video
    Sp = way if merge_before else S
    X = torch.empty(B * Q * T + B * Sp * (T + 1), E, device="cuda")
    hip.build_sequences(feats.cuda(), text.cuda(), lab.cuda(), real.cuda(), X, B, S, Q, T, E, way, merge_before)
    Xc = X.cpu()
    assert maxdiff(Xc[:B * Q * T].reshape(B, Q, T, E), feats[:, S:]) == 0.0
    sup = Xc[B * Q * T:].reshape(B, Sp, T + 1, E)
    for b in range(B):
        fs, ctx = feats[b, :S], text[real[b].long()].unsqueeze(1)
        if merge_before:
            fs, _ = orc.class_means(fs, lab[b])
            ctx, _ = orc.class_means(ctx, lab[b])
        assert maxdiff(sup[b], torch.cat([fs, ctx], 1)) < 1e-6
    protos = torch.empty(B, way, T, E, device="cuda")
    hip.prototypes(X[B * Q * T:], lab.cuda(), protos, B, S, Sp, T, E, way, merge_before)
    for b in range(B):
        ref = sup[b][:, :T]
        if not merge_before:
            ref, _ = orc.class_means(ref, lab[b])
        assert maxdiff(protos[b].cpu(), ref) < 1e-6


def test_seq_attention(hip):
    heads, hd, na, la, nb, lb = 8, 64, 5, 8, 5, 9
    inner = heads * hd
    rows = na * la + nb * lb
    qkv = _rand(rows, 3 * inner, seed=20)
    out = torch.empty(rows, inner, device="cuda")
    hip.seq_attention(qkv.cuda(), out, na, la, nb, lb, heads, hd, hd ** -0.5)
    got = out.cpu()
    r0 = 0
    for n, L in ((na, la), (nb, lb)):
        for s in range(n):
            blk = qkv[r0:r0 + L]
            q, k, v = [t.reshape(L, heads, hd).transpose(0, 1) for t in blk.split(inner, dim=1)]
            a = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1)
            ref = (a @ v).transpose(0, 1).reshape(L, inner)
            assert maxdiff(got[r0:r0 + L], ref) < 1e-5
            r0 += L


@pytest.mark.parametrize("T", [1, 4, 8, 16])
@pytest.mark.parametrize("single_direct", [False, True])
def test_cos_otam_logits(hip, T, single_direct):
    B, Q, way, E = 2, 5, 5, 512
    xq, pr = _rand(B, Q, T, E, seed=21), _rand(B, way, T, E, seed=22)
    logits = torch.empty(B, Q, way, device="cuda")
    dists = torch.empty(B, Q, way, T, T, device="cuda")
    hip.cos_otam_logits(xq.cuda(), pr.cuda(), logits, B, Q, way, T, E, 0.5, single_direct, dists_out=dists)
    for b in range(B):
        sim = orc.cos_sim(xq[b].reshape(Q * T, E), pr[b].reshape(way * T, E))
        d = (1 - sim).reshape(Q, T, way, T).permute(0, 2, 1, 3)
        cum = orc.otam_cum_dist(d)
        if not single_direct:
            cum = cum + orc.otam_cum_dist(d.transpose(-1, -2))
        assert maxdiff(dists[b].cpu(), d) < 1e-5
        assert maxdiff(logits[b].cpu(), -cum) < 1e-4


def test_otam_known_answers(hip):
    """SURVEY.md section 4 known answers through the HIP kernel: dists are produced from orthogonal / identical
    unit vectors so that 1 - cos takes known values."""
    T, E = 8, 64
    # identical unit vectors everywhere -> sim = 1/(1+0.01), dist = 1 - 1/1.01 everywhere
    x = torch.zeros(1, 1, T, E)
    x[..., 0] = 1.0
    logits = torch.empty(1, 1, 1, device="cuda")
    hip.cos_otam_logits(x.cuda(), x.cuda(), logits, 1, 1, 1, T, E, 0.5, True)
    dconst = 1.0 - 1.0 / 1.01
    ref = orc.otam_cum_dist(torch.full((1, 1, T, T), dconst))
    assert abs(float(logits.cpu()) + float(ref)) < 1e-5


# ---------------------------------------------------------------------------------------------- N3: RN50 tower ops
_TD = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
_TOL = {"f32": 2e-4, "bf16": 2e-2, "f16": 2.5e-3}           # of max |ref|: the operands are rounded first, so this is accumulation order + the output's rounding


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_gemm_ex_relu_and_bf16_residual(hip, dtype):
    """cfsar_gemm_ex: relu applied LAST, residual in the activation dtype (Bottleneck: relu(bn3(conv3) + identity),
    few_shot.py:213-226).  f16: the RN50 tower's fp16 mode (fp32 add, ONE rounding)."""
    td = _TD[dtype]
    for (M, N, K) in [(196 * 3, 256, 64), (49 * 5, 2048, 512), (1000, 64, 576)]:
        A = _rand(M, K, seed=1).to(td)
        W = _rand(N, K, seed=2, scale=K ** -0.5).to(td)
        bias = _rand(N, seed=3)
        res = _rand(M, N, seed=4).to(td)
        ref = torch.relu(A.float() @ W.float().t() + bias + res.float())
        out = torch.empty(M, N, device="cuda", dtype=td)
        hip.gemm(A.cuda(), W.cuda(), out, bias=bias.cuda(), residual=res.cuda(), relu=True)
        tol = _TOL[dtype]
        assert maxdiff(out.float().cpu(), ref) < tol * max(1.0, float(ref.abs().max()))
        assert float(out.float().min()) >= 0.0
        out2 = torch.empty(M, N, device="cuda", dtype=td)
        hip.gemm(A.cuda(), W.cuda(), out2, bias=bias.cuda(), relu=True)
        ref2 = torch.relu(A.float() @ W.float().t() + bias)
        assert maxdiff(out2.float().cpu(), ref2) < tol * max(1.0, float(ref2.abs().max()))


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("stride,C,H", [(1, 64, 14), (2, 3, 32), (1, 32, 9), (2, 8, 7)])
def test_conv3x3_as_im2col_gemm(hip, dtype, stride, C, H):
    """nchw_to_nhwc + im2col3x3_nhwc + GEMM with tap-major weights == nn.Conv2d(3, padding=1, stride) (bit-exact
    gather: the only arithmetic is in the GEMM)."""
    import torch.nn.functional as F
    td = _TD[dtype]
    Fn, W_, Co = 3, H + 2, 32
    x = _rand(Fn, C, H, W_, seed=5)
    w = _rand(Co, C, 3, 3, seed=6, scale=(9 * C) ** -0.5)
    xd = torch.empty(Fn, H, W_, C, device="cuda", dtype=td)
    hip.nchw_to_nhwc(x.cuda(), xd)
    assert torch.equal(xd.float().cpu(), x.to(td).float().permute(0, 2, 3, 1))
    Ho, Wo = (H - 1) // stride + 1, (W_ - 1) // stride + 1
    kq = 32 if dtype == "f32" else 64
    kpad = -(-9 * C // kq) * kq
    cols = torch.full((Fn * Ho * Wo, kpad), float("nan"), device="cuda", dtype=td)
    hip.im2col3x3(xd, cols, Fn, H, W_, C, stride)
    # gather reference (exact)
    xp = F.pad(x.to(td).float().permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))
    ref_cols = torch.zeros(Fn, Ho, Wo, kpad)
    for ky in range(3):
        for kx in range(3):
            ref_cols[..., (ky * 3 + kx) * C:(ky * 3 + kx + 1) * C] = \
                xp[:, ky:ky + (Ho - 1) * stride + 1:stride, kx:kx + (Wo - 1) * stride + 1:stride, :]
    assert torch.equal(cols.float().cpu().reshape(Fn, Ho, Wo, kpad), ref_cols)
    wt = torch.zeros(Co, kpad)
    wt[:, :9 * C] = w.permute(0, 2, 3, 1).reshape(Co, 9 * C)
    out = torch.empty(Fn * Ho * Wo, Co, device="cuda", dtype=torch.float32)
    hip.gemm(cols, wt.to(td).cuda(), out)
    ref = F.conv2d(x.to(td).float(), w.to(td).float(), stride=stride, padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    assert maxdiff(out.cpu(), ref) < (2e-4 if dtype == "f32" else 2e-2) * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_avgpool_and_attnpool_tokens(hip, dtype):
    td = _TD[dtype]
    Fn, H, W_, C = 3, 14, 14, 64
    x = _rand(Fn, H, W_, C, seed=7).to(td)
    out = torch.empty(Fn, H // 2, W_ // 2, C, device="cuda", dtype=td)
    hip.avgpool2x2(x.cuda(), out, Fn, H, W_, C)
    ref = x.float().reshape(Fn, H // 2, 2, W_ // 2, 2, C).mean((2, 4))
    assert maxdiff(out.float().cpu(), ref) < {"f32": 1e-6, "bf16": 1e-2, "f16": 1.5e-3}[dtype]
    # odd sizes floor like nn.AvgPool2d(2)
    xo = _rand(2, 7, 9, 8, seed=8).to(td)
    oo = torch.empty(2, 3, 4, 8, device="cuda", dtype=td)
    hip.avgpool2x2(xo.cuda(), oo, 2, 7, 9, 8)
    assert maxdiff(oo.float().cpu(), xo.float()[:, :6, :8].reshape(2, 3, 2, 4, 2, 8).mean((2, 4))) < {"f32": 1e-6, "bf16": 1e-2, "f16": 1.5e-3}[dtype]
    xs = _rand(2, 6, 4, 12, seed=18).to(td)                 # C % 8 != 0: the scalar kernel
    os_ = torch.empty(2, 3, 2, 12, device="cuda", dtype=td)
    hip.avgpool2x2(xs.cuda(), os_, 2, 6, 4, 12)
    assert maxdiff(os_.float().cpu(), xs.float().reshape(2, 3, 2, 2, 2, 12).mean((2, 4))) < {"f32": 1e-6, "bf16": 1e-2, "f16": 1.5e-3}[dtype]
    # AttentionPool2d tokens (few_shot.py:446-448): [mean ; x] + pos
    HW = 49
    t = _rand(Fn, HW, C, seed=9).to(td)
    pos = _rand(HW + 1, C, seed=10)
    tok = torch.empty(Fn, HW + 1, C, device="cuda", dtype=td)
    hip.attnpool_tokens(t.cuda(), pos.cuda(), tok, Fn, HW, C)
    ref_tok = torch.cat([t.float().mean(1, keepdim=True), t.float()], 1) + pos
    assert maxdiff(tok.float().cpu(), ref_tok) < {"f32": 2e-6, "bf16": 3e-2, "f16": 4e-3}[dtype]


def _gemm_epilogue_checks(hip, M, N, K, tag):
    """bf16 GEMM through the fused epilogues the ViT and RN50 towers use, against the fp32 product of the bf16-rounded operands
    (torch fp32 matmul on the device: a checker, not the thing under test)."""
    A = _rand(M, K, seed=11).to(torch.bfloat16).cuda()
    W = _rand(N, K, seed=12, scale=K ** -0.5).to(torch.bfloat16).cuda()
    bias = _rand(N, seed=13).cuda()
    ref0 = A.float() @ W.float().t() + bias
    scale = max(1.0, float(ref0.abs().max()))
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    hip.gemm(A, W, out, bias=bias)                                                       # QKV-like
    assert maxdiff(out.float(), ref0) < 2e-2 * scale, ("plain", tag, M, N, K)
    out.fill_(float("nan"))
    hip.gemm(A, W, out, bias=bias, act=hip.ACT_QUICKGELU)                                # c_fc-like
    assert maxdiff(out.float(), orc.quick_gelu(ref0)) < 4e-2 * scale, ("gelu", tag, M, N, K)
    x = _rand(M, N, seed=14).cuda()
    xr = x.clone()
    hip.gemm(A, W, x, bias=bias, residual=x)                                             # fp32 residual stream, in place
    assert maxdiff(x, ref0 + xr) < 2e-2 * scale, ("residual f32", tag, M, N, K)
    xh = _rand(M, N, seed=16).to(torch.float16).cuda()
    xh0 = xh.clone()
    hip.gemm(A, W, xh, bias=bias, residual=xh)                                           # fp16 residual stream (bf16 mode), in place
    assert maxdiff(xh.float(), ref0 + xh0.float()) < 2e-2 * scale, ("residual f16", tag, M, N, K)
    rb = _rand(M, N, seed=15).to(torch.bfloat16).cuda()
    hip.gemm(A, W, out, bias=bias, residual=rb, relu=True)                               # RN50 bottleneck tail
    assert maxdiff(out.float(), torch.relu(ref0 + rb.float())) < 2e-2 * scale, ("relu", tag, M, N, K)


# one shape per branch of the kernel policy in cfsar_gemm_ex (csrc/gemm.hip), every one with ragged M (and N where the kernel
# supports it): v1 128x128 | p3 256x128 (M >= 1024, few 256x256 tiles) | p12 one workgroup per tile (240 <= tiles < 512) |
# the persistent ViT kernel of gemm_vit.hip (>= 512 tiles; N % 64 == 0) incl. a long-K case, an N that is not a multiple of 256,
# and a grid with fewer tiles per workgroup than stages
@pytest.mark.parametrize("M,N,K", [(515, 260, 192), (777, 516, 768), (1300, 768, 3072), (20500, 768, 768),
                                   (44000, 768, 768), (33000, 1024, 128), (22100, 1600, 256), (16500, 2304, 3072)])
def test_gemm_policy_reaches_every_kernel(hip, M, N, K):
    _gemm_epilogue_checks(hip, M, N, K, "auto")


@pytest.mark.skipif(os.environ.get("CFSAR_DEV_LIB", "0") != "1", reason="developer library only (CFSAR_DEV_LIB=1)")
@pytest.mark.parametrize("variant", [1, 2, 10, 11, 12, 13, 20, 21, 22, 24, 25, 26, 28, 30, 36, 38, 40, 42])
def test_gemm_forced_variants_dev(hip, variant):
    """Developer build: every kernel / operand path / store policy forced on ragged shapes (incl. shapes the policy would not
    give it), plus the alternative tile walks of the ViT kernel."""
    L = hip.lib()
    try:
        for dbg in ([0, 256, 512 | 256, 1024] if variant >= 20 else [0]):
            L.cfsar_debug_set_gemm_variant(variant, dbg)
            for (M, N, K) in [(777, 512, 768), (1300, 768, 3072), (5000, 320, 192), (9000, 768, 128)]:
                _gemm_epilogue_checks(hip, M, N, K, (variant, dbg))
    finally:
        L.cfsar_debug_set_gemm_variant(0, 0)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("M,N,K", [(125440, 1024, 256), (31360, 2048, 512), (140000, 512, 128), (70001, 1024, 320)])
def test_rn50_conv3_residual_relu_on_the_persistent_kernel(hip, M, N, K, dtype):
    """relu(A W^T + bias + identity) in bf16 at RN50 batch scale (>= 512 tiles: the persistent kernel of csrc/gemm_vit.hip, bf16-residual
    instance; K = 128 its two-K-tile form) against the fp32 product of the bf16-rounded operands.  f16 (the tower's fp16 mode): the
    persistent 256 x 256 kernel of csrc/gemm.hip on fp16 operands, fp32 epilogue, one rounding."""
    td, tol = _TD[dtype], _TOL[dtype]
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.randn(M, K, device="cuda", generator=g).to(td)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(td)
    bias = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g).to(td)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=td)
    hip.gemm(A, W, out, bias=bias, residual=r, relu=True)
    ref = torch.relu(A.float() @ W.float().t() + bias + r.float())
    assert not torch.isnan(out.float()).any()
    assert float((out.float() - ref).abs().max()) < tol * max(1.0, float(ref.abs().max()))
    out3 = torch.full_like(out, float("nan"))                  # relu(A W^T + bias): conv1 of the bottlenecks (plain instance with ReLU)
    hip.gemm(A, W, out3, bias=bias, relu=True)
    ref3 = torch.relu(A.float() @ W.float().t() + bias)
    assert float((out3.float() - ref3).abs().max()) < tol * max(1.0, float(ref3.abs().max()))
    out2 = torch.full_like(out, float("nan"))                  # no ReLU
    hip.gemm(A, W, out2, bias=bias, residual=r)
    ref2 = A.float() @ W.float().t() + bias + r.float()
    assert float((out2.float() - ref2).abs().max()) < tol * max(1.0, float(ref2.abs().max()))
    if dtype == "f16":                                         # fp32 output (the attention pool's k / v GEMM)
        o32 = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32)
        hip.gemm(A, W, o32, bias=bias)
        assert float((o32 - (A.float() @ W.float().t() + bias)).abs().max()) < 2e-4 * max(1.0, float(ref3.abs().max()))


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("C,H,W_,Co,res", [(32, 12, 10, 64, False), (32, 11, 13, 32, True), (64, 9, 14, 64, False), (128, 7, 7, 128, True), (8, 5, 6, 260, True), (64, 10, 9, 256, False)])
def test_conv3x3_implicit_gemm(hip, C, H, W_, Co, res, dtype):
    """cfsar_conv3x3_nhwc (patch gather inside the GEMM operand staging) == nn.Conv2d(3, padding=1) + bias (+ residual) +
    ReLU on bf16-rounded operands; ragged M (F*H*W not a multiple of 256), ragged N, image borders, K padding."""
    import torch.nn.functional as F
    td, tol = _TD[dtype], _TOL[dtype]
    Fn = 5
    x = _rand(Fn, C, H, W_, seed=21).to(td)
    w = _rand(Co, C, 3, 3, seed=22, scale=(9 * C) ** -0.5).to(td)
    bias = _rand(Co, seed=23)
    kpad = -(-9 * C // 64) * 64
    wt = torch.zeros(Co, kpad, dtype=td)
    wt[:, :9 * C] = w.permute(0, 2, 3, 1).reshape(Co, 9 * C)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()                                       # NHWC
    ref = F.conv2d(x.float(), w.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, Co) + bias
    r = _rand(Fn * H * W_, Co, seed=24).to(td) if res else None
    if res:
        ref = ref + r.float()
    ref = torch.relu(ref)
    out = torch.full((Fn * H * W_, Co), float("nan"), device="cuda", dtype=td)
    hip.conv3x3(xd, wt.cuda(), out, Fn, H, W_, C, bias=bias.cuda(), residual=r.cuda() if res else None, relu=True)
    assert maxdiff(out.float().cpu(), ref) < tol * max(1.0, float(ref.abs().max()))
    if dtype == "f16":
        with pytest.raises(RuntimeError):                      # fp16 activations write fp16 outputs
            hip.conv3x3(xd, wt.cuda(), torch.empty(Fn * H * W_, Co, device="cuda", dtype=torch.float32), Fn, H, W_, C)
        return
    out32 = torch.empty(Fn * H * W_, Co, device="cuda", dtype=torch.float32)
    hip.conv3x3(xd, wt.cuda(), out32, Fn, H, W_, C, bias=bias.cuda())
    ref2 = F.conv2d(x.float(), w.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, Co) + bias
    assert maxdiff(out32.cpu(), ref2) < 2e-3 * max(1.0, float(ref2.abs().max()))


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("C,Co,Fn,H,W_", [(32, 32, 24, 112, 112), (32, 64, 7, 112, 112), (64, 64, 90, 56, 56), (32, 64, 3, 9, 127),
                                         (64, 64, 2, 33, 5), (32, 32, 2, 8, 8), (64, 64, 3, 6, 63)])
def test_conv3x3_direct_kernel(hip, C, Co, Fn, H, W_, dtype):
    """The direct kernel behind cfsar_conv3x3_nhwc for Cin, Cout in {32, 64} (csrc/conv.hip: LDS pixel ring + weights in registers) ==
    relu(nn.Conv2d(3, padding=1) + bias) on bf16-rounded operands: many tiles per workgroup (the ring wraps several times), the
    widest image the halo allows (W = 127), images narrower than the halo, a ragged last tile, frame borders inside a tile."""
    import torch.nn.functional as F
    td, tol = _TD[dtype], _TOL[dtype]
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(Fn, C, H, W_, generator=g)).to(td).cuda()
    w = (torch.randn(Co, C, 3, 3, generator=g) * (9 * C) ** -0.5).to(td).cuda()
    bias = torch.randn(Co, generator=g).cuda()
    kpad = -(-9 * C // 64) * 64
    wt = torch.zeros(Co, kpad, dtype=td, device="cuda")
    wt[:, :9 * C] = w.permute(0, 2, 3, 1).reshape(Co, 9 * C)
    xd = x.permute(0, 2, 3, 1).contiguous()
    ref = torch.relu(F.conv2d(x.float(), w.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, Co) + bias)
    out = torch.full((Fn * H * W_, Co), float("nan"), device="cuda", dtype=td)
    hip.conv3x3(xd, wt, out, Fn, H, W_, C, bias=bias, relu=True)
    torch.cuda.synchronize()
    d = (out.float() - ref).abs()
    assert not torch.isnan(out.float()).any()
    assert float(d.max()) < tol * max(1.0, float(ref.abs().max())), (float(d.max()), int(d.argmax()) // Co)
    out2 = torch.full_like(out, float("nan"))                    # no ReLU, no bias
    hip.conv3x3(xd, wt, out2, Fn, H, W_, C)
    ref2 = F.conv2d(x.float(), w.float(), padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    assert float((out2.float() - ref2).abs().max()) < tol * max(1.0, float(ref2.abs().max()))


@pytest.mark.parametrize("T,heads,hd", [(50, 32, 64), (5, 4, 8), (82, 2, 128)])
def test_attnpool_attend_single_query(hip, T, heads, hd):
    """cfsar_attnpool_attend == row 0 of softmax(scale q k^T) v per head (the only row AttentionPool2d returns,
    few_shot.py:450-469), fp32."""
    Fn, C = 3, heads * hd
    q = _rand(Fn, C, seed=31)
    kv = _rand(Fn * T, 2 * C, seed=32)
    k = kv[:, :C].reshape(Fn, T, heads, hd).permute(0, 2, 1, 3)
    v = kv[:, C:].reshape(Fn, T, heads, hd).permute(0, 2, 1, 3)
    qh = q.reshape(Fn, heads, 1, hd) * hd ** -0.5
    ref = (torch.softmax(qh @ k.transpose(-1, -2), -1) @ v).reshape(Fn, C)
    out = torch.empty(Fn, C, device="cuda")
    hip.attnpool_attend(q.cuda(), kv.cuda(), out, Fn, T, heads, hd, hd ** -0.5)
    assert maxdiff(out.cpu(), ref) < 2e-5


@pytest.mark.parametrize("Co,H,W_", [(32, 24, 20), (8, 9, 11), (64, 6, 6)])
def test_stem_conv_direct(hip, Co, H, W_):
    """cfsar_stem_conv3x3_s2 == relu(conv2d(x, w, stride=2, padding=1) + b) from NCHW fp32 frames to NHWC (fp32 arithmetic)."""
    import torch.nn.functional as F
    x = _rand(3, 3, H, W_, seed=41)
    w = _rand(Co, 3, 3, 3, seed=42, scale=27 ** -0.5)
    b = _rand(Co, seed=43)
    ref = torch.relu(F.conv2d(x, w, b, stride=2, padding=1)).permute(0, 2, 3, 1).contiguous()
    Ho, Wo = ref.shape[1], ref.shape[2]
    out = torch.empty(3 * Ho * Wo, Co, device="cuda", dtype=torch.float32)
    hip.stem_conv(x.cuda(), w.cuda(), b.cuda(), out)
    assert maxdiff(out.cpu().reshape(ref.shape), ref) < 1e-5
    out16 = torch.empty(3 * Ho * Wo, Co, device="cuda", dtype=torch.bfloat16)
    hip.stem_conv(x.cuda(), w.cuda(), b.cuda(), out16)
    assert maxdiff(out16.float().cpu().reshape(ref.shape), ref) < 2e-2
    outh = torch.empty(3 * Ho * Wo, Co, device="cuda", dtype=torch.float16)                 # the tower's fp16 mode
    hip.stem_conv(x.cuda(), w.cuda(), b.cuda(), outh)
    assert torch.equal(outh.cpu().reshape(ref.shape), out.cpu().reshape(ref.shape).half())   # the same fp32 sums, rounded once


def test_fp16_residual_stream_ops(hip):
    """The fp16 residual stream of the bf16 mode: GEMM with fp16 output + fp16 (or fp32) residual on every kernel that serves it
    (p12 / p3 / v1, incl. the patch-embed row scatter), LayerNorm with fp16 input (bf16 / fp16 / fp32 output), cls rows."""
    for (M, N, K) in [(1300, 768, 768), (200, 260, 192), (61 * 256, 1024, 128)]:       # p3, v1, p12 (61 x 4 = 244 tiles)
        A = _rand(M, K, seed=51).to(torch.bfloat16)
        W = _rand(N, K, seed=52, scale=K ** -0.5).to(torch.bfloat16)
        bias = _rand(N, seed=53)
        x0 = _rand(M, N, seed=54, scale=4.0).to(torch.float16)
        ref = A.float() @ W.float().t() + bias + x0.float()
        x = x0.clone().cuda()
        hip.gemm(A.cuda(), W.cuda(), x, bias=bias.cuda(), residual=x)                    # in place on the stream
        assert x.dtype == torch.float16
        assert maxdiff(x.float().cpu(), ref) < 2e-2 * max(1.0, float(ref.abs().max())), (M, N, K)
    # patch-embed form: fp16 output rows scattered behind each frame's class token, fp32 residual (positional embedding)
    Fn, npatch, D, Kp = 3, 49, 256, 192
    P = _rand(Fn * npatch, Kp, seed=55).to(torch.bfloat16)
    Wp = _rand(D, Kp, seed=56, scale=Kp ** -0.5).to(torch.bfloat16)
    pos = _rand(npatch + 1, D, seed=57)
    xs = torch.zeros(Fn * (npatch + 1), D, device="cuda", dtype=torch.float16)
    hip.gemm(P.cuda(), Wp.cuda(), xs, residual=pos.cuda(), M=Fn * npatch, N=D, K=Kp, ldo=D, ldr=D, row_group=npatch, row_gap=1,
             row_off=1, res_mod=npatch, res_off=1)
    refp = (P.float() @ Wp.float().t()).reshape(Fn, npatch, D) + pos[1:]
    got = xs.float().cpu().reshape(Fn, npatch + 1, D)
    assert maxdiff(got[:, 1:], refp) < 2e-2 * max(1.0, float(refp.abs().max()))
    assert float(got[:, 0].abs().max()) == 0.0
    cls = _rand(D, seed=58)
    hip.cls_rows(xs, cls.cuda(), pos.cuda(), Fn, npatch + 1, D)
    assert maxdiff(xs.float().cpu().reshape(Fn, npatch + 1, D)[:, 0], (cls + pos[0]).expand(Fn, D)) < 4e-3
    # LayerNorm on fp16 rows
    rows, D = 333, 768
    xh = _rand(rows, D, seed=59, scale=3.0).to(torch.float16)
    w, b = _rand(D, seed=60), _rand(D, seed=61)
    ref_ln = torch.nn.functional.layer_norm(xh.float(), (D,), w, b, 1e-5)
    for od, tol in ((torch.float32, 1e-4), (torch.float16, 4e-3), (torch.bfloat16, 3e-2)):
        out = torch.empty(rows, D, device="cuda", dtype=od)
        hip.layernorm(xh.cuda(), out, w.cuda(), b.cuda(), rows, D)
        assert maxdiff(out.float().cpu(), ref_ln) < tol * max(1.0, float(ref_ln.abs().max())), od


# ------------------------------------------------------------------------------------------------ LayerNorm folded into the GEMMs
@pytest.mark.parametrize("M,N,K,act", [(777, 384, 128, "none"), (900, 320, 192, "none"), (5000, 2304, 768, "none"), (44000, 3072, 768, "gelu"),
                                       (300, 512, 1024, "gelu"), (20500, 768, 256, "none"), (15760, 2304, 768, "none"), (31520, 3072, 768, "gelu")])
def test_gemm_lnfold_matches_layernorm_then_gemm(hip, M, N, K, act):
    _check_lnfold(hip, M, N, K, act)


def _check_lnfold(hip, M, N, K, act, od=torch.bfloat16):
    """cfsar_row_stats + cfsar_gemm_lnfold == act(F.layer_norm(x) @ W.T + b) (few_shot.py:605-611 + :626-628 / :636-640) on the
    raw fp16 stream: the reference of the folded form is the UNFOLDED fp32 computation, so the test covers the algebra (Wg, c, d,
    the rank-1 mean term, the 1/std row scale) and not just the kernel.  Rows get different means / scales (column offsets,
    a few large-magnitude channels like CLIP's outlier dimensions) so that a wrong row statistic cannot hide."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, K, generator=g) * (0.5 + torch.rand(M, 1, generator=g) * 3.0) + torch.randn(M, 1, generator=g) * 2.0
    x[:, 5] += 25.0
    x[:, K // 2] -= 12.0
    x = x.to(torch.float16).cuda()
    W = (_rand(N, K, seed=4, scale=K ** -0.5)).cuda()
    gamma = (1.0 + 0.5 * _rand(K, seed=5)).cuda()
    beta = (0.3 * _rand(K, seed=6)).cuda()
    bias = _rand(N, seed=7).cuda()
    ref = F.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ W.t() + bias
    if act == "gelu":
        ref = orc.quick_gelu(ref)
    Wg = (W * gamma[None, :]).to(torch.float16).contiguous()
    c = Wg.double().sum(1).float()
    d = (W.double() @ beta.double() + bias.double()).float()
    rstat = torch.empty(M, 4, device="cuda")
    hip.row_stats(x, rstat, M, K)
    mean, var = x.float().mean(1), x.float().var(1, unbiased=False)
    assert maxdiff(rstat[:, 0], mean) < 1e-4 * max(1.0, float(mean.abs().max()))
    assert maxdiff(rstat[:, 2], torch.rsqrt(var + 1e-5)) < 2e-4 * float(torch.rsqrt(var + 1e-5).max())
    out = torch.full((M, N), float("nan"), device="cuda", dtype=od)
    hip.gemm_lnfold(x, Wg, out, c, d, rstat, act=hip.ACT_QUICKGELU if act == "gelu" else hip.ACT_NONE)
    scale = max(1.0, float(ref.abs().max()))
    assert maxdiff(out.float(), ref) < 2e-2 * scale, (M, N, K, act)
    # tighter, against the same math with the operands the kernel sees (fp16 x, fp16 Wg): isolates the kernel from fp16 rounding
    ref2 = ((x.float() @ Wg.float().t()) - mean[:, None] * c[None, :]) * torch.rsqrt(var + 1e-5)[:, None] + d
    if act == "gelu":
        ref2 = orc.quick_gelu(ref2)
    assert maxdiff(out.float(), ref2) < (6e-3 if od == torch.bfloat16 else 1.5e-3) * scale, (M, N, K, act)    # output rounding (2^-9 / 2^-12) dominates


@pytest.mark.parametrize("M,N,K,act,od", [(15760, 2304, 768, "none", torch.bfloat16), (7880, 3072, 768, "gelu", torch.bfloat16),
                                          (40, 768, 768, "none", torch.bfloat16), (15760, 3072, 768, "gelu", torch.float16),
                                          (12850, 3072, 1024, "none", torch.bfloat16), (44000, 2304, 768, "none", torch.bfloat16),
                                          (900, 384, 256, "gelu", torch.bfloat16)])
def test_gemm_lnfold_partials_matches_finalize_then_lnfold(hip, M, N, K, act, od):
    """cfsar_gemm_lnfold_partials (statistics finalized inside the 192-row GEMM instances from the producer's partials; two launches
    from inside the library at batch scale and for other widths) == cfsar_ln_stats_finalize + cfsar_gemm_lnfold on the same partials."""
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(M, K, generator=g) * (0.5 + 2.0 * torch.rand(M, 1, generator=g)) + torch.randn(M, 1, generator=g)).to(torch.float16).cuda()
    S = K // 64
    xs = x.float().reshape(M, S, 64)
    part = torch.stack([xs.sum(2), (xs * xs).sum(2)], 2).contiguous()                   # what cfsar_gemm_residual_stats writes
    Wg = _rand(N, K, seed=12, scale=K ** -0.5).to(torch.float16).cuda()
    c, d = Wg.double().sum(1).float(), _rand(N, seed=13).cuda()
    a = hip.ACT_QUICKGELU if act == "gelu" else hip.ACT_NONE
    rstat = torch.empty(M, 4, device="cuda")
    hip.ln_stats_finalize(part, rstat, M, S, K)
    ref = torch.full((M, N), float("nan"), device="cuda", dtype=od)
    hip.gemm_lnfold(x, Wg, ref, c, d, rstat, act=a)
    ws = torch.full((M, 4), float("nan"), device="cuda")
    out = torch.full((M, N), float("nan"), device="cuda", dtype=od)
    hip.gemm_lnfold_partials(x, Wg, out, c, d, part, S, ws, act=a)
    assert not torch.isnan(out.float()).any()
    # the in-kernel finalize repeats cfsar_ln_stats_finalize's arithmetic in its order: bit-identical outputs, so an episode's logits do
    # not depend on which instance (192-row fused / 256-row two-launch) served it
    assert torch.equal(out, ref), float((out.float() - ref.float()).abs().max())


@pytest.mark.parametrize("M,N,K", [(777, 128, 128), (1300, 192, 192), (5000, 768, 768), (44000, 768, 3072), (20500, 1024, 256),
                                   (15760, 768, 3072), (15760, 768, 768), (70000, 768, 768)])    # 15 760 rows (one episode): 192-row tiles, several per workgroup
def test_gemm_residual_stats_and_finalize(hip, M, N, K):
    _check_residual_stats(hip, M, N, K)


@pytest.mark.parametrize("M,N,K", [(777, 128, 128), (5000, 768, 768), (30000, 768, 3072), (15760, 768, 3072)])
def test_gemm_residual_stats_fp16_operands(hip, M, N, K):
    _check_residual_stats(hip, M, N, K, td=torch.float16)


@pytest.mark.parametrize("M,N,K,act", [(777, 384, 128, "none"), (5000, 2304, 768, "none"), (30000, 3072, 768, "gelu"), (15760, 2304, 768, "none")])
def test_gemm_lnfold_fp16_output(hip, M, N, K, act):
    _check_lnfold(hip, M, N, K, act, od=torch.float16)


def test_gemm_fp16_operands_patch_embed_and_small(hip):
    """cfsar_gemm with fp16 operands (the fp16 numerics mode): the patch-embed scatter (row remap + positional residual into the fp16
    stream; 256x128 kernel from 1 024 rows on, 128x128 kernel below) and a plain fp32-output launch."""
    for F_ in (2, 9):
        npatch, D, K = 196, 256, 768
        patches = _rand(F_ * npatch, K, seed=31).to(torch.float16).cuda()
        w = _rand(D, K, seed=32, scale=K ** -0.5).to(torch.float16).cuda()
        pos = _rand(npatch + 1, D, seed=33).cuda()
        x = torch.zeros(F_ * (npatch + 1), D, device="cuda", dtype=torch.float16)
        hip.gemm(patches, w, x, residual=pos, M=F_ * npatch, N=D, K=K, ldo=D, ldr=D, row_group=npatch, row_gap=1, row_off=1,
                 res_mod=npatch, res_off=1)
        ref = (patches.float() @ w.float().t()).reshape(F_, npatch, D) + pos[1:]
        got = x.float().reshape(F_, npatch + 1, D)
        assert maxdiff(got[:, 1:], ref) < 4e-3 * max(1.0, float(ref.abs().max())), F_
        assert torch.all(got[:, 0] == 0)
    A = _rand(300, 256, seed=34).to(torch.float16).cuda()
    W = _rand(192, 256, seed=35, scale=1.0 / 16).to(torch.float16).cuda()
    b = _rand(192, seed=36).cuda()
    out = torch.empty(300, 192, device="cuda")
    hip.gemm(A, W, out, bias=b)
    assert maxdiff(out, A.float() @ W.float().t() + b) < 1e-4


def test_vit_gemm_random_shapes(hip):
    """The LN-folded and the residual + statistics GEMMs on 14 seeded random shapes: ragged M (last row band partial, also M < one
    tile), N any multiple of 64 (partial column tiles), K any multiple of 64 from 128 (K = 128 is its own instance; both operand paths: K <= 1024 LDS-DMA,
    longer K register-staged)."""
    import random
    rng = random.Random(20260928)
    for i in range(14):
        M = rng.choice([rng.randint(1, 300), rng.randint(300, 3000), rng.randint(3000, 9000)])
        N = 64 * rng.randint(1, 40)
        K = 64 * rng.choice([2, 3, 5, 8, 12, 16, 17, 24, 40])
        if i % 2 == 0:
            _check_lnfold(hip, M, N, K, "gelu" if i % 4 == 0 else "none")
        else:
            _check_residual_stats(hip, M, N, K)


def _check_residual_stats(hip, M, N, K, td=torch.bfloat16):
    """cfsar_gemm_residual_stats: x += A W^T + b in place on the fp16 stream, and its partial statistics, finalized by
    cfsar_ln_stats_finalize, are the LayerNorm statistics of the NEW (stored, fp16-rounded) x.  td: operand type (bf16, or fp16 in the
    fp16 numerics mode)."""
    A = _rand(M, K, seed=11).to(td).cuda()
    W = _rand(N, K, seed=12, scale=K ** -0.5).to(td).cuda()
    bias = _rand(N, seed=13).cuda()
    x = (_rand(M, N, seed=14) * 2.0 + 1.5).to(torch.float16).cuda()
    x0 = x.clone()
    part = torch.full((M, N // 64, 2), float("nan"), device="cuda")
    hip.gemm_residual_stats(A, W, x, bias, part)
    ref = x0.float() + A.float() @ W.float().t() + bias
    assert maxdiff(x.float(), ref) < 2e-2 * max(1.0, float(ref.abs().max()))
    xs = x.float()
    assert maxdiff(part[:, :, 0], xs.reshape(M, N // 64, 64).sum(2)) < 1e-3 * 64
    assert maxdiff(part[:, :, 1], (xs * xs).reshape(M, N // 64, 64).sum(2)) < 1e-3 * 64 * float(xs.abs().max()) ** 2
    rstat = torch.empty(M, 4, device="cuda")
    hip.ln_stats_finalize(part, rstat, M, N // 64, N)
    assert maxdiff(rstat[:, 0], xs.mean(1)) < 1e-4 * max(1.0, float(xs.mean(1).abs().max()))
    assert maxdiff(rstat[:, 1], torch.sqrt(xs.var(1, unbiased=False) + 1e-5)) < 1e-3
    x2 = x0.clone()
    hip.gemm_residual_stats(A, W, x2, bias, None)                    # statistics are optional
    assert torch.equal(x2, x)


# ------------------------------------------------------------------------------------------------ two concurrent streams
def test_vit_gemms_are_bit_stable_under_a_second_stream(hip):
    """The single-episode path runs the support and the query frames as two concurrent forwards on two HIP streams
    (engine.py: ClipFsarEngine.dual_frames).  Every ViT-block GEMM must give bit-identical results whether or not a second
    instance shares the chip.  Regression test for a fault seen in round 2: builds of the LN-folded QKV kernel that used packed-fp32
    VALU instructions returned stale values in lanes 48-63 of single accumulator registers -- in one build about once per 100
    concurrent launches and never alone (docs/history/design_r01-r03.md "A fault worth recording"; csrc/gemm_vit.hip is now compiled without those
    instructions; tools/stream_stress.py is the long version of this test)."""
    F_, N, D = 40, 197, 768
    M = F_ * N

    def make(seed):
        x16 = (_rand(M, D, seed=seed) * 1.5 + 0.3).to(torch.float16).cuda()
        rstat = torch.empty(M, 4, device="cuda")
        hip.row_stats(x16, rstat, M, D)
        Wq = _rand(3 * D, D, seed=seed + 1, scale=D ** -0.5).to(torch.float16).cuda()
        Wf = _rand(4 * D, D, seed=seed + 2, scale=D ** -0.5).to(torch.float16).cuda()
        cq, dq, cf, df = (_rand(n, seed=seed + 3 + i).cuda() for i, n in enumerate((3 * D, 3 * D, 4 * D, 4 * D)))
        qkv = torch.empty(M, 3 * D, device="cuda", dtype=torch.bfloat16)
        u = torch.empty(M, 4 * D, device="cuda", dtype=torch.bfloat16)
        o = _rand(M, D, seed=seed + 8).to(torch.bfloat16).cuda()
        Wo = _rand(D, D, seed=seed + 9, scale=D ** -0.5).to(torch.bfloat16).cuda()
        Wp = _rand(D, 4 * D, seed=seed + 10, scale=(4 * D) ** -0.5).to(torch.bfloat16).cuda()
        uin = _rand(M, 4 * D, seed=seed + 11).to(torch.bfloat16).cuda()
        bo = _rand(D, seed=seed + 12).cuda()
        x0 = _rand(M, D, seed=seed + 13).to(torch.float16).cuda()
        xa, xb = x0.clone(), x0.clone()
        pa, pb = torch.empty(M, D // 64, 2, device="cuda"), torch.empty(M, D // 64, 2, device="cuda")

        def run():
            hip.gemm_lnfold(x16, Wq, qkv, cq, dq, rstat, M=M)
            hip.gemm_lnfold(x16, Wf, u, cf, df, rstat, act=hip.ACT_QUICKGELU, M=M)
            xa.copy_(x0)
            hip.gemm_residual_stats(o, Wo, xa, bo, pa, M=M)
            xb.copy_(x0)
            hip.gemm_residual_stats(uin, Wp, xb, bo, pb, M=M)
        return run, lambda: [qkv, u, xa, pa, xb, pb]

    runs = [make(100), make(200)]
    refs = []
    for run, outs in runs:
        run()
        torch.cuda.synchronize()
        refs.append([t.clone() for t in outs()])
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for it in range(12):
        torch.cuda.synchronize()
        for st, (run, _) in zip(streams, runs):
            with torch.cuda.stream(st):
                run()
                run()
        torch.cuda.synchronize()
        for (run, outs), ref in zip(runs, refs):
            for name, t, r in zip(("qkv", "c_fc", "x_out", "part_out", "x_proj", "part_proj"), outs(), ref):
                assert torch.equal(t, r), (it, name, float((t.float() - r.float()).abs().max()))


def _two_stream_bit_stability(make, names, iters=8):
    """Two independent instances of a launch sequence: reference results alone, then both concurrently on two streams, `iters` times."""
    runs = [make(100), make(200)]
    refs = []
    for run, outs in runs:
        run()
        torch.cuda.synchronize()
        refs.append([t.clone() for t in outs()])
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for it in range(iters):
        torch.cuda.synchronize()
        for st, (run, _) in zip(streams, runs):
            with torch.cuda.stream(st):
                run()
                run()
        torch.cuda.synchronize()
        for (run, outs), ref in zip(runs, refs):
            for name, t, r in zip(names, outs(), ref):
                assert torch.equal(t, r), (it, name, float((t.float() - r.float()).abs().max()))


def test_p12_gemms_are_bit_stable_under_a_second_stream(hip):
    """gemm_kernel_p12 (gemm.hip) serves the ViT-block GEMMs between 240 and 512 output tiles -- out_proj / c_proj of a 2-episode
    step (124 row bands x 3) with the fp16 residual stream, QKV-shaped plain launches of ~30 row bands.  Same accumulator-scaling
    epilogue patterns as the kernel that showed the round-2 stale-lanes fault; compiled without packed-fp32 ops since round 3."""
    D = 768
    M = 160 * 197                                   # 124 bands: N = 768 -> 372 tiles (p12), N = 2304 -> 1 116 (vit kernel)
    Mq = 8000                                       # 32 bands x 9 = 288 tiles: p12 with bf16 output

    def make(seed):
        o = _rand(M, D, seed=seed).to(torch.bfloat16).cuda()
        uin = _rand(M, 4 * D, seed=seed + 1).to(torch.bfloat16).cuda()
        Wo = _rand(D, D, seed=seed + 2, scale=D ** -0.5).to(torch.bfloat16).cuda()
        Wp = _rand(D, 4 * D, seed=seed + 3, scale=(4 * D) ** -0.5).to(torch.bfloat16).cuda()
        bo = _rand(D, seed=seed + 4).cuda()
        x0 = _rand(M, D, seed=seed + 5).to(torch.float16).cuda()
        xa, xb = x0.clone(), x0.clone()
        h = _rand(Mq, D, seed=seed + 6).to(torch.bfloat16).cuda()
        Wq = _rand(3 * D, D, seed=seed + 7, scale=D ** -0.5).to(torch.bfloat16).cuda()
        bq = _rand(3 * D, seed=seed + 8).cuda()
        qkv = torch.empty(Mq, 3 * D, device="cuda", dtype=torch.bfloat16)
        ug = torch.empty(Mq, 3 * D, device="cuda", dtype=torch.bfloat16)

        def run():
            xa.copy_(x0)
            hip.gemm(o, Wo, xa, bias=bo, residual=xa)
            xb.copy_(x0)
            hip.gemm(uin, Wp, xb, bias=bo, residual=xb)
            hip.gemm(h, Wq, qkv, bias=bq)
            hip.gemm(h, Wq, ug, bias=bq, act=hip.ACT_QUICKGELU)
        return run, lambda: [xa, xb, qkv, ug]
    _two_stream_bit_stability(make, ("out_proj p12 f16", "c_proj p12 f16", "qkv p12", "gelu p12"))


def test_patch_embed_attention_and_rn50_conv_are_bit_stable_under_a_second_stream(hip):
    """The other hot kernels of the towers under the same two-stream regime: the patch-embed GEMM (gemm_kernel_p3, row-scatter + positional
    epilogue into the fp16 stream), the attention kernel at 197 tokens (ViT-B/16) and at 257 tokens / 16 heads (vit_attn_bf16_kernel<9,
    ...>, ViT-L/14) and the RN50 implicit-GEMM 3x3 convolutions (p3 CONV with 64 and 128 output channels, p10 CONV with 256)."""
    def make(seed):
        F_, npatch, D = 80, 196, 768
        patches = _rand(F_ * npatch, D, seed=seed).to(torch.bfloat16).cuda()
        wp = _rand(D, D, seed=seed + 1, scale=D ** -0.5).to(torch.bfloat16).cuda()
        pos = _rand(npatch + 1, D, seed=seed + 2).cuda()
        x = torch.zeros(F_ * (npatch + 1), D, device="cuda", dtype=torch.float16)
        qkvB = _rand(40 * 197, 3 * 768, seed=seed + 3).to(torch.bfloat16).cuda()
        oB = torch.empty(40 * 197, 768, device="cuda", dtype=torch.bfloat16)
        qkvL = _rand(24 * 257, 3 * 1024, seed=seed + 4).to(torch.bfloat16).cuda()
        oL = torch.empty(24 * 257, 1024, device="cuda", dtype=torch.bfloat16)
        convs = []
        for i, (C, H, Co) in enumerate(((64, 56, 64), (128, 28, 128), (256, 14, 256))):
            Fn = 16
            xc = _rand(Fn * H * H, C, seed=seed + 10 + i).to(torch.bfloat16).cuda()
            w = _rand(Co, 9 * C, seed=seed + 20 + i, scale=(9 * C) ** -0.5).to(torch.bfloat16).cuda()
            b = _rand(Co, seed=seed + 30 + i).cuda()
            out = torch.empty(Fn * H * H, Co, device="cuda", dtype=torch.bfloat16)
            convs.append((xc, w, b, out, Fn, H, C))

        def run():
            hip.gemm(patches, wp, x, residual=pos, M=F_ * npatch, N=D, K=D, ldo=D, ldr=D, row_group=npatch, row_gap=1, row_off=1,
                     res_mod=npatch, res_off=1)
            hip.vit_attention(qkvB, oB, 40, 197, 768, 12)
            hip.vit_attention(qkvL, oL, 24, 257, 1024, 16)
            for xc, w, b, out, Fn, H, C in convs:
                hip.conv3x3(xc, w, out, Fn, H, H, C, bias=b, relu=True)
        return run, lambda: [x, oB, oL] + [c[3] for c in convs]
    _two_stream_bit_stability(make, ("patch embed p3", "attention 197", "attention 257 (ViT-L)", "conv 64", "conv 128", "conv 256"))


# ------------------------------------------------------------------------------------------------ head-blocked layouts
@pytest.mark.parametrize("F_,T,H", [(5, 197, 12), (3, 257, 16), (9, 128, 4)])
def test_head_blocked_qkv_attention_outproj_chain(hip, F_, T, H):
    """cfsar_gemm_lnfold_heads writes q | k | v per (frame, head) as one contiguous block, cfsar_vit_attention consumes / produces
    that blocking when called with D = 64, heads = 1, and cfsar_gemm_residual_stats_heads reads the blocked attention output as its
    A operand: every stage must equal its row-major counterpart bit for bit (same kernels, same arithmetic, other addresses)."""
    D = 64 * H
    M = F_ * T
    x = (_rand(M, D, seed=41) * 1.5 + 0.3).to(torch.float16).cuda()
    rstat = torch.empty(M, 4, device="cuda")
    hip.row_stats(x, rstat, M, D)
    Wg = _rand(3 * D, D, seed=42, scale=D ** -0.5).to(torch.float16).cuda()
    c, d = _rand(3 * D, seed=43).cuda(), _rand(3 * D, seed=44).cuda()
    qkv_rm = torch.empty(M, 3 * D, device="cuda", dtype=torch.bfloat16)
    hip.gemm_lnfold(x, Wg, qkv_rm, c, d, rstat, M=M)
    qkv_hb = torch.full((M * 3 * D,), 7.0, device="cuda", dtype=torch.bfloat16).reshape(M, 3 * D)
    hip.gemm_lnfold_heads(x, Wg, qkv_hb, c, d, rstat, T, H, M=M)
    # row-major [f, t, which, h, c] -> blocked [f, h, t, which, c]
    want = qkv_rm.reshape(F_, T, 3, H, 64).permute(0, 3, 1, 2, 4).contiguous().reshape(M, 3 * D)
    assert torch.equal(qkv_hb, want)
    o_rm = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    hip.vit_attention(qkv_rm, o_rm, F_, T, D, H)
    o_hb = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    hip.vit_attention(qkv_hb.reshape(F_ * H * T, 192), o_hb.reshape(F_ * H * T, 64), F_ * H, T, 64, 1)
    want_o = o_rm.reshape(F_, T, H, 64).permute(0, 2, 1, 3).contiguous().reshape(M, D)
    assert torch.equal(o_hb, want_o)
    Wo = _rand(D, D, seed=45, scale=D ** -0.5).to(torch.bfloat16).cuda()
    bo = _rand(D, seed=46).cuda()
    x0 = _rand(M, D, seed=47).to(torch.float16).cuda()
    xa, xb = x0.clone(), x0.clone()
    pa, pb = torch.empty(M, D // 64, 2, device="cuda"), torch.empty(M, D // 64, 2, device="cuda")
    hip.gemm_residual_stats(o_rm, Wo, xa, bo, pa, M=M)
    hip.gemm_residual_stats_heads(o_hb, Wo, xb, bo, T, pb, M=M)
    assert torch.equal(xa, xb) and torch.equal(pa, pb)
    # argument checks
    with pytest.raises(RuntimeError, match="tokens"):
        hip.gemm_lnfold_heads(x, Wg, qkv_hb, c, d, rstat, 64, H, M=M)


# ---------------------------------------------------------------------------------------------- round 4: fp16 numerics mode forms
def _hilo(W):
    hi = W.to(torch.float16)
    lo = (W - hi.float()).to(torch.float16)
    return torch.cat([hi, lo], 1).contiguous()


@pytest.mark.parametrize("M,N,K", [(777, 128, 128), (1300, 192, 192), (5000, 768, 768), (30000, 768, 3072), (15760, 768, 3072),
                                   (15760, 768, 768), (80, 768, 768), (1280, 1024, 4096)])
@pytest.mark.parametrize("two_word", [False, True])
@pytest.mark.parametrize("wsplit", [False, True])
def test_gemm_residual_wide(hip, M, N, K, two_word, wsplit):
    """cfsar_gemm_residual_wide == x + A W^T + b with the add in fp32 and ONE rounding (few_shot.py:633-635 / :639-640): the result is
    compared with a float64 reference of the operands the kernel sees; tolerance = fp32 accumulation round-off (+ half an fp16 ulp when
    the stream keeps one word).  Statistics: exact sums of the STORED hi words per 64-column slot."""
    g = torch.Generator().manual_seed(21)
    A = (torch.randn(M, K, generator=g) * 0.7).to(torch.float16).cuda()
    W32 = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    W = _hilo(W32) if wsplit else W32.to(torch.float16).contiguous()
    Weff = (W[:, :K].double() + W[:, K:].double()) if wsplit else W.double()
    bias = torch.randn(N, generator=g).cuda()
    x32 = (torch.randn(M, N, generator=g) * 3.0).cuda()
    xh = x32.to(torch.float16)
    xl = (x32 - xh.float()).to(torch.float16) if two_word else None
    ref = (xh.double() + (xl.double() if two_word else 0.0)) + A.double() @ Weff.t() + bias.double()
    S = N // 64
    part = torch.full((M, S, 2), float("nan"), device="cuda")
    xh2, xl2 = xh.clone(), (xl.clone() if two_word else None)
    hip.gemm_residual_wide(A, W, xh2, xl2, bias, part, wsplit=wsplit)
    got = xh2.double() + (xl2.double() if two_word else 0.0)
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    if two_word:
        assert err < 3e-6 * scale, (err, scale)             # ~2^-22 of the value + fp32 accumulation
        # lo is the rounding remainder of hi: at most half an fp16 ulp of it
        ulp_hi = torch.pow(2.0, torch.floor(torch.log2(xh2.float().abs().clamp_min(2.0 ** -14))) - 10)
        assert bool((xl2.float().abs() <= 0.5 * ulp_hi * 1.001).all())
    else:
        # one rounding: within half an fp16 ulp of the exact sum (+ fp32 accumulation round-off)
        ulp = torch.maximum(torch.abs(ref), torch.tensor(2.0 ** -14, dtype=torch.float64, device="cuda"))
        ulp = torch.pow(2.0, torch.floor(torch.log2(ulp)) - 10)
        assert bool(((got - ref).abs() <= 0.5 * ulp + 2e-6 * scale).all())
    hs = xh2.float().reshape(M, S, 64)
    assert maxdiff(part[:, :, 0], hs.sum(2)) < 1e-4 * max(1.0, float(hs.sum(2).abs().max()))
    assert maxdiff(part[:, :, 1], (hs * hs).sum(2)) < 1e-5 * float((hs * hs).sum(2).abs().max())
    # without statistics: same stream
    xh3, xl3 = xh.clone(), (xl.clone() if two_word else None)
    hip.gemm_residual_wide(A, W, xh3, xl3, bias, None, wsplit=wsplit)
    assert torch.equal(xh3, xh2) and (not two_word or torch.equal(xl3, xl2))


@pytest.mark.parametrize("M,N,K,act", [(777, 384, 128, "none"), (5000, 2304, 768, "none"), (30000, 3072, 768, "gelu"), (15760, 2304, 768, "none"),
                                       (80, 768, 768, "none"), (12850, 3072, 1024, "gelu")])
@pytest.mark.parametrize("from_part", [False, True])
def test_gemm_lnfold_split(hip, M, N, K, act, from_part):
    """cfsar_gemm_lnfold_hp (wsplit) == act(LayerNorm(x) W^T + b) with the weights carried as fp16 hi + lo: against the float64 computation on
    the operands the kernel sees (fp16 x, hi + lo weights) the only error left is the fp16 rounding of the OUTPUT (2^-12) + fp32
    accumulation; the same call with plain fp16 weights (cfsar_gemm_lnfold) must be measurably farther from the fp32-weight reference."""
    if from_part and K % 64 != 0:
        pytest.skip("partials need K = 64 slots")
    g = torch.Generator().manual_seed(31)
    x = (torch.randn(M, K, generator=g) * (0.5 + torch.rand(M, 1, generator=g) * 2.0) + torch.randn(M, 1, generator=g)).to(torch.float16).cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    gamma = (1.0 + 0.3 * torch.randn(K, generator=g)).cuda()
    beta = (0.3 * torch.randn(K, generator=g)).cuda()
    bias = torch.randn(N, generator=g).cuda()
    Wg32 = W * gamma[None, :]
    Wg2 = _hilo(Wg32)
    c = Wg2.double().sum(1).float()
    d = (W.double() @ beta.double() + bias.double()).float()
    S = K // 64
    xs = x.float().reshape(M, S, 64)
    part = torch.stack([xs.sum(2), (xs * xs).sum(2)], 2).contiguous()
    rstat = torch.empty(M, 4, device="cuda")
    hip.ln_stats_finalize(part, rstat, M, S, K)
    a = hip.ACT_QUICKGELU if act == "gelu" else hip.ACT_NONE
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16)
    if from_part:
        ws = torch.full((M, 4), float("nan"), device="cuda")
        hip.gemm_lnfold_hp(x, Wg2, out, c, d, partial=part, slots=S, rowstats_ws=ws, act=a, wsplit=True)
    else:
        hip.gemm_lnfold_hp(x, Wg2, out, c, d, rowstats=rstat, act=a, wsplit=True)
    xd = x.double()
    mean, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    ref = ((xd - mean) / torch.sqrt(var + 1e-5)) @ Wg32.double().t() + d.double()
    if act == "gelu":
        ref = ref * torch.sigmoid(1.702 * ref)
    scale = max(1.0, float(ref.abs().max()))
    e_split = float((out.double() - ref).abs().max())
    assert e_split < 6e-4 * scale, (e_split, scale)                # half an fp16 ulp of the largest output (+ accumulation)
    # the rounding of the weights is what the split removes: rms error against the fp32-weight reference, split vs plain
    out1 = torch.empty_like(out)
    Wg1 = Wg32.to(torch.float16).contiguous()
    hip.gemm_lnfold(x, Wg1, out1, Wg1.double().sum(1).float(), d, rstat, act=a)
    r_split = float((out.double() - ref).pow(2).mean().sqrt())
    r_plain = float((out1.double() - ref).pow(2).mean().sqrt())
    assert r_split <= r_plain * 1.02, (r_split, r_plain)


def test_copy_rows_strided_and_f16_pair(hip):
    g = torch.Generator().manual_seed(5)
    F_, N, D = 37, 197, 768
    x = torch.randn(F_ * N, D, generator=g).to(torch.float16).cuda()
    out = torch.zeros(F_, D, device="cuda", dtype=torch.float16)
    hip.copy_rows_strided(x, N * D * 2, out, D * 2, F_, D * 2)
    assert torch.equal(out, x.view(F_, N, D)[:, 0, :])
    st = torch.randn(F_ * N, 4, generator=g).cuda()
    o2 = torch.zeros(F_, 4, device="cuda")
    hip.copy_rows_strided(st, N * 16, o2, 16, F_, 16)
    assert torch.equal(o2, st.view(F_, N, 4)[:, 0, :])
    v = torch.randn(F_, D, generator=g).cuda() * 7
    hi = v.to(torch.float16)
    lo = (v - hi.float()).to(torch.float16)
    o3 = torch.empty(F_, D, device="cuda")
    hip.f16_pair_to_f32(hi, lo, o3)
    assert torch.equal(o3, hi.float() + lo.float())


@pytest.mark.parametrize("frames,tokens,K,norm", [(7, 197, 768, True), (5, 257, 1024, False), (3, 197, 3072, False), (2, 130, 128, True)])
def test_frame_col_means(hip, frames, tokens, K, norm):
    g = torch.Generator().manual_seed(41)
    M = frames * tokens
    A = (torch.randn(M, K, generator=g) * 2.0 + torch.randn(M, 1, generator=g)).to(torch.float16).cuda()
    rstat = None
    ref = A.float()
    if norm:
        mu, var = A.float().mean(1, keepdim=True), A.float().var(1, unbiased=False, keepdim=True)
        rstat = torch.cat([mu, torch.sqrt(var + 1e-5), torch.rsqrt(var + 1e-5), torch.zeros_like(mu)], 1).contiguous()
        ref = (A.float() - mu) * torch.rsqrt(var + 1e-5)
    ref = ref.view(frames, tokens, K).mean(1)
    out = torch.full((frames, K), float("nan"), device="cuda", dtype=torch.bfloat16)
    hip.frame_col_means(A, out, frames, tokens, rowstats=rstat)
    assert maxdiff(out.float(), ref) < 2.0 ** -8 * max(1.0, float(ref.abs().max())) + 1e-5          # bf16 output


@pytest.mark.parametrize("frames,tokens,N,K", [(6, 197, 768, 768), (40, 197, 768, 3072), (3, 257, 1024, 1024), (80, 197, 768, 768), (1, 197, 768, 768)])
@pytest.mark.parametrize("two_word", [False, True])
def test_gemm_residual_wide_per_frame_correction(hip, frames, tokens, N, K, two_word):
    """corr [frames, N] is added to every row of its frame (rows of a tile belong to at most two frames: both halves of the tail MFMA)."""
    g = torch.Generator().manual_seed(51)
    M = frames * tokens
    A = (torch.randn(M, K, generator=g) * 0.7).to(torch.float16).cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.float16).cuda()
    bias = torch.randn(N, generator=g).cuda()
    corr = (torch.randn(frames, N, generator=g) * 1e-2).cuda()
    x32 = (torch.randn(M, N, generator=g) * 3.0).cuda()
    xh = x32.to(torch.float16)
    xl = (x32 - xh.float()).to(torch.float16) if two_word else None
    cq = corr.to(torch.float16).double()                                                              # the kernel feeds it as an fp16 factor
    ref = (xh.double() + (xl.double() if two_word else 0.0)) + A.double() @ W.double().t() + bias.double() + cq.repeat_interleave(tokens, 0)
    xh2, xl2 = xh.clone(), (xl.clone() if two_word else None)
    hip.gemm_residual_wide(A, W, xh2, xl2, bias, None, corr=corr, corr_tokens=tokens)
    got = xh2.double() + (xl2.double() if two_word else 0.0)
    scale = float(ref.abs().max())
    tol = 3e-6 * scale if two_word else 2.0 ** -11 * scale
    assert float((got - ref).abs().max()) < tol, (float((got - ref).abs().max()), scale)
    # and it is the correction that is being added: without it the result differs by exactly corr (to rounding)
    xh3, xl3 = xh.clone(), (xl.clone() if two_word else None)
    hip.gemm_residual_wide(A, W, xh3, xl3, bias, None)
    d = (got - (xh3.double() + (xl3.double() if two_word else 0.0))).view(frames, tokens, N)
    assert float((d - cq[:, None, :]).abs().max()) < (1e-5 if two_word else 2.0 ** -10 * scale)


@pytest.mark.parametrize("frames,tokens,N,K,act", [(6, 197, 2304, 768, "none"), (40, 197, 3072, 768, "gelu"), (3, 257, 3072, 1024, "gelu"),
                                                   (80, 197, 2304, 768, "none"), (1, 197, 768, 768, "none")])
@pytest.mark.parametrize("raw", [False, True])
def test_gemm_lnfold_hp_per_frame_correction(hip, frames, tokens, N, K, act, raw):
    """out = act(LayerNorm(x) Wg^T + d + corr[frame]): the correction enters in normalised units, BEFORE the activation.  raw (corr_tokens < 0 at
    the C ABI): corr is in raw-stream units and shares the row's 1 / std with the GEMM: out = act(LayerNorm(x) Wg^T + d + corr[frame] / std_row)."""
    g = torch.Generator().manual_seed(61)
    M = frames * tokens
    x = (torch.randn(M, K, generator=g) * (0.5 + torch.rand(M, 1, generator=g) * 2.0) + torch.randn(M, 1, generator=g)).to(torch.float16).cuda()
    Wg = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.float16).cuda()
    c = Wg.double().sum(1).float()
    d = torch.randn(N, generator=g).cuda()
    corr = (torch.randn(frames, N, generator=g) * 1e-2).cuda()
    rstat = torch.empty(M, 4, device="cuda")
    hip.row_stats(x, rstat, M, K)
    a = hip.ACT_QUICKGELU if act == "gelu" else hip.ACT_NONE
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16)
    hip.gemm_lnfold_hp(x, Wg, out, c, d, rowstats=rstat, act=a, corr=corr, corr_tokens=tokens, corr_raw=raw)
    xd = x.double()
    mean, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    pre = ((xd - mean) / torch.sqrt(var + 1e-5)) @ Wg.double().t() + d.double()
    ref = pre + corr.double().repeat_interleave(tokens, 0) / (torch.sqrt(var + 1e-5) if raw else 1.0)
    ref0 = pre
    if act == "gelu":
        ref, ref0 = ref * torch.sigmoid(1.702 * ref), ref0 * torch.sigmoid(1.702 * ref0)
    scale = max(1.0, float(ref.abs().max()))
    e = float((out.double() - ref).abs().max())
    e0 = float((out.double() - ref0).abs().max())
    # tolerance: fp16 output rounding + the fp16 factors of the correction (std x corr: 2^-11 each of a 1e-2-size term)
    assert e < 6e-4 * scale + 1e-4, (e, scale)
    assert e0 > (3 if raw else 5) * e, (e0, e)            # the uncorrected reference is clearly farther away: the correction is really applied


@pytest.mark.parametrize("frames,tokens,N,K", [(6, 197, 3072, 768), (80, 197, 3072, 768), (3, 257, 4096, 1024), (1, 197, 768, 768), (33, 130, 512, 256)])
def test_gemm_lnfold_hp_emits_per_frame_output_means(hip, frames, tokens, N, K):
    """colmean_out[f] = token mean of the rows the GEMM just wrote for frame f (the c_fc GEMM hands c_proj the means of its operand): equal to
    the mean of the stored fp16 output up to the fp16 partial sums' rounding; the output itself is unchanged by asking for them."""
    g = torch.Generator().manual_seed(71)
    M = frames * tokens
    x = (torch.randn(M, K, generator=g) * (0.5 + torch.rand(M, 1, generator=g) * 2.0) + torch.randn(M, 1, generator=g)).to(torch.float16).cuda()
    Wg = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.float16).cuda()
    c, d = Wg.double().sum(1).float(), torch.randn(N, generator=g).cuda()
    rstat = torch.empty(M, 4, device="cuda")
    hip.row_stats(x, rstat, M, K)
    out0 = torch.empty(M, N, device="cuda", dtype=torch.float16)
    hip.gemm_lnfold(x, Wg, out0, c, d, rstat, act=hip.ACT_QUICKGELU)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16)
    um = torch.full((frames, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    ws = torch.full(((M // 96 + 2) * 2 * N,), -2 ** 30, device="cuda", dtype=torch.int32)
    hip.gemm_lnfold_hp(x, Wg, out, c, d, rowstats=rstat, act=hip.ACT_QUICKGELU, corr_tokens=tokens, colmean_out=um, colsum_ws=ws)
    assert torch.equal(out, out0)
    ref = out.float().view(frames, tokens, N).mean(1)
    assert not torch.isnan(um.float()).any()
    # bf16 output (2^-9) + the fixed-point resolution of the sums (2^-12 per value)
    assert maxdiff(um.float(), ref) < 4.5e-3 * max(1.0, float(ref.abs().max())), maxdiff(um.float(), ref)


def test_episode_top1(hip):
    g = torch.Generator().manual_seed(3)
    E, Q, way = 37, 5, 5
    logits = torch.randn(E, Q, way, generator=g).cuda()
    logits[3, 2, :] = 1.0                                   # a tie: the first maximum counts (torch.argmax order)
    labels = torch.randint(0, way, (E, Q), generator=g).float().cuda()
    acc = torch.full((E,), float("nan"), device="cuda")
    hip.episode_top1(logits, labels, acc)
    ref = (logits.argmax(dim=2) == labels.long()).float().mean(dim=1)
    assert torch.equal(acc, ref)


@pytest.mark.parametrize("F_,ntok,H", [(7, 197, 12), (3, 257, 16), (5, 130, 4), (4, 17, 2)])
def test_vit_attention_means(hip, F_, ntok, H):
    """cfsar_vit_attention_means: the same output bits as cfsar_vit_attention, plus the per-frame token means of that output (bf16)."""
    D = 64 * H
    g = torch.Generator().manual_seed(17)
    qkv = (torch.randn(F_ * ntok, 3 * D, generator=g)).to(torch.float16).cuda()
    o0 = torch.empty(F_ * ntok, D, device="cuda", dtype=torch.float16)
    hip.vit_attention(qkv, o0, F_, ntok, D, H)
    o1 = torch.full((F_ * ntok, D), float("nan"), device="cuda", dtype=torch.float16)
    om = torch.full((F_, D), float("nan"), device="cuda", dtype=torch.bfloat16)
    hip.vit_attention_means(qkv, o1, om, F_, ntok, D, H)
    assert torch.equal(o0, o1)
    ref = o1.float().view(F_, ntok, D).mean(1)
    assert not torch.isnan(om.float()).any()
    # bf16 output + the means are taken of the unrounded fp32 rows (fp16 rounding noise of the stored rows averages out)
    assert maxdiff(om.float(), ref) < 2.0 ** -8 * max(0.05, float(ref.abs().max())) + 2e-4, maxdiff(om.float(), ref)


@pytest.mark.parametrize("F_,ntok,H", [(7, 197, 12), (3, 257, 16), (5, 130, 4), (4, 17, 2)])
def test_vit_attention_pair(hip, F_, ntok, H):
    """cfsar_vit_attention_pair (round 6, fp16_strict): out_pair [F ntok, 2 D] = [o_hi | o_lo]; o_hi and the per-frame means are the bits of
    cfsar_vit_attention_means, o_lo is the rounding remainder of o_hi (at most half an fp16 ulp of it), and o_hi + o_lo is closer to the fp32
    attention of the same fp16 q / k / v than o_hi alone (what is left is the kernel's fp16 probabilities)."""
    D = 64 * H
    g = torch.Generator().manual_seed(17)
    qkv = (torch.randn(F_ * ntok, 3 * D, generator=g)).to(torch.float16).cuda()
    o1 = torch.empty(F_ * ntok, D, device="cuda", dtype=torch.float16)
    om1 = torch.empty(F_, D, device="cuda", dtype=torch.bfloat16)
    hip.vit_attention_means(qkv, o1, om1, F_, ntok, D, H)
    op = torch.full((F_ * ntok, 2 * D), float("nan"), device="cuda", dtype=torch.float16)
    om2 = torch.full((F_, D), float("nan"), device="cuda", dtype=torch.bfloat16)
    hip.vit_attention_pair(qkv, op, om2, F_, ntok, D, H)
    hi, lo = op[:, :D], op[:, D:]
    assert torch.equal(hi.contiguous(), o1) and torch.equal(om1, om2)
    assert not torch.isnan(lo.float()).any()
    ulp_hi = torch.pow(2.0, torch.floor(torch.log2(hi.float().abs().clamp_min(2.0 ** -14))) - 10)
    assert bool((lo.float().abs() <= 0.5 * ulp_hi * 1.001).all())
    ref = _ref_attention(qkv.cpu(), F_, ntok, D, H)
    e_hi = float((hi.float().cpu() - ref).pow(2).mean().sqrt())
    e_pair = float((hi.float().cpu() + lo.float().cpu() - ref).pow(2).mean().sqrt())
    assert e_pair < 0.9 * e_hi, (e_pair, e_hi)


@pytest.mark.parametrize("M,N,K", [(5000, 768, 768), (197 * 40, 1024, 1024), (300, 128, 128), (70000, 768, 768)])
@pytest.mark.parametrize("two_word", [True, False])
def test_gemm_residual_wide_two_word_operands(hip, M, N, K, two_word):
    """cfsar_gemm_residual_wide, wsplit = 2 (round 6, fp16_strict's out_proj): A [M, 2 K] = [a_hi | a_lo] against W [N, 3 K] = [w_hi | w_hi | w_lo] in
    one fp32 accumulation chain = x + (a_hi + a_lo)(w_hi + w_lo)^T + b up to the a_lo w_lo term (2^-24) and fp32 round-off; the same call on
    one-word operands (wsplit = 0 on a_hi, w_hi) is measurably farther from the reference on the unrounded operands."""
    g = torch.Generator().manual_seed(71)
    A32 = (torch.randn(M, K, generator=g) * 0.7).cuda()
    Ah = A32.to(torch.float16)
    Ap = torch.cat([Ah, (A32 - Ah.float()).to(torch.float16)], 1).contiguous()
    W32 = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    Wh = W32.to(torch.float16)
    Wl = (W32 - Wh.float()).to(torch.float16)
    W3 = torch.cat([Wh, Wh, Wl], 1).contiguous()
    bias = torch.randn(N, generator=g).cuda()
    x32 = (torch.randn(M, N, generator=g) * 3.0).cuda()
    xh = x32.to(torch.float16)
    xl = (x32 - xh.float()).to(torch.float16) if two_word else None
    x0 = xh.double() + (xl.double() if two_word else 0.0)
    ref = x0 + (Ap[:, :K].double() + Ap[:, K:].double()) @ (Wh.double() + Wl.double()).t() + bias.double()
    xh2, xl2 = xh.clone(), (xl.clone() if two_word else None)
    S = N // 64
    part = torch.full((M, S, 2), float("nan"), device="cuda")
    hip.gemm_residual_wide(Ap, W3, xh2, xl2, bias, part, wsplit=2)
    got = xh2.double() + (xl2.double() if two_word else 0.0)
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    if two_word:
        assert err < 3e-6 * scale, (err, scale)
        xh3, xl3 = xh.clone(), xl.clone()
        hip.gemm_residual_wide(Ah.contiguous(), Wh.contiguous(), xh3, xl3, bias, None)
        err1 = float((xh3.double() + xl3.double() - ref).abs().max())
        assert err1 > 20 * err, (err1, err)                        # the one-word operands' 11 bits
    else:
        ulp = torch.pow(2.0, torch.floor(torch.log2(torch.maximum(ref.abs(), torch.tensor(2.0 ** -14, dtype=torch.float64, device="cuda")))) - 10)
        assert bool(((got - ref).abs() <= 0.5 * ulp + 2e-6 * scale).all())
    hs = xh2.float().reshape(M, S, 64)
    assert maxdiff(part[:, :, 0], hs.sum(2)) < 1e-4 * max(1.0, float(hs.sum(2).abs().max()))
    with pytest.raises(RuntimeError, match="no per-frame correction"):
        hip.gemm_residual_wide(Ap, W3, xh2, xl2, bias, None, wsplit=2, corr=torch.zeros((M + 196) // 197, N, device="cuda"), corr_tokens=197)


@pytest.mark.parametrize("M,N,K", [(1, 768, 768), (80, 768, 768), (81, 2304, 768), (200, 768, 3072), (1280, 3072, 768), (1283, 1024, 4096)])
def test_frame_gemm_matches_fp32_and_is_row_invariant(hip, M, N, K):
    """cfsar_frame_gemm (the fp16 numerics mode's per-frame GEMMs, round 5): fp32 form == the fp32 product of the bf16 operands; bf16 form ==
    bf16(A W^T + bias + res) in place; and a row's bits do not depend on how many rows the call carries (batch-size invariance of the mode)."""
    g = torch.Generator(device="cuda").manual_seed(11)
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    ref = A.float() @ W.float().t()
    out = torch.full((M, N), float("nan"), device="cuda")
    hip.frame_gemm(A, W, out)
    assert float((out - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max())) + 1e-5
    xb = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
    want = (ref + bias + xb.float()).to(torch.bfloat16)
    got = xb.clone()
    hip.frame_gemm(A, W, got, bias=bias, res=got)
    assert float((got.float() - want.float()).abs().max()) <= 2.0 ** -7 * max(1.0, float(want.float().abs().max()))     # one bf16 ulp of the largest value
    assert (got != want).float().mean() < 0.02                                                 # (a different fp32 summation order flips a rounding now and then)
    # rows 0 .. m-1 alone give the same bits
    for m in {1, min(M, 17), min(M, 80)}:
        o2 = torch.full((m, N), float("nan"), device="cuda")
        hip.frame_gemm(A[:m].contiguous(), W, o2)
        assert torch.equal(o2, out[:m]), m
