"""ctypes binding of libclipfsar_hip.so (C ABI declared in include/clipfsar_hip.h).

There is NO fallback: if the shared library is missing, or a tensor is not a contiguous HIP device tensor of the
expected dtype, these wrappers raise.  PyTorch only provides device memory (``tensor.data_ptr()``) and the stream.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (imported first so that torch's libamdhip64.so.7 is the HIP runtime the library binds to)

F32, BF16, F16 = 0, 1, 2
ABI_VERSION = 9          # CFSAR_ABI_VERSION of include/clipfsar_hip.h this file's SIGNATURES were written against
ACT_NONE, ACT_QUICKGELU, ACT_GELU_ERF = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
# CFSAR_DEV_LIB=1 (developer tools only): the -DCFSAR_DEV build with the cfsar_debug_* hooks (clip-fsar_amd/build.py --dev)
DEV_LIB = os.environ.get("CFSAR_DEV_LIB", "0") == "1"
LIB_PATH = os.path.join(_HERE, "libclipfsar_hip_dev.so" if DEV_LIB else "libclipfsar_hip.so")
if os.environ.get("CFSAR_LIB_PATH"):       # developer A/B only (e.g. build.py --packed: profiles/r04_fault_audit.md); same ABI check as the product
    LIB_PATH = os.path.abspath(os.environ["CFSAR_LIB_PATH"])
_lib = None

_c_int, _c_p, _c_f, _c_i64 = ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_int64

# symbol -> argtypes; must match include/clipfsar_hip.h (tests/test_abi.py cross-checks against the header text)
SIGNATURES = {
    "cfsar_version": [],
    "cfsar_preprocess_frames": [_c_p, _c_p] + [_c_int] * 8 + [_c_p, _c_p, _c_p],
    "cfsar_im2col_patches": [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p],
    "cfsar_im2col_patches_split": [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p],
    "cfsar_embed_finish_pair": [_c_p] * 7 + [_c_int, _c_int, _c_int, _c_f, _c_p],
    "cfsar_patch_embed": [_c_p, _c_p, _c_int, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p],
    "cfsar_cls_rows": [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p],
    "cfsar_layernorm": [_c_p, _c_i64, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_int, _c_int, _c_f, _c_p],
    "cfsar_layernorm_ex": [_c_p, _c_int, _c_i64, _c_p, _c_i64, _c_int, _c_p, _c_p, _c_int, _c_int, _c_f, _c_p],
    "cfsar_cls_rows_ex": [_c_p, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p],
    "cfsar_gemm": [_c_p, _c_p, _c_p, _c_p, _c_p] + [_c_int] * 15 + [_c_p],
    "cfsar_gemm_ex": [_c_p, _c_p, _c_p, _c_p, _c_p] + [_c_int] * 17 + [_c_p],
    "cfsar_gemm_lnfold": [_c_p] * 6 + [_c_int] * 8 + [_c_p],
    "cfsar_gemm_residual_stats": [_c_p] * 5 + [_c_int] * 7 + [_c_p],
    "cfsar_gemm_lnfold_heads": [_c_p] * 6 + [_c_int] * 7 + [_c_p],
    "cfsar_gemm_lnfold_partials": [_c_p] * 6 + [_c_int, ctypes.c_float, _c_p] + [_c_int] * 10 + [_c_p],
    "cfsar_gemm_residual_stats_heads": [_c_p] * 5 + [_c_int] * 6 + [_c_p],
    "cfsar_gemm_lnfold_hp": [_c_p] * 7 + [_c_int, ctypes.c_float, _c_p] + [_c_int] * 9 + [_c_p, _c_int, _c_p, _c_p, _c_p],
    "cfsar_gemm_residual_wide": [_c_p] * 6 + [_c_int] * 7 + [_c_p, _c_int, _c_p],
    "cfsar_frame_col_means": [_c_p, _c_int, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p],
    "cfsar_f16_pair_to_f32": [_c_p, _c_p, _c_p, _c_i64, _c_p],
    "cfsar_copy_rows_strided": [_c_p, _c_i64, _c_p, _c_i64, _c_int, _c_int, _c_p],
    "cfsar_abi_version": [],
    "cfsar_ln_stats_finalize": [_c_p, _c_p, _c_int, _c_int, _c_int, _c_f, _c_p],
    "cfsar_row_stats": [_c_p, _c_p, _c_int, _c_int, _c_int, _c_f, _c_p],
    "cfsar_nchw_to_nhwc": [_c_p, _c_p] + [_c_int] * 5 + [_c_p],
    "cfsar_im2col3x3_nhwc": [_c_p, _c_p] + [_c_int] * 7 + [_c_p],
    "cfsar_avgpool2x2_nhwc": [_c_p, _c_p] + [_c_int] * 5 + [_c_p],
    "cfsar_attnpool_tokens": [_c_p, _c_p, _c_p] + [_c_int] * 4 + [_c_p],
    "cfsar_conv3x3_nhwc": [_c_p] * 5 + [_c_int] * 11 + [_c_p],
    "cfsar_attnpool_attend": [_c_p, _c_p, _c_p] + [_c_int] * 4 + [ctypes.c_float, _c_p],
    "cfsar_stem_conv3x3_s2": [_c_p] * 4 + [_c_int] * 6 + [_c_p],
    "cfsar_vit_attention": [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p],
    "cfsar_vit_attention_means": [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p],
    "cfsar_vit_attention_pair": [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p],
    "cfsar_vit_attention_cls": [_c_p, _c_i64, _c_p, _c_p, _c_int, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p],
    "cfsar_frame_gemm": [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p],
    "cfsar_class_text_logits": [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p],
    "cfsar_build_sequences": [_c_p, _c_p, _c_p, _c_p, _c_p] + [_c_int] * 8 + [_c_p],
    "cfsar_seq_attention": [_c_p, _c_p] + [_c_int] * 6 + [_c_f, _c_int, _c_p],
    "cfsar_embed_tokens": [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p],
    "cfsar_gather_rows": [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p],
    "cfsar_prototypes": [_c_p, _c_p, _c_p] + [_c_int] * 7 + [_c_p],
    "cfsar_text_match_probs": [_c_p] * 6 + [_c_int] * 7 + [_c_p],
    "cfsar_combine_logits": [_c_p, _c_p, _c_p, _c_int, _c_int, _c_f, _c_p],
    "cfsar_episode_top1": [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p],
    "cfsar_cos_otam_logits": [_c_p, _c_p, _c_p, _c_p] + [_c_int] * 5 + [_c_f, _c_int, _c_p],
}


def lib():
    """Load (once) and return the ctypes handle.  Raises loudly when the HIP extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "clip_fsar_amd: HIP extension %s is missing -- build it with `python clip-fsar_amd/build.py` "
                "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback for the hot path." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _c_int
        L.cfsar_last_error.restype = ctypes.c_char_p
        L.cfsar_last_error.argtypes = []
        if L.cfsar_abi_version() != ABI_VERSION:
            raise RuntimeError("clip_fsar_amd: %s has ABI revision %d, this binding was written against %d -- rebuild it "
                               "(python clip-fsar_amd/build.py --force)" % (LIB_PATH, L.cfsar_abi_version(), ABI_VERSION))
        if DEV_LIB:
            L.cfsar_debug_set_gemm_variant.argtypes = [ctypes.c_int, ctypes.c_int]
            L.cfsar_debug_set_gemm_variant.restype = None
            L.cfsar_debug_set_vit_paths.argtypes = [ctypes.c_int, ctypes.c_int]
            L.cfsar_debug_set_vit_paths.restype = None
            if os.environ.get("CFSAR_DEV_VIT_PATHS"):          # "opath,store" (10 + opath: short-K launches only; -1 = policy): bench A/B
                o, st = (int(v) for v in os.environ["CFSAR_DEV_VIT_PATHS"].split(","))
                L.cfsar_debug_set_vit_paths(o, st)
            if os.environ.get("CFSAR_DEV_VIT_DBG"):            # ablation bits of the ViT GEMM launches (bit 21 / 22: forced paths skip the LN-folded / the residual launches)
                L.cfsar_debug_set_vit_dbg.argtypes = [ctypes.c_int]
                L.cfsar_debug_set_vit_dbg.restype = None
                L.cfsar_debug_set_vit_dbg(int(os.environ["CFSAR_DEV_VIT_DBG"], 0))
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, lib().cfsar_last_error().decode(errors="replace")))


def _dev_ptr(t, dtype=None, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("clip_fsar_amd.hip: %s must be a HIP device tensor (no CPU path exists)" % name)
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("clip_fsar_amd.hip: %s has dtype %s, expected %s" % (name, t.dtype, dtype))
    if not t.is_contiguous():
        raise RuntimeError("clip_fsar_amd.hip: %s must be contiguous" % name)
    return ctypes.c_void_p(t.data_ptr())


def _opt(t, dtype, name):
    return None if t is None else _dev(t, dtype, name)


def _code(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float16:
        return F16
    raise RuntimeError("clip_fsar_amd.hip: unsupported dtype %s" % dtype)


import threading

_tls = threading.local()        # per host thread: the device of the operands of the wrapper call in progress


def _dev(t, dtype=None, name="tensor"):
    p = _dev_ptr(t, dtype, name)
    _tls.dev = t.device
    return p


def _stream():
    """Stream handle for the launch: the current stream OF THE DEVICE THE OPERANDS LIVE ON (the wrappers evaluate their
    tensor arguments first), and that device must be the thread's current device -- HIP launches go to the current device,
    so a mismatch would run the kernel on the wrong GPU."""
    d = getattr(_tls, "dev", None)      # thread-local: two host threads driving two GPUs cannot pick each other's device
    if d is not None and d.index is not None and d.index != torch.cuda.current_device():
        raise RuntimeError("clip_fsar_amd.hip: operands live on %s but the current device is cuda:%d -- wrap the call in "
                           "torch.cuda.device(%d) (one process per GPU sets it once)" % (d, torch.cuda.current_device(), d.index))
    return ctypes.c_void_p(torch.cuda.current_stream(d).cuda_stream)


# ----------------------------------------------------------------------------------------------- N2 frame transform
def preprocess_frames(frames_u8, out, scale_hw, crop, y0, x0, mean, std):
    """uint8 device frames [T,H,W,3] -> out [T,3,crop,crop] fp32 (resize, crop window, normalise)."""
    T, H, W, C = frames_u8.shape
    assert C == 3
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    sd = (ctypes.c_float * 3)(*[float(v) for v in std])
    _check(lib().cfsar_preprocess_frames(_dev(frames_u8, torch.uint8, "frames"), _dev(out, torch.float32, "out"), T, H, W,
                                         int(scale_hw[0]), int(scale_hw[1]), int(crop), int(y0), int(x0),
                                         ctypes.cast(m, ctypes.c_void_p), ctypes.cast(sd, ctypes.c_void_p), _stream()),
           "cfsar_preprocess_frames")


# ----------------------------------------------------------------------------------------------- ViT tower ops
def im2col_patches(frames, out, patch):
    """frames [F,3,H,W] f32 -> out [F*(H/P)*(W/P), k_pad] (f32 | bf16 | fp16), zero-padded columns."""
    F_, C, H, W = frames.shape
    assert C == 3
    _check(lib().cfsar_im2col_patches(_dev(frames, torch.float32, "frames"), _dev(out, None, "out"), _code(out.dtype),
                                      F_, H, W, patch, out.shape[1], _stream()), "cfsar_im2col_patches")


def im2col_patches_split(frames, out, patch):
    """frames [F,3,H,W] f32 -> out [F*(H/P)*(W/P), 3 k_pad] fp16 = [hi | lo | hi] of every pixel (cfsar_im2col_patches_split: the fp16_strict
    mode's three-pass patch-embed GEMM against [W_hi | W_hi | W_lo])."""
    F_, C, H, W = frames.shape
    assert C == 3 and out.shape[1] % 3 == 0
    _check(lib().cfsar_im2col_patches_split(_dev(frames, torch.float32, "frames"), _dev(out, torch.float16, "out"), F_, H, W, patch,
                                            out.shape[1] // 3, _stream()), "cfsar_im2col_patches_split")


def embed_finish_pair(tok, cls, pos, ln_w, ln_b, x_hi, x_lo, F_, ntok, D, eps=1e-5):
    """(class token | patch-embed rows) + pos -> ln_pre -> the two-word fp16 stream, no 16-bit rounding in between (cfsar_embed_finish_pair)."""
    if tok.shape[0] < F_ * (ntok - 1) or tok.shape[1] != D or x_hi.shape[-1] != D or x_lo.shape[-1] != D:
        raise RuntimeError("embed_finish_pair: tok must be [>= %d, %d], x_hi / x_lo [*, %d]" % (F_ * (ntok - 1), D, D))
    _check(lib().cfsar_embed_finish_pair(_dev(tok, torch.float32, "tok"), _dev(cls, torch.float32, "cls"), _dev(pos, torch.float32, "pos"),
                                         _dev(ln_w, torch.float32, "ln_w"), _dev(ln_b, torch.float32, "ln_b"), _dev(x_hi, torch.float16, "x_hi"),
                                         _dev(x_lo, torch.float16, "x_lo"), F_, ntok, D, eps, _stream()), "cfsar_embed_finish_pair")


PATCH_EMBED_SLOTS = {16: 768, 14: 704}       # K slots of cfsar_patch_embed's weight matrix (14: padded rows, see patch_embed_weight)


def patch_embed_ok(patch, w, x):
    """cfsar_patch_embed serves 16 x 16 and 14 x 14 patches with bf16 / fp16 weights [D, >= 768 | 704] into the fp16 residual stream."""
    return patch in PATCH_EMBED_SLOTS and w.dtype in (torch.bfloat16, torch.float16) and w.shape[1] >= PATCH_EMBED_SLOTS[patch] and x.dtype == torch.float16


def patch_embed_weight(conv_w, patch, dtype):
    """conv1.weight [D, 3, P, P] fp32 -> the weight matrix cfsar_patch_embed reads (layout only, then one rounding to `dtype`): P = 16: [D, 768]
    = reshape; P = 14: [D, 704] in PADDED ROWS, column (c*14 + dy)*16 + dx, zeros at dx = 14, 15 and behind column 672."""
    D = conv_w.shape[0]
    if patch == 16:
        return conv_w.reshape(D, 768).to(dtype).contiguous()
    w = torch.zeros(D, 3, 14, 16, device=conv_w.device, dtype=torch.float32)
    w[..., :14] = conv_w.reshape(D, 3, 14, 14)
    out = torch.zeros(D, 704, device=conv_w.device, dtype=torch.float32)
    out[:, :672] = w.reshape(D, 672)
    return out.to(dtype).contiguous()


def patch_embed(frames, w, pos, cls, x, patch=16):
    """x[f*ntok + 1 + p] = patch(f, p) @ w.T + pos[1 + p], x[f*ntok] = cls + pos[0] in one launch, gathered straight from the fp32 frames
    (include/clipfsar_hip.h: cfsar_patch_embed; few_shot.py:672-676)."""
    F_, C, H, W = frames.shape
    assert C == 3
    D = pos.shape[1]
    ntok = (H // patch) * (W // patch) + 1
    if pos.shape[0] != ntok or cls.numel() != D or w.shape[0] != D:
        raise RuntimeError("patch_embed: pos must be [%d, D], cls [D], w [D, >= %d]; got pos %s, cls %s, w %s" % (ntok, PATCH_EMBED_SLOTS.get(patch, 768), tuple(pos.shape), tuple(cls.shape), tuple(w.shape)))
    if x.shape[-1] != D or not x.is_contiguous() or x.numel() < F_ * ntok * D:
        raise RuntimeError("patch_embed: x must be contiguous with at least %d rows of %d, got %s" % (F_ * ntok, D, tuple(x.shape)))
    _check(lib().cfsar_patch_embed(_dev(frames, torch.float32, "frames"), _dev(w, None, "w"), _code(w.dtype), _dev(pos, torch.float32, "pos"),
                                   _dev(cls, torch.float32, "cls"), _dev(x, None, "x"), _code(x.dtype), F_, H, W, patch, D, w.stride(0),
                                   _stream()), "cfsar_patch_embed")


def cls_rows(x, cls, pos, F_, ntok, D):
    _check(lib().cfsar_cls_rows_ex(_dev(x, None, "x"), _code(x.dtype), _dev(cls, torch.float32, "cls"),
                                   _dev(pos, torch.float32, "pos"), F_, ntok, D, _stream()), "cfsar_cls_rows_ex")


def layernorm(x, out, weight, bias, rows, D, in_stride=None, out_stride=None, eps=1e-5):
    in_stride = D if in_stride is None else in_stride
    out_stride = D if out_stride is None else out_stride
    _check(lib().cfsar_layernorm_ex(_dev(x, None, "x"), _code(x.dtype), in_stride, _dev(out, None, "out"), out_stride,
                                    _code(out.dtype), _dev(weight, torch.float32, "weight"),
                                    _dev(bias, torch.float32, "bias"), rows, D, eps, _stream()), "cfsar_layernorm_ex")


def gemm(A, W, out, bias=None, residual=None, act=ACT_NONE, M=None, N=None, K=None, lda=None, ldw=None, ldo=None,
         ldr=None, row_group=0, row_gap=0, row_off=0, res_mod=0, res_off=0, relu=False):
    """out = [relu](act(A @ W.T + bias) + residual) (see include/clipfsar_hip.h: cfsar_gemm / cfsar_gemm_ex)."""
    if A.dtype != W.dtype:
        raise RuntimeError("gemm: A and W dtypes differ (%s vs %s)" % (A.dtype, W.dtype))
    M = A.shape[0] if M is None else M
    K = A.shape[1] if K is None else K
    N = W.shape[0] if N is None else N
    lda = A.shape[1] if lda is None else lda
    ldw = W.shape[1] if ldw is None else ldw
    ldo = out.shape[-1] if ldo is None else ldo
    ldr = (residual.shape[-1] if residual is not None else 0) if ldr is None else ldr
    if relu or (residual is not None and residual.dtype != torch.float32):
        _check(lib().cfsar_gemm_ex(_dev(A, None, "A"), _dev(W, None, "W"), _dev(out, None, "out"),
                                   _opt(bias, torch.float32, "bias"), _opt(residual, None, "residual"),
                                   M, N, K, lda, ldw, ldo, ldr, _code(A.dtype), _code(out.dtype), act,
                                   row_group, row_gap, row_off, res_mod, res_off,
                                   _code(residual.dtype) if residual is not None else F32, int(bool(relu)), _stream()),
               "cfsar_gemm_ex")
        return
    _check(lib().cfsar_gemm(_dev(A, None, "A"), _dev(W, None, "W"), _dev(out, None, "out"),
                            _opt(bias, torch.float32, "bias"), _opt(residual, torch.float32, "residual"),
                            M, N, K, lda, ldw, ldo, ldr, _code(A.dtype), _code(out.dtype), act,
                            row_group, row_gap, row_off, res_mod, res_off, _stream()), "cfsar_gemm")


_gemm_unwrapped = gemm


def _frame_gemm_ok(K, N):
    """Which kernel serves a per-frame GEMM of this (K, N): cfsar_frame_gemm for the long-K and the narrow ones (out_proj / c_proj corrections and
    stream-mean updates: 15 vs 29 us at 80 frames, 26 vs 41 us at 1 280), the generic cfsar_gemm for short K with N > 1 024 (the QKV / c_fc
    corrections: at 1 280 frames its 256 x 128 tiles take 15 us where the frame kernel's many short workgroups take 21-26; tools/frame_gemm_time.py).
    The choice depends on the architecture alone, never on the batch: an episode's bits do not depend on the batch it is served in."""
    return K % 128 == 0 and N % 16 == 0 and (K >= 2048 or N <= 1024)


def frame_gemm(A, W, out, bias=None, res=None):
    """out [M, N] (fp32, or bf16 with an optional bf16 residual that may alias out) = A [M, K] bf16 @ W [N, K]^T bf16 (+ bias) (+ res):
    cfsar_frame_gemm, whatever the shape policy of corr_gemm / mean_update_gemm says."""
    if out.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("frame_gemm: out must be fp32 or bf16, got %s" % out.dtype)
    _check(lib().cfsar_frame_gemm(_dev(A, torch.bfloat16, "A"), _dev(W, torch.bfloat16, "W"), _dev(out, None, "out"),
                                  _opt(bias, torch.float32, "bias"), _opt(res, torch.bfloat16, "res"), A.shape[0], W.shape[0], W.shape[1],
                                  _code(out.dtype), _stream()), "cfsar_frame_gemm")


def mean_update_gemm(meanA, w, bias, xbar):
    """xbar [frames, N] bf16 += meanA [frames, K] bf16 @ w [N, K]^T bf16 + bias: the per-frame mean of the residual stream follows the stream's
    update x += A W^T + b (linear in the frame's token mean).  cfsar_frame_gemm (bf16 form, in place)."""
    if not _frame_gemm_ok(w.shape[1], w.shape[0]):
        return _gemm_unwrapped(meanA, w, xbar, bias=bias, residual=xbar)
    frame_gemm(meanA, w, xbar, bias=bias, res=xbar)


def corr_gemm(meanA, w_lo, out):
    """out [frames, N] fp32 = meanA [frames, K] bf16 @ w_lo [N, K]^T bf16: the per-frame low-word correction's small GEMM (cfsar_frame_gemm, fp32
    form).  Not among the path's algorithmic GEMM launches that bench.py's per-launch timer counts."""
    if not _frame_gemm_ok(w_lo.shape[1], w_lo.shape[0]):
        return _gemm_unwrapped(meanA, w_lo, out)
    frame_gemm(meanA, w_lo, out)


def gemm_lnfold(x, Wg, out, cvec, dvec, rowstats, act=ACT_NONE, M=None):
    """out = act(LayerNorm(x) @ W.T + bias) with the LayerNorm folded into the GEMM (include/clipfsar_hip.h: cfsar_gemm_lnfold).
    out is bf16 (throughput mode) or fp16 (fp16 numerics mode)."""
    M = x.shape[0] if M is None else M
    if out.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError("gemm_lnfold: out must be bf16 or fp16, got %s" % out.dtype)
    _check(lib().cfsar_gemm_lnfold(_dev(x, torch.float16, "x"), _dev(Wg, torch.float16, "Wg"), _dev(out, None, "out"),
                                   _dev(cvec, torch.float32, "cvec"), _dev(dvec, torch.float32, "dvec"),
                                   _dev(rowstats, torch.float32, "rowstats"), M, Wg.shape[0], Wg.shape[1], x.shape[1],
                                   Wg.shape[1], out.shape[1], act, _code(out.dtype), _stream()), "cfsar_gemm_lnfold")


def lnfold_partials_ok(K, slots):
    """cfsar_gemm_lnfold_partials: K = 64 slots (fused in-kernel for the ViT-B / ViT-L widths at small M, two launches otherwise)."""
    return slots > 0 and K == 64 * slots


def gemm_lnfold_partials(x, Wg, out, cvec, dvec, partial, slots, rowstats_ws, act=ACT_NONE, M=None, tokens=0, heads=0, eps=1e-5):
    """gemm_lnfold / gemm_lnfold_heads (tokens > 0) with the statistics finalized inside the GEMM from the producer's partials
    [M, slots, 2] (include/clipfsar_hip.h: cfsar_gemm_lnfold_partials)."""
    M = x.shape[0] if M is None else M
    if out.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError("gemm_lnfold_partials: out must be bf16 or fp16, got %s" % out.dtype)
    _check(lib().cfsar_gemm_lnfold_partials(_dev(x, torch.float16, "x"), _dev(Wg, torch.float16, "Wg"), _dev(out, None, "out"),
                                            _dev(cvec, torch.float32, "cvec"), _dev(dvec, torch.float32, "dvec"),
                                            _dev(partial, torch.float32, "partial"), slots, eps,
                                            _dev(rowstats_ws, torch.float32, "rowstats_ws"), M, Wg.shape[0], Wg.shape[1], x.shape[1],
                                            Wg.shape[1], out.shape[-1], act, _code(out.dtype), tokens, heads, _stream()),
           "cfsar_gemm_lnfold_partials")


def gemm_residual_stats(A, W, x, bias, stats_partial=None, M=None):
    """x += A @ W.T + bias in place (fp16 stream) + optional partial LayerNorm statistics of the new x.  A and W: bf16, or fp16 in
    the fp16 numerics mode."""
    M = A.shape[0] if M is None else M
    if A.dtype != W.dtype or A.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError("gemm_residual_stats: A and W must both be bf16 or both fp16 (%s, %s)" % (A.dtype, W.dtype))
    _check(lib().cfsar_gemm_residual_stats(_dev(A, None, "A"), _dev(W, None, "W"), _dev(x, torch.float16, "x"),
                                           _dev(bias, torch.float32, "bias"), _opt(stats_partial, torch.float32, "stats_partial"),
                                           M, W.shape[0], W.shape[1], A.shape[1], W.shape[1], x.shape[1], _code(A.dtype), _stream()),
           "cfsar_gemm_residual_stats")


def gemm_lnfold_hp(x, Wg, out, cvec, dvec, rowstats=None, partial=None, slots=0, rowstats_ws=None, act=ACT_NONE, M=None, eps=1e-5,
                   wsplit=False, corr=None, corr_tokens=0, colmean_out=None, colsum_ws=None, corr_raw=False):
    """fp16 numerics mode's LN-folded GEMM (cfsar_gemm_lnfold_hp): split weights Wg [N, 2K] = [hi | lo] (wsplit) and / or the per-frame
    low-word correction corr [frames, N] fp32 (corr_raw: in raw-stream units, i.e. xbar W_lo^T with cvec = the EXACT column sums of W gamma; passed
    as a negative corr_tokens); colmean_out [frames, N] bf16 (+ colsum_ws): per-frame token means of the output."""
    M = x.shape[0] if M is None else M
    K = x.shape[1]
    if Wg.shape[1] != (2 * K if wsplit else K) or out.dtype != torch.float16:
        raise RuntimeError("gemm_lnfold_hp: Wg must be [N, %d] and out fp16" % (2 * K if wsplit else K))
    if corr is not None and corr.shape[1] != Wg.shape[0]:
        raise RuntimeError("gemm_lnfold_hp: corr must be [frames, N]")
    _check(lib().cfsar_gemm_lnfold_hp(_dev(x, torch.float16, "x"), _dev(Wg, torch.float16, "Wg"), _dev(out, None, "out"),
                                      _dev(cvec, torch.float32, "cvec"), _dev(dvec, torch.float32, "dvec"),
                                      _opt(rowstats, torch.float32, "rowstats"), _opt(partial, torch.float32, "partial"), slots, eps,
                                      _opt(rowstats_ws, torch.float32, "rowstats_ws"), M, Wg.shape[0], K, x.shape[1], Wg.shape[1],
                                      out.shape[-1], act, _code(out.dtype), int(bool(wsplit)), _opt(corr, torch.float32, "corr"),
                                      -int(corr_tokens) if corr_raw else int(corr_tokens), _opt(colmean_out, torch.bfloat16, "colmean_out"),
                                      _opt(colsum_ws, torch.int32, "colsum_ws"), _stream()), "cfsar_gemm_lnfold_hp")


def gemm_residual_wide(A, W, x, x_lo, bias, stats_partial=None, M=None, wsplit=False, corr=None, corr_tokens=0):
    """x (+ x_lo) += A @ W.T + bias with the add in fp32 and ONE rounding; W [N, K] or split [N, 2K] (wsplit True / 1); wsplit = 2: A [M, 2K] =
    [a_hi | a_lo] against W [N, 3K] = [w_hi | w_hi | w_lo]; corr: per-frame low-word correction [frames, N] fp32 (cfsar_gemm_residual_wide)."""
    M = A.shape[0] if M is None else M
    ws = int(wsplit)
    K = A.shape[1] // 2 if ws == 2 else A.shape[1]
    if W.shape[1] != (3 * K if ws == 2 else (2 * K if ws else K)) or (ws == 2 and A.shape[1] != 2 * K):
        raise RuntimeError("gemm_residual_wide: W has %d columns, expected %d" % (W.shape[1], 3 * K if ws == 2 else (2 * K if ws else K)))
    if corr is not None and corr.shape[1] != W.shape[0]:
        raise RuntimeError("gemm_residual_wide: corr must be [frames, N]")
    _check(lib().cfsar_gemm_residual_wide(_dev(A, torch.float16, "A"), _dev(W, torch.float16, "W"), _dev(x, torch.float16, "x"),
                                          _opt(x_lo, torch.float16, "x_lo"), _dev(bias, torch.float32, "bias"),
                                          _opt(stats_partial, torch.float32, "stats_partial"), M, W.shape[0], K, ws,
                                          A.shape[1], W.shape[1], x.shape[1], _opt(corr, torch.float32, "corr"), int(corr_tokens),
                                          _stream()), "cfsar_gemm_residual_wide")


def frame_col_means(A, out, frames, tokens, rowstats=None):
    """out [frames, K] bf16 = per-frame token mean of A's rows (normalised by the rows' (mean, 1/std) when rowstats is given)."""
    _check(lib().cfsar_frame_col_means(_dev(A, torch.float16, "A"), A.shape[1], _opt(rowstats, torch.float32, "rowstats"),
                                       _dev(out, torch.bfloat16, "out"), frames, tokens, out.shape[1], _stream()), "cfsar_frame_col_means")


def f16_pair_to_f32(hi, lo, out):
    _check(lib().cfsar_f16_pair_to_f32(_dev(hi, torch.float16, "hi"), _dev(lo, torch.float16, "lo"), _dev(out, torch.float32, "out"),
                                       hi.numel(), _stream()), "cfsar_f16_pair_to_f32")


def copy_rows_strided(src, src_stride_bytes, dst, dst_stride_bytes, rows, row_bytes):
    """dst row r <- src row r (byte strides); src / dst: device tensors whose data_ptr is row 0 (cfsar_copy_rows_strided)."""
    _check(lib().cfsar_copy_rows_strided(_dev(src, None, "src"), src_stride_bytes, _dev(dst, None, "dst"), dst_stride_bytes, rows,
                                         row_bytes, _stream()), "cfsar_copy_rows_strided")


def gemm_lnfold_heads(x, Wg, out, cvec, dvec, rowstats, tokens, heads, M=None):
    """The QKV form of gemm_lnfold with head-blocked output [(f heads + h) tokens + t][q | k | v] (cfsar_gemm_lnfold_heads)."""
    M = x.shape[0] if M is None else M
    _check(lib().cfsar_gemm_lnfold_heads(_dev(x, torch.float16, "x"), _dev(Wg, torch.float16, "Wg"), _dev(out, torch.bfloat16, "out"),
                                         _dev(cvec, torch.float32, "cvec"), _dev(dvec, torch.float32, "dvec"),
                                         _dev(rowstats, torch.float32, "rowstats"), M, Wg.shape[0], Wg.shape[1], x.shape[1],
                                         Wg.shape[1], tokens, heads, _stream()), "cfsar_gemm_lnfold_heads")


def gemm_residual_stats_heads(A, W, x, bias, tokens, stats_partial=None, M=None):
    """gemm_residual_stats whose A operand is the head-blocked attention output (cfsar_gemm_residual_stats_heads)."""
    M = x.shape[0] if M is None else M
    _check(lib().cfsar_gemm_residual_stats_heads(_dev(A, torch.bfloat16, "A"), _dev(W, torch.bfloat16, "W"), _dev(x, torch.float16, "x"),
                                                 _dev(bias, torch.float32, "bias"), _opt(stats_partial, torch.float32, "stats_partial"),
                                                 M, W.shape[0], W.shape[1], W.shape[1], x.shape[1], tokens, _stream()),
           "cfsar_gemm_residual_stats_heads")


def ln_stats_finalize(partial, rowstats, M, slots, D, eps=1e-5):
    _check(lib().cfsar_ln_stats_finalize(_dev(partial, torch.float32, "partial"), _dev(rowstats, torch.float32, "rowstats"), M,
                                         slots, D, eps, _stream()), "cfsar_ln_stats_finalize")


def row_stats(x, rowstats, M, D, eps=1e-5):
    _check(lib().cfsar_row_stats(_dev(x, torch.float16, "x"), _dev(rowstats, torch.float32, "rowstats"), M, D, x.shape[1], eps,
                                 _stream()), "cfsar_row_stats")


def vit_attention(qkv, out, F_, ntok, D, heads):
    if qkv.dtype != out.dtype:
        raise RuntimeError("vit_attention: qkv/out dtype mismatch")
    _check(lib().cfsar_vit_attention(_dev(qkv, None, "qkv"), _dev(out, None, "out"), _code(qkv.dtype), F_, ntok, D,
                                     heads, _stream()), "cfsar_vit_attention")


def vit_attention_means(qkv, out, omean, F_, ntok, D, heads):
    """vit_attention (fp16) + omean [F, D] bf16: the per-frame token means of the attention output (cfsar_vit_attention_means)."""
    _check(lib().cfsar_vit_attention_means(_dev(qkv, torch.float16, "qkv"), _dev(out, torch.float16, "out"), _dev(omean, torch.bfloat16, "omean"),
                                           F_, ntok, D, heads, _stream()), "cfsar_vit_attention_means")


def vit_attention_pair(qkv, out_pair, omean, F_, ntok, D, heads):
    """vit_attention_means with the output in two fp16 words: out_pair [F ntok, 2 D] = [o_hi | o_lo] (cfsar_vit_attention_pair, fp16_strict)."""
    if out_pair.shape[-1] != 2 * D:
        raise RuntimeError("vit_attention_pair: out_pair must be [F ntok, 2 D]")
    _check(lib().cfsar_vit_attention_pair(_dev(qkv, torch.float16, "qkv"), _dev(out_pair, torch.float16, "out_pair"), _dev(omean, torch.bfloat16, "omean"),
                                          F_, ntok, D, heads, _stream()), "cfsar_vit_attention_pair")


def vit_attention_cls(qkv, out, F_, ntok, D, heads, q=None, kv=None):
    """Attention output of ONE query row per frame (the class token in the last ViT block): out [F, D] (include/clipfsar_hip.h:
    cfsar_vit_attention_cls).  Either the packed qkv matrix [F*ntok, 3D] (query = row 0 of every frame), or q [F, D] + kv [F*ntok, 2D]
    (k | v)."""
    if q is None:
        if qkv.dtype != out.dtype:
            raise RuntimeError("vit_attention_cls: qkv/out dtype mismatch")
        base, es = qkv.data_ptr(), qkv.element_size()
        _dev(qkv, None, "qkv")
        qp, ldq, kp, vp, ldkv, dt = base, ntok * 3 * D, base + D * es, base + 2 * D * es, 3 * D, qkv.dtype
    else:
        if not (q.dtype == kv.dtype == out.dtype):
            raise RuntimeError("vit_attention_cls: q / kv / out dtype mismatch")
        _dev(q, None, "q")
        _dev(kv, None, "kv")
        es = kv.element_size()
        qp, ldq, kp, vp, ldkv, dt = q.data_ptr(), q.shape[1], kv.data_ptr(), kv.data_ptr() + D * es, 2 * D, kv.dtype
    _check(lib().cfsar_vit_attention_cls(ctypes.c_void_p(qp), ldq, ctypes.c_void_p(kp), ctypes.c_void_p(vp), ldkv, _dev(out, None, "out"),
                                         _code(dt), F_, ntok, D, heads, _stream()), "cfsar_vit_attention_cls")


# ----------------------------------------------------------------------------------------------- few-shot tail ops
def class_text_logits(feats, text, scale, out, n_videos, T, E):
    _check(lib().cfsar_class_text_logits(_dev(feats, torch.float32, "feats"), _dev(text, torch.float32, "text"),
                                         _dev(scale, torch.float32, "scale"), _dev(out, torch.float32, "out"),
                                         n_videos, T, E, text.shape[0], _stream()), "cfsar_class_text_logits")


def build_sequences(feats, text_test, support_labels, real_support_labels, X, B, S, Q, T, E, way, merge_before):
    _check(lib().cfsar_build_sequences(_dev(feats, torch.float32, "feats"), _dev(text_test, torch.float32, "text_test"),
                                       _dev(support_labels, torch.float32, "support_labels"),
                                       _dev(real_support_labels, torch.float32, "real_support_labels"),
                                       _dev(X, torch.float32, "X"), B, S, Q, T, E, way, text_test.shape[0],
                                       int(bool(merge_before)), _stream()), "cfsar_build_sequences")


def seq_attention(qkv, out, n_a, len_a, n_b, len_b, heads, head_dim, scale, causal=False):
    _check(lib().cfsar_seq_attention(_dev(qkv, torch.float32, "qkv"), _dev(out, torch.float32, "out"), n_a, len_a, n_b,
                                     len_b, heads, head_dim, float(scale), int(bool(causal)), _stream()),
           "cfsar_seq_attention")


def embed_tokens(tokens, table, pos, out):
    n_seq, L = tokens.shape
    _check(lib().cfsar_embed_tokens(_dev(tokens, torch.int32, "tokens"), _dev(table, torch.float32, "table"),
                                    _dev(pos, torch.float32, "pos"), _dev(out, torch.float32, "out"), n_seq, L,
                                    table.shape[1], table.shape[0], _stream()), "cfsar_embed_tokens")


def gather_rows(x, idx, out):
    _check(lib().cfsar_gather_rows(_dev(x, torch.float32, "x"), _dev(idx, torch.int32, "idx"),
                                   _dev(out, torch.float32, "out"), idx.shape[0], x.shape[1], x.shape[0], _stream()),
           "cfsar_gather_rows")


def prototypes(Xs, support_labels, protos, B, S, Sp, T, E, way, merge_before):
    _check(lib().cfsar_prototypes(_dev(Xs, torch.float32, "Xs"), _dev(support_labels, torch.float32, "support_labels"),
                                  _dev(protos, torch.float32, "protos"), B, S, Sp, T, E, way, int(bool(merge_before)),
                                  _stream()), "cfsar_prototypes")


def cos_otam_logits(Xq, protos, logits, B, Q, way, T, E, lbda=0.5, single_direct=False, dists_out=None):
    _check(lib().cfsar_cos_otam_logits(_dev(Xq, torch.float32, "Xq"), _dev(protos, torch.float32, "protos"),
                                       _dev(logits, torch.float32, "logits"), _opt(dists_out, torch.float32, "dists_out"),
                                       B, Q, way, T, E, float(lbda), int(bool(single_direct)), _stream()),
           "cfsar_cos_otam_logits")


def episode_top1(logits, target_labels, acc):
    """acc[e] = top-1 accuracy of episode e (logits [E, Q, way], target_labels [E, Q] float class indices)."""
    E, Q, way = logits.shape
    _check(lib().cfsar_episode_top1(_dev(logits, torch.float32, "logits"), _dev(target_labels, torch.float32, "target_labels"),
                                    _dev(acc, torch.float32, "acc"), E, Q, way, _stream()), "cfsar_episode_top1")


def text_match_probs(feats, text_test, support_labels, real_support_labels, scale, probs, B, S, Q, T, E, way):
    _check(lib().cfsar_text_match_probs(_dev(feats, torch.float32, "feats"), _dev(text_test, torch.float32, "text_test"),
                                        _dev(support_labels, torch.float32, "support_labels"),
                                        _dev(real_support_labels, torch.float32, "real_support_labels"),
                                        _dev(scale, torch.float32, "scale"), _dev(probs, torch.float32, "probs"), B, S, Q, T,
                                        E, way, text_test.shape[0], _stream()), "cfsar_text_match_probs")


def combine_logits(text_probs, visual_logits, out, n_queries, way, text_coff):
    _check(lib().cfsar_combine_logits(_dev(text_probs, torch.float32, "text_probs"),
                                      _dev(visual_logits, torch.float32, "visual_logits"), _dev(out, torch.float32, "out"),
                                      n_queries, way, float(text_coff), _stream()), "cfsar_combine_logits")


# ----------------------------------------------------------------------------------------------- N3 RN50 tower helpers
def nchw_to_nhwc(frames, out):
    F_, C, H, W = frames.shape
    _check(lib().cfsar_nchw_to_nhwc(_dev(frames, torch.float32, "frames"), _dev(out, None, "out"), _code(out.dtype), F_, C, H,
                                    W, _stream()), "cfsar_nchw_to_nhwc")


def im2col3x3(x, out, F_, H, W, C, stride):
    if x.dtype != out.dtype:
        raise RuntimeError("im2col3x3: dtype mismatch")
    _check(lib().cfsar_im2col3x3_nhwc(_dev(x, None, "x"), _dev(out, None, "out"), _code(x.dtype), F_, H, W, C, stride,
                                      out.shape[1], _stream()), "cfsar_im2col3x3_nhwc")


def avgpool2x2(x, out, F_, H, W, C):
    _check(lib().cfsar_avgpool2x2_nhwc(_dev(x, None, "x"), _dev(out, x.dtype, "out"), _code(x.dtype), F_, H, W, C, _stream()),
           "cfsar_avgpool2x2_nhwc")


def stem_conv(frames, w, bias, out, relu=True):
    """RN50 stem conv1 (3 -> Cout, 3x3, stride 2, pad 1) + folded BN + ReLU from fp32 NCHW frames to NHWC `out`."""
    F_, C, H, W = frames.shape
    if C != 3:
        raise RuntimeError("stem_conv: 3 input channels expected")
    _check(lib().cfsar_stem_conv3x3_s2(_dev(frames, torch.float32, "frames"), _dev(w, torch.float32, "w"),
                                       _opt(bias, torch.float32, "bias"), _dev(out, None, "out"), _code(out.dtype), F_, H, W,
                                       w.shape[0], int(bool(relu)), _stream()), "cfsar_stem_conv3x3_s2")


def attnpool_attend(q, kv, out, F_, T, heads, head_dim, scale):
    _check(lib().cfsar_attnpool_attend(_dev(q, torch.float32, "q"), _dev(kv, torch.float32, "kv"), _dev(out, torch.float32, "out"),
                                       F_, T, heads, head_dim, float(scale), _stream()), "cfsar_attnpool_attend")


def conv3x3(x, w, out, F_, H, W, C, bias=None, residual=None, relu=False):
    """Implicit-GEMM / direct 3x3, pad 1, stride 1 conv on bf16 (or fp16: then `out` is fp16 too) NHWC activations
    (include/clipfsar_hip.h: cfsar_conv3x3_nhwc)."""
    if x.dtype not in (torch.bfloat16, torch.float16) or w.dtype != x.dtype:
        raise RuntimeError("conv3x3: bf16 or fp16 activations, weights of the same type")
    if (x.dtype == torch.float16) != (out.dtype == torch.float16):
        raise RuntimeError("conv3x3: fp16 activations write fp16 outputs (and only they do)")
    _check(lib().cfsar_conv3x3_nhwc(_dev(x, None, "x"), _dev(w, None, "w"), _dev(out, None, "out"),
                                    _opt(bias, torch.float32, "bias"), _opt(residual, None, "residual"), F_, H, W, C,
                                    w.shape[0], w.shape[1], out.shape[-1], residual.shape[-1] if residual is not None else 0,
                                    _code(out.dtype), _code(residual.dtype) if residual is not None else F32,
                                    int(bool(relu)), _stream()), "cfsar_conv3x3_nhwc")


def attnpool_tokens(x, pos, out, F_, HW, C):
    _check(lib().cfsar_attnpool_tokens(_dev(x, None, "x"), _dev(pos, torch.float32, "pos"), _dev(out, x.dtype, "out"),
                                       _code(x.dtype), F_, HW, C, _stream()), "cfsar_attnpool_tokens")
