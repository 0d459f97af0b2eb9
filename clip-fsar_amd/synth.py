"""Deterministic synthetic weights, text features, frames and episodes.

Everything here is derived from a counter-based integer hash (splitmix64) with
exact IEEE-754 double arithmetic only (integer -> float conversion, add, mul,
div by constants), so the dev container (where the golden vectors are generated
by importing the reference) and the GPU box (where the HIP path is checked
against them) produce bit-identical fp32 inputs without shipping 350 MB of
weights or 48 MB of frames.  No torch RNG, no libm calls.

Layouts follow the reference:
  * state-dict key names / shapes of ``CNN_OTAM_CLIPFSAR`` with a CLIP ViT
    visual tower (reference models/base/few_shot.py:654-669, 619-631, 979-989,
    1035-1055, 1643-1652, 2731-2739; SURVEY.md section 5 lists the 164 keys);
  * the 7-key episode dict produced by ``Ssv2_few_shot.__getitem__``
    (reference datasets/base/ssv2_few_shot.py:275-285).
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import numpy as np

# --------------------------------------------------------------------------------------
# architectures (reference few_shot.py:232-242 `_MODELS`, :849-878 `build_model` conventions)
# --------------------------------------------------------------------------------------
ARCHS = {
    # name: ViT width, layers, heads, patch, input resolution, CLIP embed dim
    "ViT-B/16": dict(width=768, layers=12, heads=12, patch=16, res=224, embed=512),
    # Extension A16: the reference head has no ViT-L/14 branch.
    "ViT-L/14": dict(width=1024, layers=24, heads=16, patch=14, res=224, embed=768),
    # small test architectures (same code path, head_dim stays 64 like CLIP)
    "ViT-test/16": dict(width=128, layers=2, heads=2, patch=16, res=64, embed=64),
    "ViT-test197/16": dict(width=128, layers=2, heads=2, patch=16, res=224, embed=64),
    "ViT-test257/14": dict(width=128, layers=2, heads=2, patch=14, res=224, embed=64),
    # N3: CLIP ModifiedResNet towers (few_shot.py:542-602; RN50 hyper-parameters per `build_model` :859-866, :720-728)
    "RN50": dict(kind="rn", layers=(3, 4, 6, 3), width=64, res=224, embed=1024, heads=32),
    "RN-test": dict(kind="rn", layers=(1, 1, 1, 1), width=64, res=64, embed=64, heads=32),
}

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # reference yaml DATA.MEAN (K100 1-shot :49)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)   # reference yaml DATA.STD  (:50)

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _name_seed(name: str, seed: int) -> np.uint64:
    h = hashlib.sha256(("%d/%s" % (seed, name)).encode()).digest()
    return np.uint64(int.from_bytes(h[:8], "little"))


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finalizer on a uint64 array (wraps mod 2**64)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        x = x ^ (x >> np.uint64(31))
    return x


_IH_STD = float(np.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0))


def pseudo_normal(n: int, name: str, seed: int = 0, offset: int = 0) -> np.ndarray:
    """n values with zero mean / unit variance (Irwin-Hall sum of four 16-bit uniforms).

    Value i depends only on (name, seed, offset + i): any slice can be regenerated alone.
    """
    base = _name_seed(name, seed)
    out = np.empty(n, dtype=np.float64)
    chunk = 1 << 22
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        idx = np.arange(offset + s, offset + e, dtype=np.uint64)
        with np.errstate(over="ignore"):
            h = _splitmix64((idx * np.uint64(0xD1342543DE82EF95) + base) & _M64)
        acc = (h & np.uint64(0xFFFF)).astype(np.int64)
        acc += ((h >> np.uint64(16)) & np.uint64(0xFFFF)).astype(np.int64)
        acc += ((h >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
        acc += (h >> np.uint64(48)).astype(np.int64)
        out[s:e] = (acc - 2 * 65535).astype(np.float64) / _IH_STD
    return out


def tensor(shape, name: str, std: float = 1.0, mean: float = 0.0, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    v = pseudo_normal(n, name, seed) * std + mean
    return v.astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
def vit_state_dict(arch: str, seed: int = 18, prefix: str = "") -> "OrderedDict[str, np.ndarray]":
    """Random-init CLIP ViT visual tower with the reference's parameter names
    (few_shot.py:654-669 VisionTransformer, :619-631 ResidualAttentionBlock).

    Scales follow CLIP.initialize_parameters conventions (few_shot.py:765-773, the
    text tower's std's) WITHOUT the 1/sqrt(2L) residual damping and with non-trivial
    LayerNorm affine / biases, so that every parameter influences the output and the
    features stay input-dependent (PyTorch-default init gives
    degenerate, nearly input-independent features)."""
    a = ARCHS[arch]
    D, L, P, E = a["width"], a["layers"], a["patch"], a["embed"]
    ntok = (a["res"] // P) ** 2 + 1
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def put(key, shape, std, mean=0.0):
        sd[prefix + key] = tensor(shape, arch + "/" + key, std=std, mean=mean, seed=seed)

    put("class_embedding", (D,), D ** -0.5)
    put("positional_embedding", (ntok, D), D ** -0.5)
    put("proj", (D, E), D ** -0.5)
    put("conv1.weight", (D, 3, P, P), (3 * P * P) ** -0.5)
    put("ln_pre.weight", (D,), 0.1, 1.0)
    put("ln_pre.bias", (D,), 0.1)
    for i in range(L):
        b = "transformer.resblocks.%d." % i
        put(b + "ln_1.weight", (D,), 0.1, 1.0)
        put(b + "ln_1.bias", (D,), 0.1)
        put(b + "attn.in_proj_weight", (3 * D, D), 1.5 * D ** -0.5)
        put(b + "attn.in_proj_bias", (3 * D,), 0.1)
        put(b + "attn.out_proj.weight", (D, D), D ** -0.5)
        put(b + "attn.out_proj.bias", (D,), 0.1)
        put(b + "ln_2.weight", (D,), 0.1, 1.0)
        put(b + "ln_2.bias", (D,), 0.1)
        put(b + "mlp.c_fc.weight", (4 * D, D), D ** -0.5)
        put(b + "mlp.c_fc.bias", (4 * D,), 0.1)
        put(b + "mlp.c_proj.weight", (D, 4 * D), (4 * D) ** -0.5)
        put(b + "mlp.c_proj.bias", (D,), 0.1)
    put("ln_post.weight", (D,), 0.1, 1.0)
    put("ln_post.bias", (D,), 0.1)
    return sd


def rn_state_dict(arch: str, seed: int = 18, prefix: str = "") -> "OrderedDict[str, np.ndarray]":
    """Random-init CLIP ModifiedResNet with the reference's parameter / buffer names (few_shot.py:182-227 Bottleneck,
    :435-444 AttentionPool2d, :542-579 ModifiedResNet): He-scaled convs, non-trivial BatchNorm affine and running stats
    (eval mode uses the running stats)."""
    a = ARCHS[arch]
    width, layers, E = a["width"], a["layers"], a["embed"]
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def put(key, shape, std, mean=0.0):
        sd[prefix + key] = tensor(shape, arch + "/" + key, std=std, mean=mean, seed=seed)

    def conv(key, cout, cin, k):
        put(key + ".weight", (cout, cin, k, k), (2.0 / (cin * k * k)) ** 0.5)

    def bn(key, c, gamma=1.0):
        put(key + ".weight", (c,), 0.1, gamma)
        put(key + ".bias", (c,), 0.1)
        put(key + ".running_mean", (c,), 0.1)
        sd[prefix + key + ".running_var"] = (np.abs(tensor((c,), arch + "/" + key + ".rv", std=0.2, seed=seed)) + 0.6).astype(np.float32)
        sd[prefix + key + ".num_batches_tracked"] = np.asarray(0, np.int64)

    conv("conv1", width // 2, 3, 3); bn("bn1", width // 2)
    conv("conv2", width // 2, width // 2, 3); bn("bn2", width // 2)
    conv("conv3", width, width // 2, 3); bn("bn3", width)
    inplanes = width
    for li, (planes, blocks) in enumerate(zip((width, width * 2, width * 4, width * 8), layers), start=1):
        for bi in range(blocks):
            stride = 2 if (li > 1 and bi == 0) else 1
            b = "layer%d.%d." % (li, bi)
            conv(b + "conv1", planes, inplanes, 1); bn(b + "bn1", planes)
            conv(b + "conv2", planes, planes, 3); bn(b + "bn2", planes)
            conv(b + "conv3", planes * 4, planes, 1); bn(b + "bn3", planes * 4, gamma=0.25)
            if stride > 1 or inplanes != planes * 4:
                conv(b + "downsample.0", planes * 4, inplanes, 1); bn(b + "downsample.1", planes * 4, gamma=0.6)
            inplanes = planes * 4
    C = width * 32
    sp = (a["res"] // 32) ** 2 + 1
    put("attnpool.positional_embedding", (sp, C), C ** -0.5)
    for n in ("k_proj", "q_proj", "v_proj"):
        put("attnpool.%s.weight" % n, (C, C), C ** -0.5)
        put("attnpool.%s.bias" % n, (C,), 0.05)
    put("attnpool.c_proj.weight", (E, C), C ** -0.5)
    put("attnpool.c_proj.bias", (E,), 0.05)
    return sd


def visual_state_dict(arch: str, seed: int = 18, prefix: str = ""):
    return rn_state_dict(arch, seed, prefix) if ARCHS[arch].get("kind") == "rn" else vit_state_dict(arch, seed, prefix)


def context2_state_dict(dim: int, heads: int = 8, dim_head: int | None = None, mlp_dim: int = 2048,
                        depth: int = 1, seed: int = 18, prefix: str = "") -> "OrderedDict[str, np.ndarray]":
    """Temporal transformer ``Transformer_v1`` weights (few_shot.py:979-989; Attention_qkv
    :1035-1055 -- to_q/k/v have NO bias, to_out has one; FeedForward :1643-1652)."""
    dim_head = dim // 8 if dim_head is None else dim_head
    inner = heads * dim_head
    tag = "context2/%d/%d/%d" % (dim, inner, mlp_dim)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()

    def put(key, shape, std, mean=0.0):
        sd[prefix + key] = tensor(shape, tag + "/" + key, std=std, mean=mean, seed=seed)

    for d in range(depth):
        b = "layers.%d." % d
        put(b + "0.norm.weight", (dim,), 0.1, 1.0)
        put(b + "0.norm.bias", (dim,), 0.1)
        put(b + "0.fn.to_q.weight", (inner, dim), 1.5 * dim ** -0.5)
        put(b + "0.fn.to_k.weight", (inner, dim), 1.5 * dim ** -0.5)
        put(b + "0.fn.to_v.weight", (inner, dim), dim ** -0.5)
        put(b + "0.fn.to_out.0.weight", (dim, inner), 0.5 * inner ** -0.5)
        put(b + "0.fn.to_out.0.bias", (dim,), 0.05)
        put(b + "1.net.0.weight", (mlp_dim, dim), dim ** -0.5)
        put(b + "1.net.0.bias", (mlp_dim,), 0.1)
        put(b + "1.net.3.weight", (dim, mlp_dim), 0.5 * mlp_dim ** -0.5)
        put(b + "1.net.3.bias", (dim,), 0.05)
    return sd


def apply_outlier_channels(sd, prefix: str, channels, gain: float, shift: float):
    """Trained-CLIP-like activation statistics on a random-init ViT: ``ln_pre`` (few_shot.py:677) gets weight x ``gain`` and
    bias + ``shift`` on the given channels, so the residual stream carries a few channels with |x| ~ gain + shift (the "massive
    activation" channels of trained ViTs) and a row mean away from zero through every block -- the regime where the LayerNorm-folded
    GEMM's (x Wg - mean c) / std cancels large terms."""
    w, b = sd[prefix + "ln_pre.weight"].copy(), sd[prefix + "ln_pre.bias"].copy()
    for c in channels:
        w[c] *= gain
        b[c] += shift
    sd[prefix + "ln_pre.weight"], sd[prefix + "ln_pre.bias"] = w.astype(np.float32), b.astype(np.float32)
    return sd


def head_state_dict(arch: str, seed: int = 18, depth: int = 1, mlp_dim: int = 2048, outliers=None):
    """Full ``CNN_OTAM_CLIPFSAR`` state dict (keys as in SURVEY.md section 5 minus the
    ``head.`` prefix that BaseVideoModel adds).  ``outliers`` = dict(channels, gain, shift): see apply_outlier_channels."""
    E = ARCHS[arch]["embed"]
    sd = OrderedDict()
    sd["scale"] = np.ones((1,), np.float32)                      # few_shot.py:2733-2734
    sd.update(visual_state_dict(arch, seed, prefix="backbone."))
    if outliers:
        apply_outlier_channels(sd, "backbone.", outliers["channels"], float(outliers["gain"]), float(outliers["shift"]))
    sd.update(context2_state_dict(E, 8, E // 8, mlp_dim, depth, seed, prefix="context2."))
    return sd


def text_features(n_classes: int, embed: int, split: str, seed: int = 18) -> np.ndarray:
    """Stand-in for ``backbone.encode_text(tokenize(prompts))`` (few_shot.py:2714-2728):
    an [n_classes, E] fp32 table.  The text tower is init-time only;
    the per-episode path only gathers rows of this table (:2946)."""
    return tensor((n_classes, embed), "text/%s" % split, std=1.0, seed=seed)


# --------------------------------------------------------------------------------------
# episodes (A0 input contract)
# --------------------------------------------------------------------------------------
def _perm(n: int, name: str, seed: int) -> np.ndarray:
    keys = pseudo_normal(n, name, seed)
    return np.argsort(keys, kind="stable")


def make_episode(way: int = 5, shot: int = 1, query_per_class: int = 1, frames: int = 8, res: int = 224,
                 n_test_classes: int = 24, episode: int = 0, seed: int = 18, dtype=np.float32,
                 lowfreq: float = 0.0):
    """One synthetic episode with the layout of Ssv2_few_shot.__getitem__
    (datasets/base/ssv2_few_shot.py:275-285) after the loader batch dim is stripped
    (runs/test_net_few_shot.py:62).

    Frames are *structured* (SURVEY.md 8(d)): a per-real-class base pattern + per-video
    + per-frame noise in CLIP-normalised pixel space, so that logits carry signal.
    Videos are shuffled independently in the support and target lists (:267-273).

    ``lowfreq`` > 0 adds a per-class coarse pattern (8x8 cells per channel, nearest-upsampled): a convolutional tower
    with global pooling averages white noise away, so the RN50 cases need class signal below the stem's cut-off."""
    es = seed * 1000003 + episode
    S, Q, T = way * shot, way * query_per_class, frames
    classes = _perm(n_test_classes, "ep/classes", es)[:way]          # batch_class_list
    npx = 3 * res * res

    def video(real_cls: int, vid_tag: str) -> np.ndarray:
        base = pseudo_normal(npx, "frame/class%d" % real_cls, seed)           # same across episodes
        if lowfreq:
            cell = -(-res // 8)
            coarse = pseudo_normal(3 * 64, "frame/class_lowfreq%d" % real_cls, seed).reshape(3, 8, 8)
            coarse = np.repeat(np.repeat(coarse, cell, 1), cell, 2)[:, :res, :res]
            base = base + lowfreq * coarse.reshape(-1)
        vn = pseudo_normal(npx, "frame/video/" + vid_tag, es)
        out = np.empty((T, npx), dtype=np.float64)
        for t in range(T):
            fn = pseudo_normal(npx, "frame/frame/%s/%d" % (vid_tag, t), es)
            # temporal structure: the class pattern is circularly shifted a little per frame
            out[t] = 1.0 * np.roll(base, 3 * t) + 0.3 * vn + 0.2 * fn
        return out.reshape(T, 3, res, res).astype(dtype)

    sup, sup_lab, sup_real = [], [], []
    tgt, tgt_lab, tgt_real = [], [], []
    for ci in range(way):
        rc = int(classes[ci])
        for k in range(shot):
            sup.append(video(rc, "s%d_%d" % (ci, k)))
            sup_lab.append(float(ci))
            sup_real.append(float(rc))
        for k in range(query_per_class):
            tgt.append(video(rc, "q%d_%d" % (ci, k)))
            tgt_lab.append(float(ci))
            tgt_real.append(float(rc))
    ps = _perm(S, "ep/shuffle_s", es)
    pq = _perm(Q, "ep/shuffle_q", es)
    return {
        "support_set": np.concatenate([sup[i] for i in ps], 0),
        "support_labels": np.asarray([sup_lab[i] for i in ps], np.float32),
        "target_set": np.concatenate([tgt[i] for i in pq], 0),
        "target_labels": np.asarray([tgt_lab[i] for i in pq], np.float32),
        "real_support_labels": np.asarray([sup_real[i] for i in ps], np.float32),
        "real_target_labels": np.asarray([tgt_real[i] for i in pq], np.float32),
        "batch_class_list": classes.astype(np.float32),
    }
