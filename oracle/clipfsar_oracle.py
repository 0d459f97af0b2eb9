"""TEST INFRASTRUCTURE -- CPU restatement (torch fp32, plain tensor ops) of the
CLIP-FSAR episodic-inference hot path of the reference, function by function.

This file is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``clip-fsar_amd/``) never imports it and has no CPU fallback.

Pinning: every function below is checked in ``tests/test_oracle_golden.py`` against
golden vectors produced by importing the real reference in the dev container
(``oracle/make_golden.py`` + ``oracle/ref_harness.py``).  The reference itself has no
tests / golden vectors for this path, so the pin is "outputs of the
reference itself run here".

All citations are to /root/reference/models/base/few_shot.py unless stated otherwise.
"""
from __future__ import annotations

import math

import torch


# ------------------------------------------------------------------ A3  LayerNorm (:605-611)
def layer_norm(x, weight, bias, eps: float = 1e-5):
    """nn.LayerNorm over the last dim, fp32 statistics, biased variance (:605-611)."""
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * weight + bias


def quick_gelu(x):
    """QuickGELU: x * sigmoid(1.702 x) (:614-616)."""
    return x * torch.sigmoid(1.702 * x)


def gelu_erf(x):
    """nn.GELU() default (exact erf form) used by FeedForward (:1647)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


# ------------------------------------------------------------------ A2  patch embed (:672-676)
def patch_embed(frames, conv_w, cls, pos, patch: int):
    """conv1 (stride = kernel = patch, no bias) as a GEMM over non-overlapping patches,
    patches row-major over the grid, patch vector flattened channel-major then dy then dx;
    then [cls ; patches] + positional embedding (:672-676).  frames [F,3,H,W] -> [F,N,D]."""
    F_, C, H, W = frames.shape
    g = H // patch
    D = conv_w.shape[0]
    p = frames.reshape(F_, C, g, patch, g, patch).permute(0, 2, 4, 1, 3, 5).reshape(F_, g * g, C * patch * patch)
    tok = p @ conv_w.reshape(D, -1).t()                                    # [F, g*g, D]
    x = torch.cat([cls.reshape(1, 1, D).expand(F_, 1, D), tok], dim=1)
    return x + pos


# ------------------------------------------------------------------ A4-A6 residual block (:619-640)
def resblock(x, sd, pre: str, heads: int):
    """x [F,N,D].  x += out_proj(MHA(ln_1 x)); x += c_proj(QuickGELU(c_fc(ln_2 x)))  (:637-640).
    nn.MultiheadAttention semantics (:623,:635): packed in_proj [3D,D]+bias, per-head
    softmax(q k^T / sqrt(hd)) v, no mask, no dropout."""
    F_, N, D = x.shape
    hd = D // heads
    h = layer_norm(x, sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"])
    qkv = h @ sd[pre + "attn.in_proj_weight"].t() + sd[pre + "attn.in_proj_bias"]
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(F_, N, heads, hd).transpose(1, 2)
    k = k.reshape(F_, N, heads, hd).transpose(1, 2)
    v = v.reshape(F_, N, heads, hd).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd)), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(F_, N, D)
    x = x + o @ sd[pre + "attn.out_proj.weight"].t() + sd[pre + "attn.out_proj.bias"]
    h = layer_norm(x, sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"])
    u = quick_gelu(h @ sd[pre + "mlp.c_fc.weight"].t() + sd[pre + "mlp.c_fc.bias"])
    return x + u @ sd[pre + "mlp.c_proj.weight"].t() + sd[pre + "mlp.c_proj.bias"]


# ------------------------------------------------------------------ A1  VisionTransformer.forward (:671-688)
def vit_forward(frames, sd, arch, prefix: str = "backbone.", chunk: int = 40, taps=None):
    """frames [F,3,res,res] fp32 -> features [F,E].  ``arch`` = dict(width, layers, heads, patch, ...).
    ``taps`` (optional dict) receives intermediates for the first chunk (test use)."""
    outs = []
    for s in range(0, frames.shape[0], chunk):
        f = frames[s:s + chunk]
        x = patch_embed(f, sd[prefix + "conv1.weight"], sd[prefix + "class_embedding"],
                        sd[prefix + "positional_embedding"], arch["patch"])
        x = layer_norm(x, sd[prefix + "ln_pre.weight"], sd[prefix + "ln_pre.bias"])        # :677
        if taps is not None and s == 0:
            taps["ln_pre"] = x.clone()
        for i in range(arch["layers"]):                                                      # :679-681
            x = resblock(x, sd, "%stransformer.resblocks.%d." % (prefix, i), arch["heads"])
            if taps is not None and s == 0:
                taps["block%d" % i] = x.clone()
        c = layer_norm(x[:, 0, :], sd[prefix + "ln_post.weight"], sd[prefix + "ln_post.bias"])  # :683
        outs.append(c @ sd[prefix + "proj"])                                                 # :685-686
    return torch.cat(outs, 0)


# ------------------------------------------------------------------ N3 ModifiedResNet (:182-227, :435-539, :542-602)
def _conv_bn(x, sd, conv, bn, padding=0, stride=1, relu=True):
    x = torch.nn.functional.conv2d(x, sd[conv + ".weight"], None, stride=stride, padding=padding)
    x = torch.nn.functional.batch_norm(x, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"],
                                       sd[bn + ".bias"], training=False, eps=1e-5)
    return torch.relu(x) if relu else x


def resnet_forward(frames, sd, arch, prefix: str = "backbone.", chunk: int = 40):
    """CLIP ModifiedResNet eval forward: 3-conv stem + avgpool (:582-587), Bottlenecks with avgpool anti-aliasing
    (:213-226), AttentionPool2d with the mean token as the only query (:515-538, spatial=False)."""
    outs = []
    width, layers, heads = arch["width"], arch["layers"], arch["heads"]
    P = prefix
    for s0 in range(0, frames.shape[0], chunk):
        x = frames[s0:s0 + chunk]
        x = _conv_bn(x, sd, P + "conv1", P + "bn1", padding=1, stride=2)
        x = _conv_bn(x, sd, P + "conv2", P + "bn2", padding=1)
        x = _conv_bn(x, sd, P + "conv3", P + "bn3", padding=1)
        x = torch.nn.functional.avg_pool2d(x, 2)
        inplanes = width
        for li, (planes, blocks) in enumerate(zip((width, width * 2, width * 4, width * 8), layers), start=1):
            for bi in range(blocks):
                stride = 2 if (li > 1 and bi == 0) else 1
                b = "%slayer%d.%d." % (P, li, bi)
                out = _conv_bn(x, sd, b + "conv1", b + "bn1")
                out = _conv_bn(out, sd, b + "conv2", b + "bn2", padding=1)
                if stride > 1:
                    out = torch.nn.functional.avg_pool2d(out, stride)
                out = _conv_bn(out, sd, b + "conv3", b + "bn3", relu=False)
                idn = x
                if stride > 1 or inplanes != planes * 4:
                    idn = torch.nn.functional.avg_pool2d(x, stride) if stride > 1 else x
                    idn = _conv_bn(idn, sd, b + "downsample.0", b + "downsample.1", relu=False)
                x = torch.relu(out + idn)
                inplanes = planes * 4
        F_, C = x.shape[0], x.shape[1]
        t = x.flatten(start_dim=2).permute(0, 2, 1)                                   # [F, HW, C]
        t = torch.cat([t.mean(dim=1, keepdim=True), t], dim=1) + sd[P + "attnpool.positional_embedding"]
        hd = C // heads
        q = (t[:, :1] @ sd[P + "attnpool.q_proj.weight"].t() + sd[P + "attnpool.q_proj.bias"]).reshape(F_, 1, heads, hd).transpose(1, 2)
        k = (t @ sd[P + "attnpool.k_proj.weight"].t() + sd[P + "attnpool.k_proj.bias"]).reshape(F_, -1, heads, hd).transpose(1, 2)
        v = (t @ sd[P + "attnpool.v_proj.weight"].t() + sd[P + "attnpool.v_proj.bias"]).reshape(F_, -1, heads, hd).transpose(1, 2)
        att = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
        o = (att @ v).transpose(1, 2).reshape(F_, C)
        outs.append(o @ sd[P + "attnpool.c_proj.weight"].t() + sd[P + "attnpool.c_proj.bias"])
    return torch.cat(outs, 0)


# ------------------------------------------------------------------ A11 Transformer_v1 (:971-999,:1035-1073,:1643-1654)
def context2_forward(x, sd, heads: int = 8, prefix: str = "context2.", depth: int = 1):
    """x [B,L,E] used as q=k=v.  Layer 0: ONE shared LayerNorm for q,k,v (:971-977); to_q/k/v
    without bias, to_out with bias (:1046-1053); scale = dim_head**-0.5 (:1042); residual on q;
    FeedForward Linear-GELU(erf)-Linear + residual (:994-995).  Layers >= 1 self-attend on the
    output (:996-999)."""
    B, L, E = x.shape
    for d in range(depth):
        p = "%slayers.%d." % (prefix, d)
        n = layer_norm(x, sd[p + "0.norm.weight"], sd[p + "0.norm.bias"])
        inner = sd[p + "0.fn.to_q.weight"].shape[0]
        hd = inner // heads
        q = (n @ sd[p + "0.fn.to_q.weight"].t()).reshape(B, L, heads, hd).transpose(1, 2)
        k = (n @ sd[p + "0.fn.to_k.weight"].t()).reshape(B, L, heads, hd).transpose(1, 2)
        v = (n @ sd[p + "0.fn.to_v.weight"].t()).reshape(B, L, heads, hd).transpose(1, 2)
        a = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
        o = (a @ v).transpose(1, 2).reshape(B, L, inner)
        y = o @ sd[p + "0.fn.to_out.0.weight"].t() + sd[p + "0.fn.to_out.0.bias"] + x
        u = gelu_erf(y @ sd[p + "1.net.0.weight"].t() + sd[p + "1.net.0.bias"])
        x = u @ sd[p + "1.net.3.weight"].t() + sd[p + "1.net.3.bias"] + y
    return x


# ------------------------------------------------------------------ A13 cos_sim (:1115-1124)
def cos_sim(x, y, epsilon: float = 0.01):
    """(x y^T) / (|x| |y|^T + eps): eps is ADDED to the product of norms, no clamp."""
    num = x @ y.transpose(-1, -2)
    xn = torch.linalg.vector_norm(x, dim=-1, keepdim=True)
    yn = torch.linalg.vector_norm(y, dim=-1, keepdim=True)
    return num / (xn @ yn.transpose(-1, -2) + epsilon)


# ------------------------------------------------------------------ A14 OTAM_cum_dist_v2 (:2657-2687)
def otam_cum_dist(dists, lbda: float = 0.5):
    """dists [Q,C,T,T'] -> [Q,C].  Zero-pad one column each side (:2663); first row is a plain
    running sum (:2668-2671); remaining rows use the un-stabilised soft-min
    -lbda*log(sum exp(-c/lbda)) with 3 predecessors in column 1 and in the last column,
    2 in the middle columns (:2675-2685)."""
    d = torch.nn.functional.pad(dists, (1, 1), "constant", 0.0)
    Qn, Cn, R, M = d.shape
    c = torch.zeros_like(d)
    for m in range(1, M):
        c[:, :, 0, m] = d[:, :, 0, m] + c[:, :, 0, m - 1]
    for l in range(1, R):
        c[:, :, l, 1] = d[:, :, l, 1] - lbda * torch.log(
            torch.exp(-c[:, :, l - 1, 0] / lbda) + torch.exp(-c[:, :, l - 1, 1] / lbda)
            + torch.exp(-c[:, :, l, 0] / lbda))
        for m in range(2, M - 1):
            c[:, :, l, m] = d[:, :, l, m] - lbda * torch.log(
                torch.exp(-c[:, :, l - 1, m - 1] / lbda) + torch.exp(-c[:, :, l, m - 1] / lbda))
        c[:, :, l, M - 1] = d[:, :, l, M - 1] - lbda * torch.log(
            torch.exp(-c[:, :, l - 1, M - 2] / lbda) + torch.exp(-c[:, :, l - 1, M - 1] / lbda)
            + torch.exp(-c[:, :, l, M - 2] / lbda))
    return c[:, :, -1, -1]


# ------------------------------------------------------------------ A12 class means (:1127-1136, :2949-2962)
def class_means(x, labels):
    """mean of x[s] over the supports of each class, classes in ascending label order
    (torch.unique sorts, :2950,:2958)."""
    uniq = torch.unique(labels)
    return torch.stack([x[labels == c].mean(dim=0) for c in uniq]), uniq


# ------------------------------------------------------------------ A9-A15 head forward, eval default branch (:2932-2990)
def head_forward(episode, sd, text_train, text_test, arch, frames: int, merge_before: bool = False,
                 single_direct: bool = False, depth: int = 1, taps=None):
    """episode: the A0 dict of torch tensors.  Returns {'logits' [Q,way], 'class_logits' [(S+Q),n_train]}."""
    T = frames
    sup_lab = episode["support_labels"]
    tower = resnet_forward if arch.get("kind") == "rn" else vit_forward
    feats_s = tower(episode["support_set"], sd, arch)                            # :2761
    feats_q = tower(episode["target_set"], sd, arch)                             # :2762
    E = feats_s.shape[-1]
    Fs = feats_s.reshape(-1, T, E)                                               # :2765
    Fq = feats_q.reshape(-1, T, E)                                               # :2764
    # aux class logits -- TRAIN text table even in eval (:2937-2939)
    cls_in = torch.cat([Fs, Fq], dim=0).mean(dim=1)
    class_logits = cos_sim(cls_in, text_train) * sd["scale"]
    ctx = text_test[episode["real_support_labels"].long()].unsqueeze(1)          # :2946
    Fq2 = context2_forward(Fq, sd, depth=depth)                                  # :2948
    if merge_before:                                                             # :2949-2954
        Fs, _ = class_means(Fs, sup_lab)
        ctx, _ = class_means(ctx, sup_lab)
    Fs2 = context2_forward(torch.cat([Fs, ctx], dim=1), sd, depth=depth)[:, :T, :]   # :2955-2956
    if not merge_before:                                                         # :2957-2962
        Fs2, _ = class_means(Fs2, sup_lab)
    nq, ns = Fq2.shape[0], Fs2.shape[0]
    sim = cos_sim(Fq2.reshape(nq * T, E), Fs2.reshape(ns * T, E))                # :2973
    dists = (1.0 - sim).reshape(nq, T, ns, T).permute(0, 2, 1, 3)                # :2974-2976
    cum = otam_cum_dist(dists)                                                   # :2979-2982
    if not single_direct:
        cum = cum + otam_cum_dist(dists.transpose(-1, -2))
    # :2986-2988 -- class c <- column c (unique labels are sorted, so this is the identity)
    logits = -cum
    if taps is not None:
        taps.update(feats_s=feats_s, feats_q=feats_q, ctx_q=Fq2, protos=Fs2, dists=dists, cum_dists=cum)
    return {"logits": logits, "class_logits": class_logits}


# ------------------------------------------------------------------ N4 EVAL_TEXT (:2835-2852) / COMBINE (:2855-2930)
def text_match_probs(feats_q, text_test, support_labels, real_support_labels, scale):
    """feats_q [Q,T,E].  Class-mean of text_test[real labels] in ascending label order, both sides L2-normalised,
    softmax over classes of scale * cosine (:2836-2849)."""
    txt, _ = class_means(text_test[real_support_labels.long()], support_labels)
    img = feats_q.mean(dim=1)
    img = img / img.norm(dim=1, keepdim=True)
    txt = txt / txt.norm(dim=1, keepdim=True)
    return torch.softmax(scale * img @ txt.t(), dim=1)


def head_forward_text_modes(episode, sd, text_train, text_test, arch, frames: int, mode: str, merge_before: bool = False,
                            single_direct: bool = False, depth: int = 1, text_coff: float = 0.9):
    """mode 'eval_text': logits = softmax probabilities; mode 'combine': p_text^coff * softmax((8-cum)/8)^(1-coff)."""
    T = frames
    feats_q = vit_forward(episode["target_set"], sd, arch).reshape(-1, T, arch["embed"])
    p_text = text_match_probs(feats_q, text_test, episode["support_labels"], episode["real_support_labels"], sd["scale"])
    if mode == "eval_text":
        return {"logits": p_text, "class_logits": None}
    vis = head_forward(episode, sd, text_train, text_test, arch, frames, merge_before, single_direct, depth)["logits"]
    soft = torch.softmax((8.0 + vis) / 8.0, dim=1)                    # (8 - cum)/8 with cum = -vis (:2921)
    return {"logits": p_text.pow(text_coff) * soft.pow(1.0 - text_coff), "class_logits": None}


# ------------------------------------------------------------------ N2 test-time frame transform
def preprocess_frames(frames_u8, scale_hw, crop, y0, x0, mean, std):
    """uint8 [T,H,W,3] -> [T,3,crop,crop].  ToTensorVideo (/255, THWC->CTHW), F.interpolate(size=scale_hw, bilinear,
    align_corners=False) as KineticsResizedCropFewshot does (datasets/utils/transformations.py:676-688), crop window,
    NormalizeVideo, permute(1,0,2,3) (datasets/base/ssv2_few_shot.py:439,627-642)."""
    clip = frames_u8.permute(3, 0, 1, 2).float() / 255.0
    clip = torch.nn.functional.interpolate(clip, size=(int(scale_hw[0]), int(scale_hw[1])), mode="bilinear")
    clip = clip[:, :, y0:y0 + crop, x0:x0 + crop]
    m = torch.tensor(mean).reshape(3, 1, 1, 1)
    sd = torch.tensor(std).reshape(3, 1, 1, 1)
    return ((clip - m) / sd).permute(1, 0, 2, 3).contiguous()


def top1_correct(logits, target_labels):
    """metrics.topks_correct(...,(1,)) (reference utils/metrics.py:100-138): count of argmax hits."""
    return int((logits.argmax(dim=1) == target_labels.long()).sum())


# ------------------------------------------------------------------ N1 CLIP.encode_text (:793-806), causal mask :778-784
def text_resblock(x, sd, pre: str, heads: int):
    """ResidualAttentionBlock with the text tower's additive causal mask (-inf above the diagonal)."""
    n, L, W = x.shape
    hd = W // heads
    h = layer_norm(x, sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"])
    qkv = h @ sd[pre + "attn.in_proj_weight"].t() + sd[pre + "attn.in_proj_bias"]
    q, k, v = [t.reshape(n, L, heads, hd).transpose(1, 2) for t in qkv.split(W, dim=-1)]
    mask = torch.full((L, L), float("-inf")).triu_(1)
    att = torch.softmax((q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd)) + mask, dim=-1)
    o = (att @ v).transpose(1, 2).reshape(n, L, W)
    x = x + o @ sd[pre + "attn.out_proj.weight"].t() + sd[pre + "attn.out_proj.bias"]
    h = layer_norm(x, sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"])
    u = quick_gelu(h @ sd[pre + "mlp.c_fc.weight"].t() + sd[pre + "mlp.c_fc.bias"])
    return x + u @ sd[pre + "mlp.c_proj.weight"].t() + sd[pre + "mlp.c_proj.bias"]


def encode_text(tokens, sd):
    """tokens [n, L] integer ids -> [n, embed].  token embedding + positional embedding (:794-796), causal transformer,
    ln_final (:800), features at the position of the largest id = EOT (:804), @ text_projection."""
    tokens = tokens.long()
    W = sd["token_embedding.weight"].shape[1]
    heads = W // 64                                                      # build_model convention (:867)
    layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"][: tokens.shape[1]]
    for i in range(layers):
        x = text_resblock(x, sd, "transformer.resblocks.%d." % i, heads)
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    return x[torch.arange(x.shape[0]), tokens.argmax(dim=-1)] @ sd["text_projection"]
