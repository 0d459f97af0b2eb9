"""TEST INFRASTRUCTURE -- dev-container only.  Imports the *real* reference
(/root/reference, read-only, never shipped) so that

  * the torch-fp32 restatement in ``oracle/clipfsar_oracle.py`` can be pinned against it, and
  * golden input/output vectors can be generated (``oracle/make_golden.py`` -> tests/golden/).

Nothing under ``clip-fsar_amd/`` may import this module.  It does not exist for the GPU box:
``/root/reference`` is absent there, and the functions below raise if it is missing.

How the reference is made importable: stub modules for the
non-arithmetic dependencies the image lacks (torchvision, ipdb, ftfy, oss2, simplejson,
decord, joblib), ``sys.dont_write_bytecode`` (the tree is read-only),
``few_shot.load`` replaced by a no-network constructor, ``Tensor.cuda`` neutralised.
None of the stubs touches hot-path arithmetic.
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import torch

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models", "base"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules.setdefault(name, m)
    return sys.modules[name]


_fs = None


def import_reference():
    """Returns the reference module ``models.base.few_shot`` (few_shot.py)."""
    global _fs
    if _fs is not None:
        return _fs
    if not available():
        raise RuntimeError("reference tree %s not present (GPU box?)" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    ident = lambda *a, **k: None
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models")
    tv.utils = _stub("torchvision.utils", make_grid=ident, save_image=ident)
    tv.transforms = _stub(
        "torchvision.transforms", Compose=ident, Resize=ident, CenterCrop=ident, ToTensor=ident,
        Normalize=ident, InterpolationMode=SimpleNamespace(BICUBIC=3))
    tv.transforms.__path__ = []                          # let `import torchvision.transforms._functional_video` resolve
    _stub("torchvision.transforms._functional_video")
    _stub("torchvision.transforms._transforms_video")
    tv.transforms.Lambda = ident
    _stub("ipdb", set_trace=ident)
    _stub("ftfy", fix_text=lambda s: s)
    _stub("oss2")
    _stub("simplejson")
    _stub("decord")
    _stub("joblib")
    # our own package dir may shadow `models`/`utils`; the reference must win here
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "utils"
              or k.startswith("utils.")]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    import models.base.few_shot as fs  # noqa: E402  (the reference)
    assert fs.__file__.startswith(REFERENCE_ROOT), fs.__file__
    torch.Tensor.cuda = lambda self, *a, **k: self          # few_shot.py:2719,2726 hard-code .cuda()
    _fs = fs
    return fs


def to_torch_sd(sd):
    return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


def make_cfg(arch="ViT-B/16", way=5, shot=1, frames=8, n_train=64, n_test=24, merge_before=False,
             depth=None, single_direct=False, eval_text=False, combine=False, text_coff=None):
    """SimpleNamespace tree with exactly the attributes the head reads."""
    train = SimpleNamespace(CLASS_NAME=["train class %d" % i for i in range(n_train)], WAY=way, SHOT=shot,
                            BATCH_SIZE=1)
    if merge_before:
        train.MERGE_BEFORE = True
    if depth:
        train.TRANSFORMER_DEPTH = depth
    if single_direct:
        train.SINGLE_DIRECT = True
    if eval_text:
        train.EVAL_TEXT = True
    if combine:
        train.COMBINE = True
    if text_coff is not None:
        train.TEXT_COFF = text_coff
    test = SimpleNamespace(CLASS_NAME=["test class %d" % i for i in range(n_test)])
    return SimpleNamespace(
        VIDEO=SimpleNamespace(HEAD=SimpleNamespace(NAME="CNN_OTAM_CLIPFSAR", BACKBONE_NAME=arch),
                              BACKBONE=SimpleNamespace(META_ARCH="Identity")),
        TRAIN=train, TEST=test, DATA=SimpleNamespace(NUM_INPUT_FRAMES=frames),
        MODEL=SimpleNamespace(NAME="BaseVideoModel", EMA=SimpleNamespace(ENABLE=False)),
        BN=SimpleNamespace(SYNC_BN=False, FREEZE=False), NUM_GPUS=0, NUM_SHARDS=1, PAI=False)


class _FakeCLIP:
    """What ``load()`` returns as far as the head's __init__ uses it (few_shot.py:2706-2728):
    ``.visual`` and ``.encode_text``.  The text tower is init-time only;
    the harness injects the synthetic text tables afterwards."""

    def __init__(self, visual, embed):
        self.visual = visual
        self._embed = embed

    def encode_text(self, tokens):
        return torch.zeros(tokens.shape[0], self._embed)


def build_reference_vit(arch_params, sd_np):
    """The reference's own VisionTransformer (few_shot.py:654-688) with the given weights."""
    fs = import_reference()
    a = arch_params
    vit = fs.VisionTransformer(input_resolution=a["res"], patch_size=a["patch"], width=a["width"],
                               layers=a["layers"], heads=a["heads"], output_dim=a["embed"]).float().eval()
    missing = vit.load_state_dict(to_torch_sd(sd_np), strict=True)
    return vit


def build_reference_head(cfg, arch_params, head_sd_np, text_train, text_test):
    """The reference's CNN_OTAM_CLIPFSAR (few_shot.py:2690-2993), eval mode, with weights and
    text tables injected.  For archs the head rejects (ViT-L/14, test archs) the name is
    temporarily presented as "ViT-B/16" and ``mid_dim``/``context2`` are rebuilt by the same
    formula (few_shot.py:2737-2739) -- extension A16."""
    fs = import_reference()
    a = arch_params
    E = a["embed"]
    if a.get("kind") == "rn":
        vit = fs.ModifiedResNet(layers=a["layers"], output_dim=E, heads=a["heads"], input_resolution=a["res"],
                                width=a["width"]).float()
    else:
        vit = fs.VisionTransformer(input_resolution=a["res"], patch_size=a["patch"], width=a["width"],
                                   layers=a["layers"], heads=a["heads"], output_dim=E).float()
    old_load, old_tok = fs.load, fs.tokenize
    fs.load = lambda name, device="cpu", cfg=None, jit=False: (_FakeCLIP(vit, E), None)
    fs.tokenize = lambda texts, *a_, **k_: torch.zeros(len(texts), 77, dtype=torch.long)
    real_name = cfg.VIDEO.HEAD.BACKBONE_NAME
    try:
        cfg.VIDEO.HEAD.BACKBONE_NAME = "RN50" if E == 1024 else "ViT-B/16"      # the head's own RN50 branch (:2699-2704)
        head = fs.CNN_OTAM_CLIPFSAR(cfg)
    finally:
        cfg.VIDEO.HEAD.BACKBONE_NAME = real_name
        fs.load, fs.tokenize = old_load, old_tok
    if E not in (512, 1024):
        head.mid_dim = E
        depth = int(getattr(cfg.TRAIN, "TRANSFORMER_DEPTH", 0) or 1)
        head.context2 = fs.Transformer_v1(dim=E, heads=8, dim_head_k=E // 8, dropout_atte=0.2, depth=depth)
    res = head.load_state_dict(to_torch_sd(head_sd_np), strict=True)
    head.text_features_train = torch.from_numpy(text_train.copy())
    head.text_features_test = torch.from_numpy(text_test.copy())
    head.float().eval()
    return head


def episode_to_torch(ep):
    return {k: torch.from_numpy(v.copy()) for k, v in ep.items()}
