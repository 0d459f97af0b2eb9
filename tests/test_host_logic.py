"""CPU: host-side mirror of the reference plugin surface -- registry, builder, BaseVideoModel, head construction and
state-dict naming (against key/shape lists captured from the REAL reference), synthetic generators, metrics, meters."""
import json
import os
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

import clip_fsar_amd.synth as synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _cfg(arch="ViT-test/16", **train):
    return NS(VIDEO=NS(HEAD=NS(NAME="CNN_OTAM_CLIPFSAR", BACKBONE_NAME=arch), BACKBONE=NS(META_ARCH="Identity")),
              TRAIN=NS(CLASS_NAME=["c%d" % i for i in range(64)], WAY=5, **train),
              TEST=NS(CLASS_NAME=["t%d" % i for i in range(24)]), DATA=NS(NUM_INPUT_FRAMES=8),
              MODEL=NS(NAME="BaseVideoModel", EMA=NS(ENABLE=False)), BN=NS(FREEZE=False), NUM_GPUS=0, NUM_SHARDS=1,
              RANDOM_SEED=18)


def test_registry_semantics():
    from clip_fsar_amd.utils.registry import Registry
    R = Registry("T")

    @R.register()
    class Foo:
        pass
    assert R.get("Foo") is Foo and R.get("Bar") is None and "Foo" in R.get_all_registered()
    with pytest.raises(AssertionError):
        R.register()(Foo)


def test_builder_and_model_surface():
    import clip_fsar_amd.models.base  # noqa: F401
    from clip_fsar_amd.models.base.backbone import BACKBONE_REGISTRY, Identity
    from clip_fsar_amd.models.base.base_blocks import HEAD_REGISTRY
    from clip_fsar_amd.models.base.builder import build_model
    from clip_fsar_amd.models.base.models import BaseVideoModel
    assert BACKBONE_REGISTRY.get("Identity") is Identity
    assert HEAD_REGISTRY.get("CNN_OTAM_CLIPFSAR") is not None
    model, ema = build_model(_cfg())
    assert isinstance(model, BaseVideoModel) and ema is None
    assert isinstance(model.backbone, Identity)
    x = {"a": 1}
    assert model.backbone(x) is x
    model.eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        model({"support_set": torch.zeros(40, 3, 64, 64), "support_labels": torch.zeros(5),
               "target_set": torch.zeros(40, 3, 64, 64), "real_support_labels": torch.zeros(5)})
    model.train()
    with pytest.raises(NotImplementedError):
        model({"support_set": torch.zeros(1), "support_labels": None, "target_set": None, "real_support_labels": None})
    rn, _ = build_model(_cfg(arch="RN50"))                       # N3: the CLIP ModifiedResNet tower is built
    assert rn.head.mid_dim == 1024 and "backbone.layer4.2.bn3.running_var" in rn.head.state_dict()
    with pytest.raises(ValueError, match="unsupported BACKBONE_NAME"):
        build_model(_cfg(arch="RN101"))


def test_state_dict_names_and_shapes_match_reference():
    """Key names + shapes of the reference head (captured from the real reference's modules) == ours, so that
    reference-trained .pyth checkpoints load (reference utils/checkpoint.py:329)."""
    import clip_fsar_amd.models.base  # noqa: F401
    from clip_fsar_amd.models.base.base_blocks import HEAD_REGISTRY
    ref = json.load(open(os.path.join(GOLD, "state_dict_keys_B16.json")))
    head = HEAD_REGISTRY.get("CNN_OTAM_CLIPFSAR")(_cfg(arch="ViT-B/16"))
    ours = {k: list(v.shape) for k, v in head.state_dict().items()}
    assert ours == ref
    assert len(ours) == 164
    # text tables are plain attributes, not in the state dict (few_shot.py:2714-2728)
    assert head.text_features_train.shape == (64, 512) and head.text_features_test.shape == (24, 512)
    # RN50 (every shipped config's backbone): conv/BN parameters AND BatchNorm buffers under the reference's names
    ref_rn = json.load(open(os.path.join(GOLD, "state_dict_keys_RN50.json")))
    head_rn = HEAD_REGISTRY.get("CNN_OTAM_CLIPFSAR")(_cfg(arch="RN50"))
    assert {k: list(v.shape) for k, v in head_rn.state_dict().items()} == ref_rn
    assert len(ref_rn) == 351 and head_rn.text_features_test.shape == (24, 1024)
    # depth flag (absent-by-default hasattr semantics)
    head2 = HEAD_REGISTRY.get("CNN_OTAM_CLIPFSAR")(_cfg(TRANSFORMER_DEPTH=2))
    assert any(k.startswith("context2.layers.1.") for k in head2.state_dict())


def test_synth_is_deterministic_and_structured():
    a = synth.tensor((4, 5), "x", std=2.0, mean=1.0, seed=3)
    b = synth.tensor((4, 5), "x", std=2.0, mean=1.0, seed=3)
    assert np.array_equal(a, b) and a.dtype == np.float32
    big = synth.pseudo_normal(200000, "stat", 1)
    assert abs(big.mean()) < 0.01 and abs(big.std() - 1.0) < 0.01
    # slices regenerate independently
    assert np.array_equal(synth.pseudo_normal(10, "stat", 1, offset=100), big[100:110])
    ep = synth.make_episode(way=5, shot=2, query_per_class=1, frames=4, res=32, episode=7)
    assert ep["support_set"].shape == (5 * 2 * 4, 3, 32, 32) and ep["target_set"].shape == (5 * 4, 3, 32, 32)
    assert sorted(ep["support_labels"].tolist()) == sorted([float(i) for i in range(5)] * 2)
    assert set(ep["real_support_labels"].tolist()) == set(ep["batch_class_list"].tolist())
    # label <-> real-class mapping is consistent between support and target lists
    m = {l: r for l, r in zip(ep["support_labels"], ep["real_support_labels"])}
    assert all(m[l] == r for l, r in zip(ep["target_labels"], ep["real_target_labels"]))
    ep2 = synth.make_episode(way=5, shot=2, query_per_class=1, frames=4, res=32, episode=7)
    assert all(np.array_equal(ep[k], ep2[k]) for k in ep)


def test_topks_correct_and_valmeter():
    from clip_fsar_amd.utils import metrics
    from clip_fsar_amd.utils.meters import ValMeter
    preds = torch.tensor([[0.1, 0.9, 0.0, 0.0, 0.0], [0.8, 0.1, 0.0, 0.0, 0.05], [0.0, 0.0, 0.1, 0.2, 0.7]])
    labels = torch.tensor([1.0, 2.0, 4.0])
    c1, c5 = metrics.topks_correct(preds, labels, (1, 5))
    assert float(c1) == 2.0 and float(c5) == 3.0
    e1, e5 = metrics.topk_errors(preds, labels, (1, 5))
    assert abs(float(e1) - 100.0 / 3) < 1e-4 and float(e5) == 0.0
    vm = ValMeter(10, NS(LOG_PERIOD=2))
    vm.update_stats(20.0, 0.0, 1)
    vm.update_stats(40.0, 0.0, 3)
    st = vm.log_epoch_stats(0)
    assert abs(st["top1_err"] - 35.0) < 1e-9 and abs(st["top1_acc"] - 65.0) < 1e-9


def test_install_as_reference_modules():
    import sys
    import clip_fsar_amd
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k in ("models", "utils", "runs", "datasets") or
             k.startswith(("models.", "utils.", "runs.", "datasets."))}
    try:
        clip_fsar_amd.install_as_reference_modules()
        from models.base.builder import build_model  # noqa: F401  (the reference harness's import, :26)
        from utils.meters import ValMeter  # noqa: F401
        from datasets.base.builder import build_loader  # noqa: F401
        import utils.registry as ur
        import clip_fsar_amd.utils.registry as ours
        assert ur is ours
    finally:
        for k in [k for k in sys.modules if k in ("models", "utils", "runs", "datasets") or
                  k.startswith(("models.", "utils.", "runs.", "datasets."))]:
            del sys.modules[k]
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


def test_patch_embed_weight_layouts():
    """hip.patch_embed_weight (pure layout + one rounding): 16 x 16 patches = conv1.weight.reshape(D, 768); 14 x 14 = padded rows, column
    (c 14 + dy) 16 + dx, zeros at dx = 14, 15 and behind column 672 (include/clipfsar_hip.h: cfsar_patch_embed)."""
    import torch
    from clip_fsar_amd import hip
    g = torch.Generator().manual_seed(3)
    w16 = torch.randn(8, 3, 16, 16, generator=g)
    assert torch.equal(hip.patch_embed_weight(w16, 16, torch.float16), w16.reshape(8, 768).half())
    w14 = torch.randn(8, 3, 14, 14, generator=g)
    out = hip.patch_embed_weight(w14, 14, torch.bfloat16)
    assert tuple(out.shape) == (8, 704) and out.dtype == torch.bfloat16
    for (c, dy, dx) in ((0, 0, 0), (1, 5, 13), (2, 13, 7)):
        assert torch.equal(out[:, (c * 14 + dy) * 16 + dx], w14[:, c, dy, dx].bfloat16())
    pad = out.float().reshape(8, 44, 16)
    assert float(pad[:, :42, 14:].abs().max()) == 0.0 and float(pad[:, 42:].abs().max()) == 0.0
    assert hip.PATCH_EMBED_SLOTS == {16: 768, 14: 704}

