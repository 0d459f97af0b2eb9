"""N4: EVAL_TEXT and COMBINE eval branches (few_shot.py:2835-2930).  CPU: oracle == reference golden.
GPU: HIP engine == reference golden, through the engine and through the registered head with the cfg flags."""
import json
import os

import numpy as np
import pytest
import torch

import clip_fsar_amd.synth as synth
import clipfsar_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["n4_evaltext_5w2s_T4", "n4_combine_5w1s_T8", "n4_combine_5w3s_T4_mb_c05"]


def _load(name):
    z = np.load(os.path.join(GOLD, "head_%s.npz" % name))
    m = json.loads(str(z["meta"]))
    a = synth.ARCHS[m["arch"]]
    sd = {k: torch.from_numpy(v) for k, v in synth.head_state_dict(m["arch"], seed=m["seed"]).items()}
    sd["scale"] = torch.tensor([m["scale"]])
    tt = torch.from_numpy(synth.text_features(m["n_train"], a["embed"], "train", m["seed"]))
    te = torch.from_numpy(synth.text_features(m["n_test"], a["embed"], "test", m["seed"]))
    ep = {k: torch.from_numpy(v) for k, v in synth.make_episode(m["way"], m["shot"], m["q"], m["T"], a["res"], m["n_test"],
                                                                 m["episode"], m["seed"]).items()}
    return m, a, sd, tt, te, ep, torch.from_numpy(z["logits"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_text_modes_match_reference(name):
    m, a, sd, tt, te, ep, ref = _load(name)
    with torch.no_grad():
        out = orc.head_forward_text_modes(ep, sd, tt, te, a, m["T"], m["mode"], merge_before=m.get("merge_before", False),
                                          text_coff=m.get("text_coff", 0.9))
    assert out["class_logits"] is None
    assert float((out["logits"] - ref).abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_text_modes_match_reference(name):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from types import SimpleNamespace as NS
    import clip_fsar_amd.models.base  # noqa: F401
    from clip_fsar_amd.models.base.builder import build_model
    m, a, sd, tt, te, ep, ref = _load(name)
    train = NS(CLASS_NAME=["c"] * m["n_train"], WAY=m["way"])
    if m["mode"] == "eval_text":
        train.EVAL_TEXT = True
    else:
        train.COMBINE = True
    if m.get("merge_before"):
        train.MERGE_BEFORE = True
    if "text_coff" in m:
        train.TEXT_COFF = m["text_coff"]
    cfg = NS(VIDEO=NS(HEAD=NS(NAME="CNN_OTAM_CLIPFSAR", BACKBONE_NAME=m["arch"], PRECISION="fp32"),
                      BACKBONE=NS(META_ARCH="Identity")), TRAIN=train, TEST=NS(CLASS_NAME=["t"] * m["n_test"]),
             DATA=NS(NUM_INPUT_FRAMES=m["T"]), MODEL=NS(NAME="BaseVideoModel", EMA=NS(ENABLE=False)), BN=NS(FREEZE=False),
             NUM_GPUS=1, NUM_SHARDS=1, RANDOM_SEED=m["seed"])
    model, _ = build_model(cfg)
    model.eval()
    with torch.no_grad():
        model.head.scale.fill_(m["scale"])
        out = model({k: v.cuda() for k, v in ep.items()})
    assert out["class_logits"] is None
    assert float((out["logits"].cpu() - ref).abs().max()) < 1e-4
