"""CPU: pin the torch-fp32 restatement (oracle/clipfsar_oracle.py) against golden vectors that
were produced by importing the REAL reference (oracle/make_golden.py).  Tolerances are fp32
round-off class (different summation order only)."""
import json
import os

import numpy as np
import pytest
import torch

import clip_fsar_amd.synth as synth
import clipfsar_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    z = np.load(os.path.join(GOLD, name))
    return {k: z[k] for k in z.files}


def _t(x):
    return torch.from_numpy(np.asarray(x))


def test_known_answers():
    z = _load("known_answers.npz")
    # SURVEY.md section 4 table (values captured from the reference by the survey probe)
    assert np.allclose(z["cos_ones"], 4.0 / (2 * 2 + 0.01))
    assert abs(float(z["otam_zeros"][0, 0]) - (-4.217774391174316)) < 1e-6
    assert abs(float(z["otam_ones"][0, 0]) - 4.203567028045654) < 1e-6
    assert abs(float(z["otam_1meye"][0, 0]) - (-0.37329304218292236)) < 1e-6
    assert abs(float(z["otam_l01_zeros"][0, 0]) - (-0.8435549736022949)) < 1e-6
    # restatement vs reference outputs
    assert torch.allclose(orc.cos_sim(_t(z["cos_x"]), _t(z["cos_y"])), _t(z["cos_xy"]), atol=1e-6)
    assert torch.allclose(orc.otam_cum_dist(torch.zeros(1, 1, 8, 8)), _t(z["otam_zeros"]), atol=1e-6)
    assert torch.allclose(orc.otam_cum_dist(torch.zeros(1, 1, 8, 8), 0.1), _t(z["otam_l01_zeros"]), atol=1e-6)
    assert torch.allclose(orc.otam_cum_dist(_t(z["d8"])), _t(z["otam_d8"]), atol=1e-5)
    assert torch.allclose(orc.otam_cum_dist(_t(z["d16"])), _t(z["otam_d16"]), atol=1e-5)
    r = _t(z["bidir_in"])
    bid = orc.otam_cum_dist(r) + orc.otam_cum_dist(r.transpose(-1, -2))
    assert torch.allclose(bid, _t(z["bidir_out"]), atol=1e-6)
    assert torch.allclose(orc.quick_gelu(_t(z["quickgelu_in"])), _t(z["quickgelu_out"]), atol=1e-6)
    assert torch.allclose(orc.gelu_erf(_t(z["quickgelu_in"])), _t(z["gelu_out"]), atol=1e-6)


def test_vit_layer_taps_tiny():
    z = _load("vit_taps_tiny.npz")
    meta = json.loads(str(z["meta"]))
    a = synth.ARCHS[meta["arch"]]
    sd = {k: _t(v) for k, v in synth.vit_state_dict(meta["arch"], meta["seed"], prefix="backbone.").items()}
    ep = synth.make_episode(frames=meta["frames"], res=a["res"], seed=meta["seed"], episode=meta["episode"])
    frames = _t(ep["support_set"][:meta["n"]])
    taps = {}
    out = orc.vit_forward(frames, sd, a, taps=taps)
    assert torch.allclose(taps["ln_pre"], _t(z["ln_pre"]), atol=2e-5)
    for i in range(a["layers"]):
        assert torch.allclose(taps["block%d" % i], _t(z["block%d" % i]), atol=5e-5), i
    assert torch.allclose(out, _t(z["out"]), atol=5e-5)


SMALL = ["t_5w1s_T8", "t_5w5s_T8_mb", "t_5w5s_q2_T8", "t_5w3s_T16_mb_d2", "t_5w2s_T4_sd",
         "t197_5w1s_T2", "t257_5w1s_T2", "rn_t_5w2s_T4", "t_outlier_5w1s_T8", "t197_outlier_5w1s_T2"]
LARGE = ["cfg2_B16_5w1s_T8", "rn50_5w1s_T2"]


def _run_case(name, atol_feat, atol_logit):
    z = _load("head_%s.npz" % name)
    m = json.loads(str(z["meta"]))
    a = synth.ARCHS[m["arch"]]
    depth = m.get("depth", 1)
    sd = {k: _t(v) for k, v in synth.head_state_dict(m["arch"], seed=m["seed"], depth=depth, outliers=m.get("outliers")).items()}
    tt = _t(synth.text_features(m["n_train"], a["embed"], "train", m["seed"]))
    te = _t(synth.text_features(m["n_test"], a["embed"], "test", m["seed"]))
    ep = synth.make_episode(way=m["way"], shot=m["shot"], query_per_class=m["q"], frames=m["T"], res=a["res"],
                            n_test_classes=m["n_test"], episode=m["episode"], seed=m["seed"], lowfreq=m.get("lowfreq", 0.0))
    ep = {k: _t(v) for k, v in ep.items()}
    taps = {}
    with torch.no_grad():
        out = orc.head_forward(ep, sd, tt, te, a, frames=m["T"], merge_before=m.get("merge_before", False),
                               single_direct=m.get("single_direct", False), depth=depth, taps=taps)
    for k in ("feats_s", "feats_q"):
        assert torch.allclose(taps[k], _t(z[k]), atol=atol_feat), (name, k, float((taps[k] - _t(z[k])).abs().max()))
    for k in ("ctx_q", "protos", "dists"):
        assert torch.allclose(taps[k], _t(z[k]), atol=atol_feat), (name, k, float((taps[k] - _t(z[k])).abs().max()))
    assert torch.allclose(taps["cum_dists"], _t(z["cum_dists"]), atol=atol_logit)
    assert torch.allclose(out["logits"], _t(z["logits"]), atol=atol_logit), float((out["logits"] - _t(z["logits"])).abs().max())
    assert torch.allclose(out["class_logits"], _t(z["class_logits"]), atol=atol_logit)
    # the fixture is not degenerate: logits carry signal
    lg = z["logits"]
    assert lg.max() - lg.min() > 0.3


@pytest.mark.parametrize("name", SMALL)
def test_head_small_cases(name):
    _run_case(name, atol_feat=1e-4, atol_logit=1e-4)


@pytest.mark.slow
@pytest.mark.parametrize("name", LARGE)
def test_head_full_size(name):
    if not os.path.exists(os.path.join(GOLD, "head_%s.npz" % name)):
        pytest.skip("fixture not generated")
    _run_case(name, atol_feat=5e-4, atol_logit=2e-4)


@pytest.mark.slow
@pytest.mark.parametrize("name,episode", [("mc_cfg2_B16_5w1s_T8", 5), ("hc_cfg2_B16_5w1s_T8", 9)])
def test_oracle_matches_the_multi_episode_reference_goldens(name, episode):
    """The multi-episode goldens (oracle/make_golden.py --multi: 13 episodes per configuration run through the REFERENCE, standard and high
    contrast) pin the oracle too: one episode of each cfg2 set, logits and class logits."""
    path = os.path.join(GOLD, "multi_%s.npz" % name)
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    z = np.load(path)
    m = json.loads(str(z["meta"]))
    a = synth.ARCHS[m["arch"]]
    sd = {k: _t(v) for k, v in synth.head_state_dict(m["arch"], seed=m["seed"]).items()}
    tt = _t(synth.text_features(m["n_train"], a["embed"], "train", m["seed"]))
    te = _t(synth.text_features(m["n_test"], a["embed"], "test", m["seed"]))
    ep = {k: _t(v) for k, v in synth.make_episode(way=m["way"], shot=m["shot"], query_per_class=m["q"], frames=m["T"], res=a["res"],
                                                  n_test_classes=m["n_test"], episode=episode, seed=m["seed"],
                                                  lowfreq=m.get("lowfreq", 0.0)).items()}
    with torch.no_grad():
        out = orc.head_forward(ep, sd, tt, te, a, frames=m["T"], merge_before=m.get("merge_before", False))
    assert torch.allclose(out["logits"], _t(z["logits"][episode]), atol=2e-4), float((out["logits"] - _t(z["logits"][episode])).abs().max())
    assert torch.allclose(out["class_logits"], _t(z["class_logits"][episode]), atol=2e-4)
    lg = z["logits"]
    spread = float(np.mean(lg.max((1, 2)) - lg.min((1, 2))))
    assert (spread > 2.5) == name.startswith("hc_"), spread          # the high-contrast set really is one
