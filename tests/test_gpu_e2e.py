"""GPU parity tests, end to end: episode dict -> logits through the HIP engine vs (a) golden vectors generated from
the real reference and (b) the CPU oracle, for every golden case; layer-by-layer ViT taps; batched-episode
consistency; the registry/builder drop-in surface.

Tolerances: fp32 mode -- logits within 1e-3 of the reference (north star); observed ~1e-5.
            fp16_strict mode (round 6) -- the 16-bit mode held to the NORTH-STAR 1e-3 as a bound on every reference golden: each of the 455 logit
            rows of the six multi-episode ViT sets and the three single goldens asserted < 1e-3 (test_strict_mode_*).
            fp16 mode -- a STATISTIC (rms <= 4e-4, p99 <= 1e-3 over 65 rows per configuration), its tail bounded at LOGITS_TOLERANCE["fp16"] = 1.5e-3
            on the multi-episode sets; the single goldens (deterministic, measured 3.8e-4 ... 7.6e-4) are asserted < 1e-3; the tiny test
            architectures are bounded at 2 x their measured deviation (tests/_cases.py: MEASURED_DLOGITS).
            bf16 mode -- deviation is REPORTED; every case is bounded at 2 x its measured deviation (3e-3 ... 2.3e-2 on
            logits whose spread is ~1-3), and the fp32-mode argmax must be reproduced on clearly separated queries.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import clip_fsar_amd.synth as synth  # noqa: E402
import clipfsar_oracle as orc  # noqa: E402
from _cases import OUTLIER_CASES, GOLD, SMALL_CASES, bound, case_inputs, load_golden, maxdiff, multi_case_stats, run_engine  # noqa: E402
from clip_fsar_amd import LOGITS_TOLERANCE, NORTH_STAR_TOLERANCE  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_vit_layer_taps_tiny_fp32():
    z = np.load(os.path.join(GOLD, "vit_taps_tiny.npz"))
    meta = json.loads(str(z["meta"]))
    a = synth.ARCHS[meta["arch"]]
    sd = {k: torch.from_numpy(v) for k, v in synth.vit_state_dict(meta["arch"], meta["seed"]).items()}
    ep = synth.make_episode(frames=meta["frames"], res=a["res"], seed=meta["seed"], episode=meta["episode"])
    frames = torch.from_numpy(ep["support_set"][:meta["n"]]).cuda()
    from clip_fsar_amd.engine import HipViT
    vit = HipViT(a, sd, precision="fp32")
    taps = {}
    out = vit.forward(frames, taps=taps)
    n, N, D = meta["n"], vit.ntok, a["width"]
    assert maxdiff(taps["ln_pre"].cpu().reshape(n, N, D), z["ln_pre"]) < 1e-4
    for i in range(a["layers"]):
        assert maxdiff(taps["block%d" % i].cpu().reshape(n, N, D), z["block%d" % i]) < 2e-4, i
    assert maxdiff(out.cpu(), z["out"]) < 2e-4


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("fp16", 4e-3), ("bf16", 3e-2)])
def test_last_block_class_token_pruning_equals_full_block(precision, tol):
    """VisionTransformer.forward reads only x[:, 0] of the last block's output (few_shot.py:683): the engine computes that block's
    attention output / out_proj / MLP for the class-token rows alone.  Same features as the full block (taps force the full block)."""
    z = np.load(os.path.join(GOLD, "vit_taps_tiny.npz"))
    meta = json.loads(str(z["meta"]))
    a = synth.ARCHS[meta["arch"]]
    sd = {k: torch.from_numpy(v) for k, v in synth.vit_state_dict(meta["arch"], meta["seed"]).items()}
    ep = synth.make_episode(frames=meta["frames"], res=a["res"], seed=meta["seed"], episode=meta["episode"])
    frames = torch.from_numpy(ep["support_set"][:meta["n"]]).cuda()
    from clip_fsar_amd.engine import HipViT
    vit = HipViT(a, sd, precision=precision)
    assert vit.prune_last
    pruned = vit.forward(frames).clone()
    full = vit.forward(frames, taps={}).clone()
    assert maxdiff(pruned, full) < tol * max(1.0, float(full.abs().max()))
    if precision == "fp32":
        assert maxdiff(pruned.cpu(), z["out"]) < 2e-4


@pytest.mark.parametrize("name", SMALL_CASES)
def test_small_cases_fp32_vs_reference_golden(name):
    g = load_golden(name)
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    taps = {}
    logits, cl = run_engine(m, a, sd, tt, te, [ep], "fp32", taps=taps)
    S = m["way"] * m["shot"]
    feats = taps["feats"].cpu()[0]
    E = a["embed"]
    assert maxdiff(feats[:S].reshape(-1, E), g["feats_s"]) < 5e-4
    assert maxdiff(feats[S:].reshape(-1, E), g["feats_q"]) < 5e-4
    assert maxdiff(taps["ctx_q"].cpu()[0], g["ctx_q"]) < 5e-4
    assert maxdiff(taps["protos"].cpu()[0], g["protos"]) < 5e-4
    assert maxdiff(taps["dists"].cpu()[0], g["dists"]) < 5e-4
    assert maxdiff(logits[0], g["logits"]) < 1e-3                 # north-star tolerance
    assert maxdiff(cl[0], g["class_logits"]) < 1e-3


@pytest.mark.parametrize("name", SMALL_CASES)
def test_small_cases_bf16_bounded(name):
    g = load_golden(name)
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    logits, cl = run_engine(m, a, sd, tt, te, [ep], "bf16")
    assert maxdiff(logits[0], g["logits"]) < bound(name, "bf16")
    ref = torch.from_numpy(g["logits"])
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 0.3                      # queries whose decision is not marginal
    assert torch.equal(logits[0].argmax(1)[clear], ref.argmax(1)[clear])


@pytest.mark.parametrize("name", SMALL_CASES)
def test_small_cases_fp16_bounded(name):
    """precision "fp16" (IEEE-half operands everywhere) on the small cases (ViT and, round 4, the RN tower): about 4-8 x closer to the
    reference than bf16."""
    g = load_golden(name)
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    logits, cl = run_engine(m, a, sd, tt, te, [ep], "fp16")
    assert maxdiff(logits[0], g["logits"]) < bound(name, "fp16")
    ref = torch.from_numpy(g["logits"])
    assert torch.equal(logits[0].argmax(1), ref.argmax(1))


@pytest.mark.parametrize("name", ["t_5w1s_T8", "t_5w3s_T16_mb_d2", "t197_5w1s_T2", "t257_5w1s_T2"])
def test_small_cases_fp16_strict_bounded(name):
    """precision "fp16_strict" on the tiny architectures (17 tokens per frame: every block GEMM falls back to split weights; 197 / 257 tokens: the
    product's split-QKV + per-frame-correction form; 14 x 14 patches): inside the fp16 mode's per-case bound, same argmax as the reference."""
    g = load_golden(name)
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    logits, _ = run_engine(m, a, sd, tt, te, [ep], "fp16_strict")
    assert maxdiff(logits[0], g["logits"]) < bound(name, "fp16"), maxdiff(logits[0], g["logits"])
    assert torch.equal(logits[0].argmax(1), torch.from_numpy(g["logits"]).argmax(1))


@pytest.mark.parametrize("name", OUTLIER_CASES)
def test_outlier_channel_statistics(name, monkeypatch):
    """Reference-generated goldens whose residual stream carries two channels at |x| ~ 100 and a non-zero row mean (trained-CLIP-like
    "massive activations"; synth.apply_outlier_channels): the regime where the LayerNorm-folded GEMM computes (x Wg - mean c) / std by
    cancelling large terms.  fp32 mode: the north-star 1e-3.  16-bit modes: bounded at 2 x their measured deviation, and the folded
    block must not be worse than the unfolded one (separate fp32 LayerNorm kernels) beyond rounding noise."""
    g = load_golden(name)
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    assert float(sd["backbone.ln_pre.bias"].abs().max()) > 50.0             # the fixture really carries the outlier channels
    taps = {}
    l32, _ = run_engine(m, a, sd, tt, te, [ep], "fp32", taps=taps)
    assert float(taps["block0"].abs().max()) > 60.0                          # ... and so does the stream on the device
    assert maxdiff(l32[0], g["logits"]) < NORTH_STAR_TOLERANCE
    lh, _ = run_engine(m, a, sd, tt, te, [ep], "fp16")
    assert maxdiff(lh[0], g["logits"]) < bound(name, "fp16")
    lb, _ = run_engine(m, a, sd, tt, te, [ep], "bf16")
    assert maxdiff(lb[0], g["logits"]) < bound(name, "bf16")
    monkeypatch.setenv("CFSAR_LN_FOLD", "0")
    lu, _ = run_engine(m, a, sd, tt, te, [ep], "bf16")
    assert maxdiff(lb[0], g["logits"]) < 2.0 * maxdiff(lu[0], g["logits"]) + 1e-3, (maxdiff(lb[0], g["logits"]), maxdiff(lu[0], g["logits"]))


def test_fp16_mode_steady_parity_statistic():
    """The golden cases hold ONE episode each (5 ... 25 logits): max |dlogits| of one episode moves by +-30 % with any change of a last
    bit.  The steadier regression guard: 8 fresh cfg2 episodes (40 logits) in the fp16 mode against the fp32 mode of the same engine
    (itself within 1e-5 of the reference on every golden) -- rms and max.  Measured (16 episodes, profiles/r04_parity_table.md): rms
    2.4e-4, max 7.4e-4; round 3's fp16 mode: 4.2e-4 / 1.18e-3."""
    g = load_golden("cfg2_B16_5w1s_T8")
    m = g["meta"]
    a, sd, tt, te, _ = case_inputs(m)
    eps = [{k: torch.from_numpy(v) for k, v in synth.make_episode(way=m["way"], shot=m["shot"], query_per_class=m["q"], frames=m["T"],
                                                                  res=a["res"], n_test_classes=m["n_test"], episode=2000 + e,
                                                                  seed=m["seed"]).items()} for e in range(8)]
    l32, _ = run_engine(m, a, sd, tt, te, eps, "fp32")
    l16, _ = run_engine(m, a, sd, tt, te, eps, "fp16")
    d = l16 - l32
    assert float(d.pow(2).mean().sqrt()) < 3.5e-4, float(d.pow(2).mean().sqrt())
    assert float(d.abs().max()) < LOGITS_TOLERANCE["fp16"], float(d.abs().max())     # the tail's regression bound (an episode's largest deviation passes 1e-3 in ~1 of 13-60)
    assert torch.equal(l16.argmax(2), l32.argmax(2))


def test_fp16_mode_refuses_what_it_cannot_represent():
    """A checkpoint whose BatchNorm-folded conv weights leave the fp16 range cannot run the RN tower's fp16 mode; the engine says so
    instead of producing infinities."""
    g = load_golden("rn_t_5w2s_T4")
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    sd = dict(sd)
    sd["backbone.layer1.0.bn2.weight"] = sd["backbone.layer1.0.bn2.weight"] * 1e7
    with pytest.raises(ValueError, match="fp16 range"):
        run_engine(m, a, sd, tt, te, [ep], "fp16")
    run_engine(m, a, sd, tt, te, [ep], "bf16")


@pytest.mark.parametrize("lowfreq", [0.0, 2.0])
def test_rn50_fp16_mode_steady_parity_statistic(lowfreq):
    """CLIP RN50 tower, precision "fp16" (IEEE-half activations and weights, fp32 accumulation, one rounding per stored tensor) against the fp32
    validation mode on 8 fresh 8-frame episodes (200 logits).  The tower's features carry a relative error of 8e-4 (bf16: 6e-3; hardware =
    the CPU model tools/numerics_lab_rn.py to three digits, tools/rn_fp16_probe.py); what that is in the logits depends on the episodes' contrast:
      lowfreq 0 (white-noise frames, the bench's synthetic episodes; logits spread 0.1): 1.7e-4 max (bf16: 1.2e-3) -- inside the north-star 1e-3;
      lowfreq 2 (smooth frames as in the RN goldens; spread 1.5): rms 8e-4, max 2.5e-3 (bf16: 6e-3 / 1.7e-2) -- NOT inside 1e-3 (the 2-frame golden is: 3.7e-4);
    three comparable sources (block outputs, inner activations, weights: 4-5e-4 each), so nothing short of more than 11 bits per stored
    activation closes it: precision "fp32" is the RN50 mode for a 1e-3 contract on such inputs.  Bounds: 2 x measured."""
    g = load_golden("rn50_5w1s_T2")
    m = dict(g["meta"])
    m["T"] = 8
    m["lowfreq"] = lowfreq
    eps = []
    for e in range(8):
        a, sd, tt, te, ep = case_inputs(m, episode=100 + e)
        eps.append(ep)
    l32, _ = run_engine(m, a, sd, tt, te, eps, "fp32")
    l16, _ = run_engine(m, a, sd, tt, te, eps, "fp16")
    lb, _ = run_engine(m, a, sd, tt, te, eps, "bf16")
    d, db = l16 - l32, lb - l32
    rms, rms_b = float(d.pow(2).mean().sqrt()), float(db.pow(2).mean().sqrt())
    print("RN50 lowfreq %.0f, 8 episodes: fp16 vs fp32 rms %.2e max %.2e; bf16 rms %.2e max %.2e; spread %.2f" % (
        lowfreq, rms, float(d.abs().max()), rms_b, float(db.abs().max()), float(l32.max() - l32.min())))
    assert rms < rms_b / 4.0, (rms, rms_b)                          # three more mantissa bits: 8 x in the features
    if lowfreq == 0.0:
        assert float(d.abs().max()) < 0.5 * NORTH_STAR_TOLERANCE, float(d.abs().max())
    else:
        assert rms < 1.6e-3 and float(d.abs().max()) < 5e-3, (rms, float(d.abs().max()))


def test_batched_episodes_match_single(tmp_path):
    g = load_golden("t_5w1s_T8")
    m = g["meta"]
    a, sd, tt, te, ep0 = case_inputs(m, episode=0)
    eps = [ep0] + [case_inputs(m, episode=e)[4] for e in (1, 2)]
    lb, clb = run_engine(m, a, sd, tt, te, eps, "fp32")
    for i, e in enumerate(eps):
        l1, c1 = run_engine(m, a, sd, tt, te, [e], "fp32")
        assert maxdiff(lb[i], l1[0]) < 1e-5
        assert maxdiff(clb[i], c1[0]) < 1e-5
    assert maxdiff(lb[0], g["logits"]) < 1e-3
    # oracle on a fresh episode (not in any fixture)
    with torch.no_grad():
        o = orc.head_forward(eps[2], sd, tt, te, a, frames=m["T"])
    assert maxdiff(lb[2], o["logits"]) < 1e-3


def test_cfg2_full_size_fp32_and_bf16():
    """BASELINE config 2 (5-way 1-shot, 8 frames, ViT-B/16): fp32 mode within 1e-3 of the reference's logits."""
    g = load_golden("cfg2_B16_5w1s_T8")
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    taps = {}
    logits, cl = run_engine(m, a, sd, tt, te, [ep], "fp32", taps=taps)
    feats = taps["feats"].cpu()[0].reshape(-1, a["embed"])
    assert maxdiff(feats[:40], g["feats_s"]) < 2e-3
    assert maxdiff(logits[0], g["logits"]) < 1e-3
    assert maxdiff(cl[0], g["class_logits"]) < 1e-3
    lb, _ = run_engine(m, a, sd, tt, te, [ep], "bf16")
    assert maxdiff(lb[0], g["logits"]) < LOGITS_TOLERANCE["bf16"]
    lh, ch = run_engine(m, a, sd, tt, te, [ep], "fp16")
    assert maxdiff(lh[0], g["logits"]) < NORTH_STAR_TOLERANCE                # single golden, deterministic: measured 6.1e-4 ... 7.6e-4 over the rounds' builds (the 1.5e-3 tail bound is for the multi-episode max only)
    assert torch.equal(lh[0].argmax(1), torch.from_numpy(g["logits"]).argmax(1))


def _harness_batch(cfgname):
    """episodes per model call the product harness (utils/batching.py, datasets/base/builder.py) and bench.py pick: cfg2 36, cfg3 12, cfg4 11"""
    import bench
    return bench.default_episodes_per_step(cfgname)


def test_timed_configuration_b36_equals_b1_and_golden():
    """The configuration bench.py times (cfg2 at the harness's batch: 36 episodes = 2 880 frames per step, M = 567 360 rows -- the first regime
    where activation buffers pass 2^31 bytes: u 3.49 GB, qkv 2.6 GB; VERDICT r5 item 5 / ADVICE r5): the FIRST, a MIDDLE and the LAST episode
    of the batch give the logits the same episode gives alone, in bf16, fp16 and fp16_strict (the single-episode call takes the two-stream
    small-batch path and 192-row tiles, the batch the one-stream path and 256-row tiles: row results do not depend on the tile a row lands
    in); episode 0 is the reference golden; the fp32 mode at the same batch meets 1e-3 against it."""
    B = _harness_batch("cfg2")
    assert B == 36
    g = load_golden("cfg2_B16_5w1s_T8")
    m = g["meta"]
    a, sd, tt, te, ep0 = case_inputs(m)
    eps = [ep0] + [case_inputs(m, episode=m["episode"] + e)[4] for e in range(1, B)]
    for prec, tol in (("bf16", LOGITS_TOLERANCE["bf16"]), ("fp16", NORTH_STAR_TOLERANCE), ("fp16_strict", NORTH_STAR_TOLERANCE)):
        lb, cb = run_engine(m, a, sd, tt, te, eps, prec)
        assert maxdiff(lb[0], g["logits"]) < tol, prec
        for i in (0, B // 2 - 1, B - 1):
            l1, c1 = run_engine(m, a, sd, tt, te, [eps[i]], prec)
            # the ViT tower is bit-identical at any batch size; the fp32 tail picks its GEMM kernel by row count (skinny FMA kernel for
            # one episode, MFMA kernel for 36): a different fp32 summation order = a few ulps of a logit of magnitude ~10
            assert maxdiff(lb[i], l1[0]) <= 4e-6, (prec, i, maxdiff(lb[i], l1[0]))
            assert maxdiff(cb[i], c1[0]) <= 4e-6, (prec, i)
    lf, cf = run_engine(m, a, sd, tt, te, eps, "fp32")
    assert maxdiff(lf[0], g["logits"]) < 1e-3
    assert maxdiff(cf[0], g["class_logits"]) < 1e-3


@pytest.mark.parametrize("cfgname,case", [("cfg3", "cfg3_B16_5w5s_T8_mb"), ("cfg4", "cfg4_L14_5w1s_T16")])
def test_harness_batch_cfg3_cfg4_last_episode_equals_b1(cfgname, case):
    """cfg3 at 12 and cfg4 at 11 episodes per call (cfg4: M x 4 096 x 2 B = 86 % of the 32-bit byte-offset range of the hidden matrix): episode 0
    against the reference golden, the LAST episode of the batch equal to the same episode alone (bf16 and fp16)."""
    B = _harness_batch(cfgname)
    assert B == {"cfg3": 12, "cfg4": 11}[cfgname]
    g = load_golden(case)
    m = g["meta"]
    a, sd, tt, te, ep0 = case_inputs(m)
    eps = [ep0] + [case_inputs(m, episode=m["episode"] + e)[4] for e in range(1, B)]
    for prec, tol in (("bf16", bound(case, "bf16")), ("fp16", NORTH_STAR_TOLERANCE)):
        lb, cb = run_engine(m, a, sd, tt, te, eps, prec)
        assert maxdiff(lb[0], g["logits"]) < tol, prec
        l1, c1 = run_engine(m, a, sd, tt, te, [eps[B - 1]], prec)
        assert maxdiff(lb[B - 1], l1[0]) <= 4e-6, (prec, maxdiff(lb[B - 1], l1[0]))
        assert maxdiff(cb[B - 1], c1[0]) <= 4e-6, prec


def test_rn50_harness_batch_last_episode():
    """RN50 tower at the harness's 32 episodes per call (2 560 frames: the stem output is 4.11e9 bytes, 4 % below 2^32; ADVICE r5: nothing compared
    the LAST episodes of such a batch with anything).  fp32 and fp16: the first and the last episode of the batch equal the same episode alone
    (measured 2e-6 / 2e-6: the towers' bits do not depend on the batch).  bf16: the tower picks its conv / GEMM kernels by tile count, so its bits DO
    depend on the batch (1.0-1.5e-2 between one episode alone and the same episode in a batch of 2 ... 32 -- the size of the mode's own deviation
    from the reference, rms 7e-3); what is asserted there is that the last episode of the batch is as close to the fp32 mode as the first
    (no corruption at the high addresses): both inside the tower's bf16 bound."""
    from clip_fsar_amd import LOGITS_TOLERANCE_RN50
    B = _harness_batch("rn50")
    assert B == 32
    g = load_golden("rn50_5w1s_T2")
    m = dict(g["meta"])
    m["T"] = 8
    eps = [case_inputs(m, episode=300 + e)[4] for e in range(B)]
    a, sd, tt, te, _ = case_inputs(m)
    l32, _ = run_engine(m, a, sd, tt, te, eps, "fp32")
    for prec in ("fp32", "fp16"):
        lb, cb = (l32, None) if prec == "fp32" else run_engine(m, a, sd, tt, te, eps, prec)
        for i in (0, B - 1):
            l1, _ = run_engine(m, a, sd, tt, te, [eps[i]], prec)
            assert maxdiff(lb[i], l1[0]) <= 4e-6, (prec, i, maxdiff(lb[i], l1[0]))
        if prec == "fp16":
            assert maxdiff(lb, l32) < LOGITS_TOLERANCE_RN50["fp16"], maxdiff(lb, l32)
    lb, _ = run_engine(m, a, sd, tt, te, eps, "bf16")
    per_ep = (lb - l32).abs().reshape(B, -1).max(1).values
    assert float(per_ep.max()) < LOGITS_TOLERANCE_RN50["bf16"], per_ep
    assert float(per_ep[B - 1]) < 2.0 * float(per_ep.median()) + 5e-3, per_ep          # the last episode is an ordinary one


def test_cfg3_four_episodes_per_step():
    """BASELINE config 3 (5-way 5-shot, MERGE_BEFORE) batched 4 episodes per step: episode 0 against the reference golden."""
    g = load_golden("cfg3_B16_5w5s_T8_mb")
    m = g["meta"]
    a, sd, tt, te, ep0 = case_inputs(m)
    eps = [ep0] + [case_inputs(m, episode=m["episode"] + e)[4] for e in range(1, 4)]
    lf, cf = run_engine(m, a, sd, tt, te, eps, "fp32")
    assert maxdiff(lf[0], g["logits"]) < 1e-3
    assert maxdiff(cf[0], g["class_logits"]) < 1e-3
    lb, _ = run_engine(m, a, sd, tt, te, eps, "bf16")
    assert maxdiff(lb[0], g["logits"]) < LOGITS_TOLERANCE["bf16"]
    lh, _ = run_engine(m, a, sd, tt, te, eps, "fp16")
    assert maxdiff(lh[0], g["logits"]) < NORTH_STAR_TOLERANCE                          # single golden: 3.7e-4 ... 5.1e-4 over builds (r3's fp16 mode: 1.06e-3)
    l1, _ = run_engine(m, a, sd, tt, te, [eps[2]], "bf16")
    assert maxdiff(lb[2], l1[0]) <= 4e-6


@pytest.mark.parametrize("name,tol_feat", [("cfg3_B16_5w5s_T8_mb", 2e-3), ("cfg4_L14_5w1s_T16", 4e-3),
                                           ("rn50_5w1s_T2", 2e-3)])
def test_cfg3_cfg4_full_size(name, tol_feat):
    """BASELINE config 3 (5-way 5-shot, MERGE_BEFORE as in the shipped 5-shot yaml) and config 4 (ViT-L/14, 16 frames:
    extension A16, oracle = the reference's own classes composed by the harness)."""
    g = load_golden(name)
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    taps = {}
    logits, cl = run_engine(m, a, sd, tt, te, [ep], "fp32", taps=taps)
    S = m["way"] * m["shot"]
    feats = taps["feats"].cpu()[0].reshape(-1, a["embed"])
    assert maxdiff(feats[:S * m["T"]], g["feats_s"]) < tol_feat
    assert maxdiff(taps["protos"].cpu()[0], g["protos"]) < tol_feat
    assert maxdiff(logits[0], g["logits"]) < 1e-3
    assert maxdiff(cl[0], g["class_logits"]) < 1e-3
    lb, _ = run_engine(m, a, sd, tt, te, [ep], "bf16")
    assert maxdiff(lb[0], g["logits"]) < bound(name, "bf16")
    if True:                                                       # every tower has its fp16 mode (RN50: round 4)
        lh, _ = run_engine(m, a, sd, tt, te, [ep], "fp16")
        # The fp16 mode (single-rounding residual add, two-word stream, per-frame low-word correction of all four GEMMs' weights) on the single
        # goldens: 5.8e-4 (cfg2), 4.7e-4 (cfg3), 3.8e-4 (cfg4) -- deterministic values, asserted at the north-star 1e-3 (the RN50 tower: its own
        # bound); what the mode guarantees over MANY episodes is a statistic: test_modes_against_multi_episode_reference_goldens.
        assert maxdiff(lh[0], g["logits"]) < (NORTH_STAR_TOLERANCE if "rn50" not in name else LOGITS_TOLERANCE["fp16"]), name
    print("%s: fp32 |dlogits| %.2e, bf16 |dlogits| %.3f" % (name, maxdiff(logits[0], g["logits"]), maxdiff(lb[0], g["logits"])))


def test_registry_builder_dropin_surface():
    """build_model(cfg) -> BaseVideoModel(Identity, CNN_OTAM_CLIPFSAR); model(task_dict) -> logits like the reference
    harness calls it (runs/test_net_few_shot.py:59-62,109)."""
    from types import SimpleNamespace as NS
    import clip_fsar_amd.models.base  # noqa: F401  (registers)
    from clip_fsar_amd.models.base.builder import build_model
    g = load_golden("t_5w1s_T8")
    m = g["meta"]
    cfg = NS(VIDEO=NS(HEAD=NS(NAME="CNN_OTAM_CLIPFSAR", BACKBONE_NAME=m["arch"], PRECISION="fp32"),
                      BACKBONE=NS(META_ARCH="Identity")),
             TRAIN=NS(CLASS_NAME=["c%d" % i for i in range(m["n_train"])], WAY=m["way"]),
             TEST=NS(CLASS_NAME=["t%d" % i for i in range(m["n_test"])]), DATA=NS(NUM_INPUT_FRAMES=m["T"]),
             MODEL=NS(NAME="BaseVideoModel", EMA=NS(ENABLE=False)), BN=NS(FREEZE=False), NUM_GPUS=1, NUM_SHARDS=1,
             RANDOM_SEED=m["seed"])
    model, ema = build_model(cfg)
    assert ema is None
    model.eval()
    _, _, _, _, ep = case_inputs(m)
    task = {k: v.cuda(non_blocking=True) for k, v in ep.items()}
    with torch.no_grad():
        out = model(task)
    assert out["logits"].shape == (m["way"] * m["q"], m["way"])
    assert maxdiff(out["logits"].cpu(), g["logits"]) < 1e-3
    assert maxdiff(out["class_logits"].cpu(), g["class_logits"]) < 1e-3
    loss = model.head.loss(task, out)
    assert torch.isfinite(loss)


@pytest.mark.parametrize("episodes", [1, 4])
def test_forward_is_bit_stable_run_to_run(episodes):
    """Same engine, same inputs, eight forwards: identical bits.  One episode takes the two-stream small-batch path, four the
    one-stream path.  (Round 2: a rare stale-lanes fault in one GEMM build first showed up as 1e-4 run-to-run differences here;
    tools/determinism_probe.py covers the other towers.)"""
    from clip_fsar_amd.engine import ClipFsarEngine
    g = load_golden("cfg2_B16_5w1s_T8")
    m = g["meta"]
    a, sd, tt, te, ep0 = case_inputs(m)
    eps = [ep0] + [case_inputs(m, episode=m["episode"] + e)[4] for e in range(1, episodes)]
    eng = ClipFsarEngine(a, sd, tt, te, precision="bf16", device="cuda")
    st = lambda k: torch.stack([e[k] for e in eps]).cuda()
    args = (st("support_set"), st("target_set"), st("support_labels"), st("real_support_labels"))
    ref = None
    for it in range(8):
        lo, cl = eng.forward(*args, way=m["way"], T=m["T"])
        torch.cuda.synchronize()
        if ref is None:
            ref = (lo.clone(), cl.clone())
        else:
            assert torch.equal(lo, ref[0]) and torch.equal(cl, ref[1]), (it, float((lo - ref[0]).abs().max()))


@pytest.mark.parametrize("name", ["mc_cfg2_B16_5w1s_T8", "hc_cfg2_B16_5w1s_T8", "hc_cfg3_B16_5w5s_T8_mb", "hc_cfg4_L14_5w1s_T16", "mc_cfg4_L14_5w1s_T16",
                                  "hc_rn50_5w1s_T8", "oc_cfg2_B16_5w1s_T8"])
def test_modes_against_multi_episode_reference_goldens(name):
    """What each numerics mode guarantees, asserted on what it actually controls: 13 episodes = 65 logit rows per
    full-size configuration, produced by the REFERENCE itself (oracle/make_golden.py --multi), at the generator's standard contrast (`mc_`,
    logits spread ~1) and at high contrast (`hc_`, spread 3-4.5).  fp32: a hard bound on every logit of every episode.  fp16: a statistic
    (rms and 99th percentile over all rows) plus the regression bound of the tail, no argmax flip.  bf16: its regression bounds (a near-tie may flip)."""
    from clip_fsar_amd import LOGITS_STATISTIC, LOGITS_STATISTIC_RN50, LOGITS_TOLERANCE_RN50
    stat, tol = (LOGITS_STATISTIC_RN50, LOGITS_TOLERANCE_RN50) if "rn50" in name else (LOGITS_STATISTIC, LOGITS_TOLERANCE)    # the RN50 tower has its own (looser) bounds
    if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multi_%s.npz" % name)):
        pytest.skip("fixture not generated")
    st = multi_case_stats(name, "fp32")
    assert st["worst_episode_max"] < 1e-4 and st["argmax_equal"] == st["rows"], st        # (the north star is 1e-3; measured 4e-6)
    st = multi_case_stats(name, "fp16")
    print(name, "fp16", st)
    assert st["rms"] <= stat["fp16"]["rms"] and st["p99"] <= stat["fp16"]["p99"], st
    assert st["max"] < tol["fp16"] and st["argmax_equal"] == st["rows"], st
    st = multi_case_stats(name, "bf16")
    print(name, "bf16", st)
    assert st["rms"] <= stat["bf16"]["rms"] and st["p99"] <= stat["bf16"]["p99"], st
    assert st["max"] < tol["bf16"] and st["argmax_equal"] >= st["rows"] - 2, st      # (a near-tie may flip: 1 of 325 rows measured)


STRICT_SETS = ["mc_cfg2_B16_5w1s_T8", "hc_cfg2_B16_5w1s_T8", "hc_cfg3_B16_5w5s_T8_mb", "hc_cfg4_L14_5w1s_T16", "mc_cfg4_L14_5w1s_T16", "oc_cfg2_B16_5w1s_T8"]


@pytest.mark.parametrize("name", STRICT_SETS)
def test_strict_mode_every_reference_golden_row_inside_1e3(name):
    """precision "fp16_strict" (VERDICT r5 item 1): the 16-bit mode whose contract is a BOUND on the reference's goldens -- every logit of every one
    of the 13 reference episodes of each of the six full-size ViT sets (455 rows: standard and high contrast, 5-shot MERGE_BEFORE, ViT-L/14,
    trained-CLIP-like outlier channels) within the north-star 1e-3, no episode over, no argmax flip; and tighter than the fp16 mode's statistic."""
    from clip_fsar_amd import LOGITS_STATISTIC
    if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "multi_%s.npz" % name)):
        pytest.skip("fixture not generated")
    st = multi_case_stats(name, "fp16_strict")
    print(name, "fp16_strict", st)
    assert st["max"] < NORTH_STAR_TOLERANCE and st["episodes_over_1e-3"] == 0, st
    assert st["argmax_equal"] == st["rows"], st
    assert st["rms"] <= LOGITS_STATISTIC["fp16_strict"]["rms"] and st["p99"] <= LOGITS_STATISTIC["fp16_strict"]["p99"], st


@pytest.mark.parametrize("name", ["cfg2_B16_5w1s_T8", "cfg3_B16_5w5s_T8_mb", "cfg4_L14_5w1s_T16"])
def test_strict_mode_single_goldens(name):
    g = load_golden(name)
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    ls, _ = run_engine(m, a, sd, tt, te, [ep], "fp16_strict")
    assert maxdiff(ls[0], g["logits"]) < NORTH_STAR_TOLERANCE, maxdiff(ls[0], g["logits"])
    assert torch.equal(ls[0].argmax(1), torch.from_numpy(g["logits"]).argmax(1))


def test_strict_mode_is_the_fp16_mode_plus_its_front_end():
    """fp16_strict = the fp16 mode with the patch embedding in three fp16 passes ([hi | lo | hi] pixels x [W_hi | W_hi | W_lo]) into fp32 tokens and
    class token + pos + ln_pre written straight into the two-word stream: ln_pre's tap is within 2e-6 of the fp32 mode's (the fp16 mode's:
    one fp16 rounding, 5e-4), the RN50 tower refuses the mode by name."""
    from clip_fsar_amd.engine import ClipFsarEngine, HipViT
    g = load_golden("t197_5w1s_T2")
    m = g["meta"]
    a, hsd, tt, te, ep = case_inputs(m)
    vsd = {k[len("backbone."):]: v for k, v in hsd.items() if k.startswith("backbone.")}
    frames = ep["support_set"][:4].cuda()
    taps32, taps16, tapss = {}, {}, {}
    HipViT(a, vsd, precision="fp32").forward(frames, taps=taps32)
    HipViT(a, vsd, precision="fp16").forward(frames, taps=taps16)
    vs = HipViT(a, vsd, precision="fp16_strict")
    assert vs.strict and vs.precision == "fp16" and not vs.fused_patch
    vs.forward(frames, taps=tapss)
    scale = float(taps32["ln_pre"].abs().max())
    assert maxdiff(tapss["ln_pre"].cpu(), taps32["ln_pre"].cpu()) < 4e-6 * max(1.0, scale)
    assert maxdiff(taps16["ln_pre"].float().cpu(), taps32["ln_pre"].cpu()) > 20 * maxdiff(tapss["ln_pre"].cpu(), taps32["ln_pre"].cpu())
    gr = load_golden("rn_t_5w2s_T4")
    ar, sdr, ttr, ter, _ = case_inputs(gr["meta"])
    with pytest.raises(ValueError, match="fp16_strict"):
        ClipFsarEngine(ar, sdr, ttr, ter, precision="fp16_strict", device="cuda")
    # the mode's defaults: split QKV weights, the per-frame correction on the other three GEMMs, its raw-stream form kept for c_fc
    gc = load_golden("cfg2_B16_5w1s_T8")
    ac, hc_sd, _, _, _ = case_inputs(gc["meta"])
    vc = HipViT(ac, {k[len("backbone."):]: v for k, v in hc_sd.items() if k.startswith("backbone.")}, precision="fp16_strict")
    assert vc.split == {"qkv"} and vc.mcorr == {"out", "fc", "pr"} and vc.rawmeans and vc.strict_front and not vc.o_pair


def test_strict_option_two_word_attention_output():
    """Developer option strict_o_pair (built and measured in round 6, off by default: 11 % of the step for -4 ... +7 % in rms on the reference sets,
    profiles/r06_strict_eval.md): cfsar_vit_attention_pair + cfsar_gemm_residual_wide(wsplit = 2) end to end, inside the mode's bound."""
    g = load_golden("cfg2_B16_5w1s_T8")
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    lp, _ = run_engine(m, a, sd, tt, te, [ep], "fp16_strict", vit_options={"strict_o_pair": True})
    ls, _ = run_engine(m, a, sd, tt, te, [ep], "fp16_strict", vit_options={})
    assert maxdiff(lp[0], g["logits"]) < NORTH_STAR_TOLERANCE and maxdiff(ls[0], g["logits"]) < NORTH_STAR_TOLERANCE
    assert not torch.equal(lp, ls)                                # the option is live


def test_fp16_mode_b16_equals_b1():
    """An episode's fp16-mode logits do not depend on the batch it is served in: the per-frame correction's k slot is chosen by the FRAME's
    parity (not by the tile a row falls into) and c_fc's per-frame means are summed per 32-row group (not per wave tile), so the 192- and
    256-row tile forms the launcher picks by batch size give the same tower bits; the fp32 tail differs by its GEMM's summation order only."""
    g = load_golden("cfg2_B16_5w1s_T8")
    m = g["meta"]
    a, sd, tt, te, ep0 = case_inputs(m)
    eps = [ep0] + [case_inputs(m, episode=m["episode"] + e)[4] for e in range(1, 16)]
    l16, c16 = run_engine(m, a, sd, tt, te, eps, "fp16")
    assert maxdiff(l16[0], g["logits"]) < NORTH_STAR_TOLERANCE
    for i in (0, 7, 15):
        l1, c1 = run_engine(m, a, sd, tt, te, [eps[i]], "fp16")
        assert maxdiff(l16[i], l1[0]) <= 4e-6, (i, maxdiff(l16[i], l1[0]))
        assert maxdiff(c16[i], c1[0]) <= 4e-6, i

def test_developer_options_are_constructor_arguments():
    """the engine reads four environment variables; every other ablation is a constructor option.  An unknown option is
    refused; fp16_wide=False (round 3's packed-fp16 residual add) drops the per-frame correction with a warning instead of silently promoting
    it to split weights and still runs inside round 3's bound; the engine source holds no other environment read."""
    import re
    import warnings
    from clip_fsar_amd.engine import ClipFsarEngine
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "clip-fsar_amd", "engine.py")).read()
    assert sorted(set(re.findall(r'environ\.get\("(CFSAR_[A-Z0-9_]+)"', src))) == ["CFSAR_FP16_MCORR", "CFSAR_FP16_SPLIT", "CFSAR_FULL_LAST_BLOCK", "CFSAR_LN_FOLD"]
    g = load_golden("cfg2_B16_5w1s_T8")
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    with pytest.raises(ValueError, match="unknown option"):
        ClipFsarEngine(a, sd, tt, te, precision="fp16", device="cuda", vit_options={"no_such_switch": True})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        lo, _ = run_engine(m, a, sd, tt, te, [ep], "fp16", vit_options={"fp16_wide": False})
    assert any("fp16_wide=False drops" in str(x.message) for x in w), [str(x.message) for x in w]
    assert maxdiff(lo[0], g["logits"]) < 2.5e-3                   # round 3's one-word, packed-add mode: 5e-4 ... 1.8e-3 on the goldens


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_fused_patch_embedding_leaves_the_logits_bit_identical(precision):
    """the 16-bit modes of the ViT-B/16 tower gather the patch-embed GEMM's rows from the frames inside the GEMM (cfsar_patch_embed,
    one launch instead of three, no patch matrix); the developer option fused_patch=False keeps the im2col form.  Same roundings, same MFMA
    order: the episode's logits are the same bits."""
    g = load_golden("cfg2_B16_5w1s_T8")
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    l_f, c_f = run_engine(m, a, sd, tt, te, [ep], precision)
    l_u, c_u = run_engine(m, a, sd, tt, te, [ep], precision, vit_options={"fused_patch": False})
    assert torch.equal(l_f, l_u) and torch.equal(c_f, c_u), (maxdiff(l_f, l_u), maxdiff(c_f, c_u))


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_fused_patch_embedding_14x14_cfg4(precision):
    """ViT-L/14 (cfg4): since round 6 the patch embedding is ONE launch there too (cfsar_patch_embed, 14 x 14 patches in padded-row columns).  Its K axis is
    walked in another order than the im2col form's (704 padded slots against 640), so the two forms are two realisations of the mode's rounding noise,
    not the same bits: both inside the mode's bound of the reference golden."""
    g = load_golden("cfg4_L14_5w1s_T16")
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    tol = bound("cfg4_L14_5w1s_T16", "bf16") if precision == "bf16" else NORTH_STAR_TOLERANCE
    l_f, _ = run_engine(m, a, sd, tt, te, [ep], precision, vit_options={})
    l_u, _ = run_engine(m, a, sd, tt, te, [ep], precision, vit_options={"fused_patch": False})
    assert maxdiff(l_f[0], g["logits"]) < tol and maxdiff(l_u[0], g["logits"]) < tol, (maxdiff(l_f[0], g["logits"]), maxdiff(l_u[0], g["logits"]))
    assert not torch.equal(l_f, l_u)


def test_fp16_raw_stream_correction_switch(monkeypatch):
    """The fp16 mode's LN-folded GEMMs take their per-frame correction in the raw-stream form by default (no pass over x: the stream's per-frame
    mean follows its updates through two [frames, K] x [K, D] GEMMs per block); the developer option fp16_rawmeans=False (tests/_cases.py maps CFSAR_FP16_RAWMEANS=0 to it) restores the normalised-mean form with its
    two frame_col_means passes per block.  Both inside the north-star tolerance on the cfg2 golden; 64 fresh episodes per configuration give the
    same rms for both (profiles/r04_parity_table.md)."""
    g = load_golden("cfg2_B16_5w1s_T8")
    m = g["meta"]
    a, sd, tt, te, ep = case_inputs(m)
    l_raw, _ = run_engine(m, a, sd, tt, te, [ep], "fp16")
    monkeypatch.setenv("CFSAR_FP16_RAWMEANS", "0")
    l_norm, _ = run_engine(m, a, sd, tt, te, [ep], "fp16")
    assert maxdiff(l_raw[0], g["logits"]) < NORTH_STAR_TOLERANCE and maxdiff(l_norm[0], g["logits"]) < NORTH_STAR_TOLERANCE
    assert not torch.equal(l_raw, l_norm)                       # the switch is live
