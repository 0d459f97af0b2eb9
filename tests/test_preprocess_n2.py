"""N2: test-time frame transform.  CPU: oracle == the reference's KineticsResizedCropFewshot pipeline
(golden from the real reference).  GPU: the HIP kernel == the same golden (bilinear weights in fp32: tolerance 2e-5)."""
import json
import os

import numpy as np
import pytest
import torch

import clip_fsar_amd.synth as synth
import clipfsar_oracle as orc
from clip_fsar_amd.preprocess import crop_window

GOLD = os.path.join(os.path.dirname(__file__), "golden", "preprocess_cases.npz")


def _cases():
    z = np.load(GOLD)
    meta = json.loads(str(z["meta"]))
    for name, c in meta["cases"].items():
        v = synth.pseudo_normal(c["T"] * c["H"] * c["W"] * 3, "u8video/" + name, meta["seed"])
        vid = np.clip(v * 60.0 + 128.0, 0, 255).astype(np.uint8).reshape(c["T"], c["H"], c["W"], 3)
        scale = c["scale"] if isinstance(c["scale"], list) else [c["scale"], c["scale"]]
        yield name, c, torch.from_numpy(vid), scale, torch.from_numpy(z[name])


def test_oracle_preprocess_matches_reference():
    for name, c, vid, scale, ref in _cases():
        y0, x0 = crop_window(scale, c["crop"], c["nsc"], c["idx"])
        out = orc.preprocess_frames(vid, scale, c["crop"], y0, x0, synth.CLIP_MEAN, synth.CLIP_STD)
        assert out.shape == ref.shape
        assert float((out - ref).abs().max()) < 1e-6, name


def test_crop_window_rules():
    assert crop_window((256, 256), 224) == (16, 16)
    assert crop_window((72, 96), 64, 3, 0) == (4, 0) and crop_window((72, 96), 64, 3, 2) == (4, 32)
    # length = short_side_range[0] = the HEIGHT entry: with (96, 72) only the "height == length" rule can fire (:697-716)
    assert crop_window((96, 72), 64, 3, 0) == (16, 0) and crop_window((96, 72), 64, 3, 2) == (16, 8)


@pytest.mark.gpu
def test_hip_preprocess_matches_reference():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from clip_fsar_amd.preprocess import preprocess_video
    for name, c, vid, scale, ref in _cases():
        ts = c["scale"] if isinstance(c["scale"], list) else int(c["scale"])
        out = preprocess_video(vid.cuda(), ts, c["crop"], synth.CLIP_MEAN, synth.CLIP_STD, c["nsc"], c["idx"]).cpu()
        assert out.shape == ref.shape
        assert float((out - ref).abs().max()) < 2e-5, (name, float((out - ref).abs().max()))
