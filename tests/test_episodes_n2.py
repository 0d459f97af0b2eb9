"""N2: the episode pipeline in front of the hot path against goldens from the REAL reference dataset class
(datasets/base/ssv2_few_shot.py + base_dataset.py run with a stubbed decoder: oracle/make_golden_n2.py).

CPU: split-list parsing (both shipped formats), frame-index sampling (207 grid rows, both modes), and whole episodes -- classes,
labels, the decoded (path, frame indices) in decode order, the shuffles -- bit for bit under the same ``random`` seed; the frame
tensors through the host transform (<= 1e-6).  GPU: the same tensors through the fused HIP transform (<= 2e-5)."""
import json
import os
import random

import numpy as np
import pytest
import torch

import make_golden_n2 as g2                                  # oracle/: synthetic "videos" + cfg builder shared with the generator
from clip_fsar_amd.datasets.base import ssv2_few_shot as n2
from clip_fsar_amd.datasets.base.builder import DATASET_REGISTRY, build_loader

GOLD = os.path.join(os.path.dirname(__file__), "golden")
J = json.load(open(os.path.join(GOLD, "episodes_n2.json")))
Z = np.load(os.path.join(GOLD, "episodes_n2_frames.npz"))


class StubReader:
    def __init__(self, path):
        self.path = path
        self.length, self.fps, _, _ = g2.video_meta(path)

    def __len__(self):
        return self.length

    def get_avg_fps(self):
        return self.fps

    def get_batch(self, idx):
        return torch.from_numpy(g2.video_frames(self.path, idx))


def _dataset(name, tmp_path, use_gpu=False, **test_extra):
    case = J["cases"][name]
    anno = tmp_path / name
    anno.mkdir(exist_ok=True)
    (anno / "test_few_shot.txt").write_text("\n".join(case["split_lines"]) + "\n")
    cfg = g2.make_cfg(case["cfg"], str(anno), str(anno))
    cfg.AUGMENTATION.USE_GPU = use_gpu
    for k, v in test_extra.items():
        setattr(cfg.TEST, k, v)
    ds = n2.Ssv2_few_shot(cfg, "test", decoder=StubReader)
    ds.decode_log = []
    return ds, case


def test_registered_under_the_reference_name():
    assert DATASET_REGISTRY.get("Ssv2_few_shot") is n2.Ssv2_few_shot


def test_split_list_parsing_matches_reference(tmp_path):
    for name, case in J["cases"].items():
        sp = n2.Split_few_shot(case["split_lines"], "test", dataset=case["cfg"]["dataset_few"])
        assert sp.videos == case["parsed"]["videos"], name
        assert sp.gt_a_list == case["parsed"]["class_ids"], name
        assert sp.get_unique_classes() == case["parsed"]["unique_classes"], name          # order feeds random.sample
        assert len(sp) == len(case["split_lines"])
    # the formats of the lists the reference ships (head lines + totals recorded from configs/projects/CLIPFSAR/*/test_few_shot.txt)
    for d, rec in J["shipped_lists"].items():
        vid, cls = n2.parse_split_line(rec["head"][0], "test", rec["dataset_few"])
        assert [vid, cls] == rec["first_parsed"], d
    with pytest.raises(ValueError):
        n2.parse_split_line("val3//videos/x.avi", "test", "Kinetics_few_shot")


def test_interval_based_sampling_matches_reference(tmp_path):
    rows = J["sampling_grid"]
    assert len(rows) > 150 and {r["mode"] for r in rows} == {"linspace", "segments"}
    for r in rows:
        cfg = g2.NS(DATA=g2.NS(SAMPLING_RATE=r["rate"], TARGET_FPS=r["target_fps"]))
        random.seed(r["py_seed"])
        got = n2.interval_based_sampling(cfg, "test", r["length"], r["fps"], 0, 1, r["frames"], r["rate"])
        assert got == r["index"], r
    cfg = g2.NS(DATA=g2.NS(SAMPLING_RATE=50, TARGET_FPS=12))
    with pytest.raises(ValueError, match="shorter"):
        n2.interval_based_sampling(cfg, "test", 5, 12.0, 0, 1, 8, 50)


@pytest.mark.parametrize("name", sorted(J["cases"]))
def test_episodes_match_reference(name, tmp_path):
    ds, case = _dataset(name, tmp_path)
    assert len(ds) == case["len"]
    for ep in case["episodes"]:
        random.seed(ep["py_seed"])
        del ds.decode_log[:]
        d = ds[ep["episode"]]
        for k in ("support_labels", "target_labels", "real_support_labels", "real_target_labels", "batch_class_list"):
            assert d[k].dtype == torch.float32 and d[k].tolist() == ep[k], (name, ep["episode"], k)
        assert [[p, i] for p, i in ds.decode_log] == ep["decoded"], (name, ep["episode"])
        assert list(d["support_set"].shape) == ep["support_shape"] and list(d["target_set"].shape) == ep["target_shape"]
        for key in ("support_set", "target_set"):
            flat = d[key].reshape(-1)
            for pos, val in ep[key + "_probe"]:
                assert abs(float(flat[pos]) - val) < 1e-6, (name, key, pos)
            assert abs(float(d[key].double().sum()) - ep[key[:-4] + "_sum"]) < 1e-3 * d[key].numel() ** 0.5
            full = "%s/%d/%s" % (name, ep["episode"], key)
            if full in Z.files:
                assert float((d[key] - torch.from_numpy(Z[full])).abs().max()) < 1e-6


def test_episode_seed_mode_is_index_deterministic(tmp_path):
    ds, _ = _dataset("hmdb_3w2s_q2_rect", tmp_path, EPISODE_SEED=1234)
    random.seed(1)
    a = ds[2]
    random.seed(99)                                           # the global stream is not consulted
    b = ds[2]
    c = ds[1]
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a["batch_class_list"], c["batch_class_list"]) or not torch.equal(a["support_set"], c["support_set"])


def test_loud_failures(tmp_path):
    case = J["cases"]["k100_1shot_linspace"]
    cfg = g2.make_cfg(case["cfg"], str(tmp_path / "nowhere"), str(tmp_path))
    with pytest.raises(FileNotFoundError, match="Data list"):
        n2.Ssv2_few_shot(cfg, "test", decoder=StubReader)
    with pytest.raises(NotImplementedError):
        n2.Ssv2_few_shot(cfg, "train", decoder=StubReader)
    try:
        import decord  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="decoder"):      # no decord, no decoder: refuse, never synthesise frames
            n2.Ssv2_few_shot(cfg, "test")
    # too few videos in a class for shot + query
    anno = tmp_path / "few"
    anno.mkdir()
    (anno / "test_few_shot.txt").write_text("test0//a.avi\ntest1//b.avi\ntest1//c.avi\n")
    c2 = dict(case["cfg"], way=2)
    ds = n2.Ssv2_few_shot(g2.make_cfg(c2, str(anno), str(anno)), "test", decoder=StubReader)
    with pytest.raises(ValueError, match="cannot draw"):
        random.seed(0)
        ds[0]
    # a decoder that returns the wrong layout is caught at the boundary
    class Bad(StubReader):
        def get_batch(self, idx):
            return super().get_batch(idx).permute(0, 3, 1, 2)
    ds_bad, _ = _dataset("hmdb_3w2s_q2_rect", tmp_path)
    ds_bad.decoder = Bad
    with pytest.raises(RuntimeError, match="expected uint8"):
        random.seed(0)
        ds_bad[0]


def test_build_loader_real_dataset_needs_its_list(tmp_path):
    case = J["cases"]["k100_1shot_linspace"]
    cfg = g2.make_cfg(case["cfg"], str(tmp_path / "missing"), str(tmp_path))
    with pytest.raises((FileNotFoundError, ImportError)):
        build_loader(cfg, "test")


@pytest.mark.gpu
def test_episode_frames_through_the_hip_transform(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    name = "hmdb_3w2s_q2_rect"
    ds, case = _dataset(name, tmp_path, use_gpu=True)
    ep = case["episodes"][0]
    random.seed(ep["py_seed"])
    d = ds[ep["episode"]]
    assert d["support_set"].is_cuda
    for key in ("support_set", "target_set"):
        ref = torch.from_numpy(Z["%s/%d/%s" % (name, ep["episode"], key)])
        assert float((d[key].cpu() - ref).abs().max()) < 2e-5
    assert d["support_labels"].tolist() == ep["support_labels"]
