#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not product code): golden vectors for the N2 episode pipeline, produced by
running the REAL reference dataset class -- datasets/base/ssv2_few_shot.py ``Ssv2_few_shot`` on top of
datasets/base/base_dataset.py ``BaseVideoDataset`` -- in THIS container with a stubbed video decoder.

What is stubbed and why (none of it is part of the algorithm under test):
  * ``decord.VideoReader``   -> a reader of deterministic synthetic uint8 frames keyed by (path, frame index); the video
                               length / fps are a hash of the path.  Every (path, index list) it is asked for is logged.
  * ``torchvision.transforms._transforms_video`` (third-party, torchvision 0.x, not in this image): ``ToTensorVideo`` and
    ``NormalizeVideo`` restated from their published definitions: uint8 THWC -> float CTHW / 255 ; (x - mean) / std.
  * ``oss2`` / ``simplejson`` / ``ipdb`` ...: unused imports.

Output: tests/golden/episodes_n2.json (split lists, per-case config, the labels / class lists / decoded (path, frame indices)
per episode slot) and tests/golden/episodes_n2_frames.npz (the final support_set / target_set tensors of the small cases).
Run:  python oracle/make_golden_n2.py        (needs /root/reference; commit the outputs)."""
import hashlib
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
REFERENCE_ROOT = "/root/reference"

CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
CLIP_STD = [0.26862954, 0.26130258, 0.27577711]

# ----------------------------------------------------------------------------------------------- synthetic "videos"


def video_meta(path):
    """(length, fps, H, W) of the synthetic video behind ``path`` -- shared with tests/test_episodes_n2.py."""
    h = int(hashlib.sha1(path.encode()).hexdigest()[:8], 16)
    length = 9 + h % 57                       # 9 .. 65 frames (some shorter than 2 x num_frames)
    fps = [12.0, 24.0, 25.0, 30.0][(h >> 8) % 4]
    H, W = [(36, 48), (40, 40), (30, 52)][(h >> 12) % 3]
    return length, fps, H, W


def video_frames(path, indices):
    """uint8 [len(indices), H, W, 3]: a function of (path, frame index) only."""
    length, _, H, W = video_meta(path)
    out = np.empty((len(indices), H, W, 3), np.uint8)
    for j, i in enumerate(indices):
        i = int(i)
        assert 0 <= i < length, (path, i, length)
        seed = int(hashlib.sha1(("%s#%d" % (path, i)).encode()).hexdigest()[:8], 16)
        out[j] = np.random.RandomState(seed).randint(0, 256, size=(H, W, 3), dtype=np.uint8)
    return out


DECODE_LOG = []


class _Batch:
    def __init__(self, arr):
        self.t = torch.from_numpy(arr)

    def to_dlpack(self):
        return torch.utils.dlpack.to_dlpack(self.t)


class StubVideoReader:
    def __init__(self, path, *a, **k):
        self.path = path
        self.length, self.fps, _, _ = video_meta(path)

    def __len__(self):
        return self.length

    def get_avg_fps(self):
        return self.fps

    def get_batch(self, idx):
        idx = [int(i) for i in (idx.tolist() if hasattr(idx, "tolist") else idx)]
        DECODE_LOG.append((self.path, idx))
        return _Batch(video_frames(self.path, idx))


# ----------------------------------------------------------------------------------------------- reference import


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Compose:
    def __init__(self, ts):
        self.transforms = ts

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class _ToTensorVideo:                     # torchvision.transforms._transforms_video.ToTensorVideo
    def __call__(self, clip):
        assert clip.dtype == torch.uint8 and clip.dim() == 4
        return clip.float().permute(3, 0, 1, 2) / 255.0


class _NormalizeVideo:                    # torchvision.transforms._transforms_video.NormalizeVideo
    def __init__(self, mean, std, inplace=False):
        self.mean, self.std = mean, std

    def __call__(self, clip):
        m = torch.as_tensor(self.mean, dtype=clip.dtype).reshape(3, 1, 1, 1)
        s = torch.as_tensor(self.std, dtype=clip.dtype).reshape(3, 1, 1, 1)
        return (clip - m) / s


def import_reference_dataset():
    sys.dont_write_bytecode = True
    ident = lambda *a, **k: None
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models")
    tv.utils = _stub("torchvision.utils", make_grid=ident, save_image=ident)
    tv.transforms = _stub("torchvision.transforms", Compose=_Compose, Lambda=ident, Resize=ident, CenterCrop=ident,
                          ToTensor=ident, Normalize=ident)
    tv.transforms.__path__ = []
    tv.transforms._functional_video = _stub("torchvision.transforms._functional_video")
    tv.transforms._transforms_video = _stub("torchvision.transforms._transforms_video", ToTensorVideo=_ToTensorVideo,
                                            NormalizeVideo=_NormalizeVideo, RandomHorizontalFlipVideo=ident)
    _stub("ipdb", set_trace=ident)
    _stub("oss2")
    _stub("simplejson")
    _stub("joblib")
    dec = _stub("decord", VideoReader=StubVideoReader, cpu=ident, gpu=ident)
    dec.bridge = types.SimpleNamespace(set_bridge=ident)
    for k in [k for k in sys.modules if k.split(".")[0] in ("models", "utils", "datasets", "sslgenerators")]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    # datasets/base/__init__.py imports every dataset of the code base; only ssv2_few_shot is wanted: import the package
    # shells by hand and the one module directly
    for pkg in ("datasets", "datasets.base", "datasets.utils"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REFERENCE_ROOT, *pkg.split("."))]
        sys.modules[pkg] = m
    import importlib
    mod = importlib.import_module("datasets.base.ssv2_few_shot")
    assert mod.__file__.startswith(REFERENCE_ROOT), mod.__file__
    return mod


class NS:
    """cfg node: attribute access + hasattr() semantics of the reference's Config."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def make_cfg(case, anno_dir, out_dir):
    c = case
    return NS(
        TRAIN=NS(META_BATCH=True, WAY=c["way"], SHOT=c["shot"], QUERY_PER_CLASS=c.get("qpc_train", 5),
                 QUERY_PER_CLASS_TEST=c["qpc"], NUM_TEST_TASKS=c["tasks"], NUM_SAMPLES=100, DATASET="Ssv2_few_shot",
                 DATASET_FEW=c["dataset_few"], **({"SHOT_TEST": c["shot_test"]} if "shot_test" in c else {}),
                 **({"WAT_TEST": c["way_test"]} if "way_test" in c else {})),
        TEST=NS(DATASET="Ssv2_few_shot", NUM_ENSEMBLE_VIEWS=1, NUM_SPATIAL_CROPS=1, ZERO_SHOT=False),
        DATA=NS(DATA_ROOT_DIR="/data/root", ANNO_DIR=anno_dir, NUM_INPUT_FRAMES=c["frames"], SAMPLING_RATE=c["rate"],
                TARGET_FPS=c.get("target_fps", 12), SAMPLING_MODE="interval_based", TEST_SCALE=c["scale"],
                TEST_CROP_SIZE=c["crop"], TEST_CENTER_CROP=True, MEAN=CLIP_MEAN, STD=CLIP_STD,
                **({"SAMPLING_RATE_TEST": c["rate_test"]} if "rate_test" in c else {})),
        AUGMENTATION=NS(USE_GPU=False), PRETRAIN=NS(ENABLE=False), OUTPUT_DIR=out_dir,
        VIDEO=NS(BACKBONE=NS(META_ARCH="Identity")), OSS=NS())


def synth_split_list(fmt, n_classes, per_class, split="test", seed=0):
    """Split-list lines in the two formats the reference ships (configs/projects/CLIPFSAR/*/test_few_shot.txt):
    'ssv2':   test{cls}/{video id}                 (Ssv2_few_shot; ".mp4" is appended to the id)
    'path':   test{cls}//{relative/path.ext}       (Kinetics_few_shot, HMDB, UCF)."""
    rs = np.random.RandomState(seed)
    lines = []
    for cls in range(n_classes):
        for v in range(per_class + int(rs.randint(0, 3))):
            if fmt == "ssv2":
                lines.append("%s%d/%d" % (split, cls, int(rs.randint(1, 220000))))
            else:
                lines.append("%s%d//videos/class_%02d/v_%02d_g%02d.avi" % (split, cls, cls, cls, v))
    order = rs.permutation(len(lines))
    return [lines[i] for i in order]


CASES = {
    # name: way/shot/query as the shipped configs use them, small frames so that the tensors fit a fixture
    "k100_1shot_linspace": dict(dataset_few="Kinetics_few_shot", fmt="path", way=5, shot=1, qpc=1, frames=8, rate=35,
                                scale=32, crop=24, tasks=6, seed=18, episodes=[0, 1, 5], n_classes=8, per_class=6),
    "ssv2_5shot_segments": dict(dataset_few="Ssv2_few_shot", fmt="ssv2", way=5, shot=5, qpc=1, frames=8, rate=50,
                                scale=32, crop=24, tasks=4, seed=7, episodes=[0, 3], n_classes=7, per_class=8),
    "hmdb_3w2s_q2_rect": dict(dataset_few="HMDB_few_shot", fmt="path", way=3, shot=2, qpc=2, frames=4, rate=35,
                              scale=[32, 40], crop=24, tasks=3, seed=3, episodes=[0, 2], n_classes=5, per_class=5,
                              target_fps=25),
    "ucf_shot_test_rate_test": dict(dataset_few="UCF_few_shot", fmt="path", way=4, shot=1, shot_test=3, qpc=1, frames=6,
                                    rate=35, rate_test=60, scale=28, crop=24, tasks=2, seed=11, episodes=[1],
                                    n_classes=6, per_class=6),
}


FULL_TENSOR_CASES = ("hmdb_3w2s_q2_rect",)


def main():
    mod = import_reference_dataset()
    out_json = {"cases": {}, "mean": CLIP_MEAN, "std": CLIP_STD}
    out_npz = {}
    tmp = os.path.join("/tmp", "n2_golden_%d" % os.getpid())
    os.makedirs(tmp, exist_ok=True)
    for name, c in CASES.items():
        lines = synth_split_list(c["fmt"], c["n_classes"], c["per_class"], seed=c["seed"])
        anno = os.path.join(tmp, name)
        os.makedirs(anno, exist_ok=True)
        with open(os.path.join(anno, "test_few_shot.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
        cfg = make_cfg(c, anno, anno)
        ds = mod.Ssv2_few_shot(cfg, "test")
        assert len(ds) == c["tasks"]
        eps = []
        for e in c["episodes"]:
            random.seed(c["seed"] * 1000 + e)
            del DECODE_LOG[:]
            d = ds[e]
            rec = {k: [float(x) for x in d[k].tolist()] for k in
                   ("support_labels", "target_labels", "real_support_labels", "real_target_labels", "batch_class_list")}
            rec["py_seed"] = c["seed"] * 1000 + e
            rec["decoded"] = [[p, idx] for p, idx in DECODE_LOG]        # in DECODE order (before the shuffles)
            rec["support_shape"], rec["target_shape"] = list(d["support_set"].shape), list(d["target_set"].shape)
            rec["support_sum"] = float(d["support_set"].double().sum())
            rec["target_sum"] = float(d["target_set"].double().sum())
            for key in ("support_set", "target_set"):         # 48 probe values per tensor at fixed flat positions
                flat = d[key].reshape(-1)
                pos = np.random.RandomState(flat.numel() % 9973).randint(0, flat.numel(), size=48)
                rec[key + "_probe"] = [[int(i), float(flat[int(i)])] for i in pos]
            if name in FULL_TENSOR_CASES and e == c["episodes"][0]:      # full tensors of the smallest case only (fixture size)
                out_npz["%s/%d/support_set" % (name, e)] = d["support_set"].numpy().astype(np.float32)
                out_npz["%s/%d/target_set" % (name, e)] = d["target_set"].numpy().astype(np.float32)
            eps.append({"episode": e, **rec})
            print("%-26s episode %d: support %s target %s classes %s" % (name, e, rec["support_shape"], rec["target_shape"],
                                                                      rec["batch_class_list"]))
        sp = ds.split_few_shot
        out_json["cases"][name] = {"cfg": {k: v for k, v in c.items() if k not in ("episodes",)}, "split_lines": lines,
                                   "parsed": {"videos": list(sp.videos), "class_ids": [int(x) for x in sp.gt_a_list],
                                              "unique_classes": [int(x) for x in sp.get_unique_classes()]},
                                   "len": len(ds), "episodes": eps}
    # _interval_based_sampling alone, over a grid (reference base_dataset.py:493-530), test split, clip_idx 0 / num_clips 1
    grid = []
    ds_lin = mod.Ssv2_few_shot(make_cfg(CASES["k100_1shot_linspace"], os.path.join(tmp, "k100_1shot_linspace"),
                                        os.path.join(tmp, "k100_1shot_linspace")), "test")
    ds_seg = mod.Ssv2_few_shot(make_cfg(CASES["ssv2_5shot_segments"], os.path.join(tmp, "ssv2_5shot_segments"),
                                        os.path.join(tmp, "ssv2_5shot_segments")), "test")
    for ds, tag in ((ds_lin, "linspace"), (ds_seg, "segments")):
        for length in (8, 9, 15, 16, 17, 31, 64, 100, 301):
            for fps in (12.0, 25.0, 30.0):
                for nf in (1, 4, 8, 16):
                    if tag == "segments" and length < nf:
                        continue                          # interval 0 -> randint(0, -1) raises in the reference
                    random.seed(length * 131 + nf)
                    idx = ds._interval_based_sampling(length, fps, 0, 1, nf, ds._sampling_rate)
                    idx = [int(i) for i in (idx.tolist() if hasattr(idx, "tolist") else idx)]
                    grid.append({"mode": tag, "rate": int(ds._sampling_rate), "target_fps": int(ds.cfg.DATA.TARGET_FPS),
                                 "length": length, "fps": fps, "frames": nf, "py_seed": length * 131 + nf, "index": idx})
    out_json["sampling_grid"] = grid
    # the shipped split lists: a few head lines of each format + totals, as data (configs/projects/CLIPFSAR/*/test_few_shot.txt)
    shipped = {}
    for d, few in (("kinetics100", "Kinetics_few_shot"), ("ssv2_full", "Ssv2_few_shot"), ("ssv2_small", "Ssv2_few_shot"),
                   ("hmdb51", "HMDB_few_shot"), ("ucf101", "UCF_few_shot")):
        p = os.path.join(REFERENCE_ROOT, "configs", "projects", "CLIPFSAR", d, "test_few_shot.txt")
        with open(p) as f:
            ls = f.readlines()
        sp = mod.Split_few_shot(ls, "test", dataset=few)
        cls = sp.get_unique_classes()
        shipped[d] = {"dataset_few": few, "head": [l.rstrip("\n") for l in ls[:3]], "n_videos": len(sp), "n_classes": len(cls),
                      "class_hist_sha1": hashlib.sha1(json.dumps(sorted((int(k), sp.get_num_videos_for_class(k)) for k in cls)).encode()).hexdigest(),
                      "first_parsed": [sp.videos[0], int(sp.gt_a_list[0])], "last_parsed": [sp.videos[-1], int(sp.gt_a_list[-1])]}
    out_json["shipped_lists"] = shipped
    with open(os.path.join(GOLD, "episodes_n2.json"), "w") as f:
        json.dump(out_json, f, indent=0)
    np.savez_compressed(os.path.join(GOLD, "episodes_n2_frames.npz"), **out_npz)
    print("wrote episodes_n2.json (%d cases, %d sampling rows) and episodes_n2_frames.npz (%d arrays)"
          % (len(out_json["cases"]), len(grid), len(out_npz)))


if __name__ == "__main__":
    main()
