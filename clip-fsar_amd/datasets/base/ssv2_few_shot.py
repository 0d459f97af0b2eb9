"""N2: the episode pipeline in front of the hot path -- split list -> way/shot/query episode -> frame indices
-> decoded uint8 frames -> test transform -> the 7-key episode dict the head consumes (8(a) A0).

Mirrors the reference's ``Ssv2_few_shot`` dataset (reference datasets/base/ssv2_few_shot.py:33-84 ``Split_few_shot``,
:190-285 ``__getitem__`` META_BATCH branch, :361-431 ``get_seq``, :614-642 test transform) and the pieces of
``BaseVideoDataset`` it uses (reference datasets/base/base_dataset.py:232-280 ``_decode_video``, :493-530
``_interval_based_sampling``).  Same class / method names, same cfg keys, and -- because evaluation protocols quote accuracies
over *these* random episodes -- the same consumption of Python's ``random`` stream: seeding ``random`` identically gives the
reference's episode (classes, videos, frame indices, shuffles) bit for bit (tests/test_episodes_n2.py, goldens produced by the
real reference class with a stubbed decoder: oracle/make_golden_n2.py).

Where this differs from the reference, by design:
  * Decoding is a pluggable ``decoder(path) -> reader`` (``len(reader)``, ``reader.get_avg_fps()``, ``reader.get_batch(indices)
    -> uint8 [T, H, W, 3]``), given as ``decoder=`` or ``DATA.DECODER`` (callable or "module:attr").  The default is decord's
    ``VideoReader`` like the reference; when decord is not importable the dataset refuses to construct (it never substitutes
    synthetic frames).
  * The test transform (ToTensorVideo -> KineticsResizedCropFewshot -> NormalizeVideo -> permute) runs as ONE HIP kernel on the
    device when ``AUGMENTATION.USE_GPU`` is set (clip_fsar_amd.preprocess.preprocess_video, csrc/preprocess.hip: resize + crop
    + normalise + layout in a single pass over the uint8 frames); otherwise on the host with the reference's own torch ops
    (the reference's DataLoader-worker path).  Both are checked against the reference golden.
  * Only the evaluation splits are built: the training augmentations (random crop / flip / colour jitter / random erasing,
    reference :538-612) belong to the training path, which SURVEY.md 8 scopes out; ``split == "train"`` raises.
  * ``TEST.EPISODE_SEED`` (optional, not in the reference): episode ``index`` draws from ``random.Random(seed + index)`` instead
    of the process-global stream, so that a sharded multi-GPU evaluation sees the same episodes at any world size.
  * A video shorter than NUM_INPUT_FRAMES in segment mode makes the reference loop through ten failed decodes and then recurse
    into another episode (``randint(0, -1)``, :505-509 -> :386-399); here it raises with the path.
"""
import os
import random

import torch

from ...utils import logging as log_utils
from .builder import DATASET_REGISTRY

logger = log_utils.get_logger(__name__)


def parse_split_line(line, split_dataset, dataset):
    """One line of ``{train,test}_few_shot.txt`` -> (video path or id, class id).

    ``Ssv2_few_shot`` lists are ``test{cls}/{video id}``; every other DATASET_FEW (Kinetics / HMDB / UCF lists) is
    ``test{cls}//{relative path}`` (reference :42-54; shipped lists e.g. configs/projects/CLIPFSAR/kinetics100/test_few_shot.txt)."""
    s = line.strip()
    sep = "/" if dataset == "Ssv2_few_shot" else "//"
    parts = s.split(sep)
    head = parts[0]
    if not head.startswith(split_dataset) or not head[len(split_dataset):].isdigit():
        raise ValueError("split-list line %r does not start with '%s<class id>%s'" % (line, split_dataset, sep))
    return parts[-1], int(head[len(split_dataset):])


class Split_few_shot:
    """Video paths and class ids of one split (reference :33-84)."""

    def __init__(self, folder, split_dataset="train", dataset="Ssv2_few_shot"):
        self.gt_a_list = []
        self.videos = []
        self.split_dataset = split_dataset
        for class_folder in folder:
            if not class_folder.strip():
                continue
            paths, class_id = parse_split_line(class_folder, split_dataset, dataset)
            self.add_vid(paths, class_id)
        self._by_class = {}
        for i, g in enumerate(self.gt_a_list):
            self._by_class.setdefault(g, []).append(i)
        logger.info("loaded {} videos from {} dataset: {} !".format(len(self.gt_a_list), split_dataset, dataset))

    def add_vid(self, paths, gt_a):
        self.videos.append(paths)
        self.gt_a_list.append(gt_a)

    def get_rand_vid(self, label, idx=-1, rng=random):
        match_idxs = self._by_class.get(label, [])
        if idx != -1:
            return self.videos[match_idxs[idx]], match_idxs[idx]
        random_idx = rng.choice(match_idxs)          # (the reference calls an unimported numpy here: never reached by its loop)
        return self.videos[random_idx], random_idx

    def get_single_video(self, index):
        return self.videos[index], self.gt_a_list[index]

    def get_num_videos_for_class(self, label):
        return len(self._by_class.get(label, []))

    def get_unique_classes(self):
        return list(set(self.gt_a_list))             # the reference's order (set iteration) feeds random.sample: keep it

    def __len__(self):
        return len(self.gt_a_list)


def interval_based_sampling(cfg, split, vid_length, vid_fps, clip_idx, num_clips, num_frames, interval, rng=random):
    """Frame indices of one clip (reference base_dataset.py:493-530).  Returns a list of ints.

    SAMPLING_RATE(_TEST) > 40 (every shipped config but Kinetics100): one uniformly random frame out of each of ``num_frames``
    equal segments.  Otherwise a clip of ``num_frames * interval * fps / TARGET_FPS`` source frames, centred for testing
    (clip_idx 0 of 1), random start for clip_idx -1, sampled with a float32 ``linspace``."""
    data = cfg.DATA
    if num_frames == 1:
        return [rng.randint(0, vid_length - 1)]
    segments = False
    if split == "train" and hasattr(data, "SAMPLING_RATE_TRAIN"):
        interval = data.SAMPLING_RATE_TRAIN
    elif hasattr(data, "SAMPLING_RATE_TEST") and data.SAMPLING_RATE_TEST > 40:
        segments = True
    elif data.SAMPLING_RATE > 40:
        segments = True
    if segments:
        seg = vid_length // num_frames
        if seg < 1:
            raise ValueError("video of %d frames is shorter than NUM_INPUT_FRAMES = %d" % (vid_length, num_frames))
        return [rng.randint(ind * seg, ind * seg + seg - 1) for ind in range(num_frames)]
    clip_length = num_frames * interval * vid_fps / data.TARGET_FPS
    if clip_length > vid_length:
        clip_length = vid_length // num_frames * num_frames
    max_idx = max(vid_length - clip_length + 1, 0)
    if clip_idx == -1:
        start_idx = rng.uniform(0, max_idx)
    elif num_clips == 1:
        start_idx = max_idx / 2
    else:
        start_idx = max_idx * clip_idx / num_clips
    end_idx = start_idx + clip_length - interval
    index = torch.linspace(start_idx, end_idx, num_frames)
    return torch.clamp(index, 0, vid_length - 1).long().tolist()


def _decord_decoder():
    try:
        import decord                                                   # noqa: F401
        from decord import VideoReader
    except ImportError as e:
        raise ImportError("Ssv2_few_shot needs a video decoder: decord is not importable here and no decoder= was given "
                          "(clip_fsar_amd never substitutes synthetic frames for a real dataset)") from e
    decord.bridge.set_bridge("native")

    class _Reader:
        def __init__(self, path):
            self.vr = VideoReader(path)

        def __len__(self):
            return len(self.vr)

        def get_avg_fps(self):
            return self.vr.get_avg_fps()

        def get_batch(self, idx):
            return torch.utils.dlpack.from_dlpack(self.vr.get_batch(idx).to_dlpack()).clone()

    return _Reader


@DATASET_REGISTRY.register()
class Ssv2_few_shot(torch.utils.data.Dataset):
    """The reference registers this one class for every few-shot benchmark; ``TRAIN.DATASET_FEW`` picks the list format and
    the path rule (reference :88, :135-138, :368-371)."""

    def __init__(self, cfg, split, decoder=None):
        if split == "train":
            raise NotImplementedError("training episodes (augmentations, reference ssv2_few_shot.py:538-612) are out of scope: "
                                      "SURVEY.md 8 builds the evaluation path")
        if split not in ("val", "test"):
            raise NotImplementedError("Split not supported")
        self.cfg = cfg
        self.split = split
        self.split_dataset = split
        self.data_root_dir = getattr(cfg.DATA, "DATA_ROOT_DIR", None)
        self.anno_dir = getattr(cfg.DATA, "ANNO_DIR", None)
        if not self.data_root_dir or not self.anno_dir:
            raise FileNotFoundError("Ssv2_few_shot needs DATA.DATA_ROOT_DIR (videos) and DATA.ANNO_DIR ({train,test}_few_shot.txt); "
                                    "for synthetic episodes set TEST.DATASET: Synthetic_few_shot")
        self.dataset_name = getattr(cfg.TRAIN, "DATASET_FEW", cfg.TEST.DATASET if split == "test" else cfg.TRAIN.DATASET)
        self._num_frames = int(cfg.DATA.NUM_INPUT_FRAMES)
        self._sampling_rate = cfg.DATA.SAMPLING_RATE
        self.gpu_transform = bool(getattr(getattr(cfg, "AUGMENTATION", None), "USE_GPU", False))
        if decoder is None:
            decoder = getattr(cfg.DATA, "DECODER", None)          # optional: a callable or "package.module:attribute"
            if isinstance(decoder, str):
                import importlib
                mod, _, attr = decoder.partition(":")
                decoder = getattr(importlib.import_module(mod), attr)
        self.decoder = decoder if decoder is not None else _decord_decoder()
        self.rng = random                               # the process-global stream, like the reference
        self._episode_seed = getattr(cfg.TEST, "EPISODE_SEED", None)
        self.decode_log = None                          # tests set a list: (path, indices) per decoded video
        self._construct_dataset(cfg)
        self._config_transform()

    # ------------------------------------------------------------------ split list (reference :133-188)
    def _get_dataset_list_name(self):
        return "{}_few_shot.txt".format("train" if self.split == "train" else "test")

    def _construct_dataset(self, cfg):
        name = self._get_dataset_list_name()
        path = os.path.join(self.anno_dir, name)
        if str(path).startswith("oss"):
            raise NotImplementedError("oss:// annotation lists (reference utils/bucket.py) are out of scope; copy the list locally")
        if not os.path.isfile(path):
            raise FileNotFoundError("Data list {} not found (DATA.ANNO_DIR = {!r})".format(path, self.anno_dir))
        with open(path) as f:
            lines = f.readlines()
        self._samples = [l.strip() for l in lines]
        # a non-train split ("test", and "val" as the training script's validation pass uses it) reads test_few_shot.txt, whose lines
        # start with "test<class id>" (reference :146-188)
        self.split_few_shot = Split_few_shot(lines, "train" if self.split == "train" else "test", dataset=self.dataset_name)
        assert len(self.split_few_shot) != 0, "Empty sample list {}".format(path)
        logger.info("Dataset {} split {} loaded. Length {}.".format(self.dataset_name, self.split, len(self.split_few_shot)))

    def __len__(self):
        n = getattr(self.cfg.TRAIN, "NUM_TEST_TASKS", 0)
        return int(n) if n else len(self.split_few_shot)

    # ------------------------------------------------------------------ transform (reference :614-642)
    def _config_transform(self):
        d = self.cfg.DATA
        ts = d.TEST_SCALE
        self._scale = [int(ts[0]), int(ts[1])] if isinstance(ts, (list, tuple)) else [int(ts), int(ts)]
        self._crop = int(d.TEST_CROP_SIZE)
        # The reference derives (temporal clip, spatial crop) of a video from _spatial_temporal_index[vid_id] (base_dataset.py:253-262);
        # its few-shot configs run ONE view (NUM_ENSEMBLE_VIEWS = NUM_SPATIAL_CROPS = 1).  get_seq takes clip 0 / crop 0, so a config
        # that asks for more views would silently diverge from the reference: refuse it.
        self._nsc = int(getattr(self.cfg.TEST, "NUM_SPATIAL_CROPS", 1))
        nev = int(getattr(self.cfg.TEST, "NUM_ENSEMBLE_VIEWS", 1))
        if self._nsc != 1 or nev != 1:
            raise NotImplementedError("TEST.NUM_SPATIAL_CROPS = %d / NUM_ENSEMBLE_VIEWS = %d: only the single-view episode sampling of the "
                                      "CLIP-FSAR configs is implemented (reference base_dataset.py:253-262)" % (self._nsc, nev))
        self._mean, self._std = [float(x) for x in d.MEAN], [float(x) for x in d.STD]

    def transform(self, frames_u8, spatial_idx=0):
        """uint8 [T, H, W, 3] -> float32 [T, 3, crop, crop] (the reference's Compose followed by get_seq's permute, :425-431)."""
        from ...preprocess import crop_window, preprocess_video
        if self.gpu_transform:
            return preprocess_video(frames_u8.cuda(non_blocking=True), self._scale, self._crop, self._mean, self._std,
                                    self._nsc, spatial_idx)
        y0, x0 = crop_window(self._scale, self._crop, self._nsc, spatial_idx)
        clip = frames_u8.float().permute(3, 0, 1, 2) / 255.0                                   # ToTensorVideo: C, T, H, W
        clip = torch.nn.functional.interpolate(clip, size=(self._scale[0], self._scale[1]), mode="bilinear")
        clip = clip[:, :, y0:y0 + self._crop, x0:x0 + self._crop]
        m = torch.tensor(self._mean).reshape(3, 1, 1, 1)
        s = torch.tensor(self._std).reshape(3, 1, 1, 1)
        return ((clip - m) / s).permute(1, 0, 2, 3).contiguous()                                # NormalizeVideo, then T, C, H, W

    # ------------------------------------------------------------------ one video (reference :361-431 + base_dataset.py:232-280)
    def _video_path(self, paths):
        if self.dataset_name == "Ssv2_few_shot":
            return os.path.join(self.data_root_dir, paths + ".mp4")
        return os.path.join(self.data_root_dir, paths)

    def _get_video_frames_list(self, vid_length, vid_fps, clip_idx, rng):
        mode = getattr(self.cfg.DATA, "SAMPLING_MODE", "interval_based")
        if mode != "interval_based":
            raise NotImplementedError("DATA.SAMPLING_MODE = %r (the CLIP-FSAR configs use interval_based)" % mode)
        return interval_based_sampling(self.cfg, self.split, vid_length, vid_fps, clip_idx,
                                       int(getattr(self.cfg.TEST, "NUM_ENSEMBLE_VIEWS", 1)), self._num_frames,
                                       self._sampling_rate, rng)

    def get_seq(self, label, idx=-1, rng=None):
        """(frames float32 [T, 3, crop, crop], index of the video in the split list)."""
        rng = rng or self.rng
        paths, vid_id = self.split_few_shot.get_rand_vid(label, idx, rng)
        path = self._video_path(paths)
        reader = self.decoder(path)
        # test / val: the first (only) temporal clip and spatial crop 0 (base_dataset.py:253-262 with _num_clips = 1, :144)
        clip_idx = -1 if self.split == "val" else 0
        index = self._get_video_frames_list(len(reader), reader.get_avg_fps(), clip_idx, rng)
        frames = reader.get_batch(index)
        if not isinstance(frames, torch.Tensor):
            frames = torch.as_tensor(frames)
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3 or frames.shape[0] != len(index):
            raise RuntimeError("decoder returned %s %s for %s: expected uint8 [%d, H, W, 3]"
                               % (frames.dtype, tuple(frames.shape), path, len(index)))
        if self.decode_log is not None:
            self.decode_log.append((path, [int(i) for i in index]))
        return self.transform(frames, 0), vid_id

    # ------------------------------------------------------------------ one episode (reference :190-285)
    def __getitem__(self, index):
        cfg = self.cfg
        if not getattr(cfg.TRAIN, "META_BATCH", True):
            raise NotImplementedError("TRAIN.META_BATCH = False (plain per-video classification) is not part of CLIP-FSAR's path")
        rng = self.rng if self._episode_seed is None else random.Random(int(self._episode_seed) + int(index))
        c = self.split_few_shot
        classes = c.get_unique_classes()
        batch_classes = rng.sample(classes, cfg.TRAIN.WAY)
        if hasattr(cfg.TRAIN, "WAT_TEST"):                                  # (sic: the reference's spelling, :210)
            batch_classes = rng.sample(classes, cfg.TRAIN.WAT_TEST)
        n_queries = cfg.TRAIN.QUERY_PER_CLASS_TEST
        shot = cfg.TRAIN.SHOT_TEST if hasattr(cfg.TRAIN, "SHOT_TEST") else cfg.TRAIN.SHOT
        support, target = [], []                                            # (frames, episode label, dataset class id)
        for bl, bc in enumerate(batch_classes):
            n_total = c.get_num_videos_for_class(bc)
            if n_total < shot + n_queries:
                raise ValueError("class %d of %s has %d videos: cannot draw %d shot + %d query"
                                 % (bc, self._get_dataset_list_name(), n_total, shot, n_queries))
            idxs = rng.sample([i for i in range(n_total)], shot + n_queries)
            for idx in idxs[0:shot]:
                vid, _ = self.get_seq(bc, idx, rng)
                support.append((vid, bl, bc))
            for idx in idxs[shot:]:
                vid, _ = self.get_seq(bc, idx, rng)
                target.append((vid, bl, bc))
        rng.shuffle(support)
        rng.shuffle(target)
        ft = lambda xs: torch.tensor([float(x) for x in xs], dtype=torch.float32)
        return {"support_set": torch.cat([s[0] for s in support]), "support_labels": ft(s[1] for s in support),
                "target_set": torch.cat([t[0] for t in target]), "target_labels": ft(t[1] for t in target),
                "real_target_labels": ft(t[2] for t in target), "batch_class_list": ft(batch_classes),
                "real_support_labels": ft(s[2] for s in support)}
