"""build_loader(cfg, split) (reference datasets/base/builder.py:47-94) for the synthetic episode dataset.

The reference's loader yields the 7-key episode dict with a leading batch dim of 1 per GPU
(TEST.BATCH_SIZE / NUM_GPUS = 1) which the test loop strips (runs/test_net_few_shot.py:59-62).  This loader yields
the same dict with a leading dim of ``cfg.TEST.EPISODES_PER_STEP`` episodes (unset / 0: chosen by ``auto_episodes_per_step``), for the static shard
{e : e % world == rank} of a seed-indexed episode list."""
import torch

from ... import synth
from ...utils import distributed as du
from ...utils.registry import Registry

DATASET_REGISTRY = Registry("DATASET")


@DATASET_REGISTRY.register()
class Synthetic_few_shot(torch.utils.data.Dataset):
    """Deterministic structured episodes (clip_fsar_amd.synth.make_episode) in the layout of
    Ssv2_few_shot.__getitem__ (reference datasets/base/ssv2_few_shot.py:275-285)."""

    def __init__(self, cfg, split):
        self.cfg = cfg
        self.split = split
        self.num = int(cfg.TRAIN.NUM_TEST_TASKS)
        arch = synth.ARCHS[cfg.VIDEO.HEAD.BACKBONE_NAME]
        self.res = int(getattr(cfg.DATA, "TEST_CROP_SIZE", arch["res"]))
        self.shot = int(getattr(cfg.TRAIN, "SHOT_TEST", getattr(cfg.TRAIN, "SHOT", 1)))
        self.qpc = int(getattr(cfg.TRAIN, "QUERY_PER_CLASS_TEST", getattr(cfg.TRAIN, "QUERY_PER_CLASS", 1)))

    def __len__(self):
        return self.num

    def __getitem__(self, index):
        cfg = self.cfg
        ep = synth.make_episode(way=int(cfg.TRAIN.WAY), shot=self.shot, query_per_class=self.qpc,
                                frames=int(cfg.DATA.NUM_INPUT_FRAMES), res=self.res,
                                n_test_classes=len(cfg.TEST.CLASS_NAME), episode=int(index),
                                seed=int(getattr(cfg, "RANDOM_SEED", 18)))
        return {k: torch.from_numpy(v) for k, v in ep.items()}


def build_dataset(name, cfg, split):
    cls = DATASET_REGISTRY.get(name)
    if cls is None:
        raise KeyError("dataset %r is not registered (built: Synthetic_few_shot, Ssv2_few_shot -- the latter serves every "
                       "TRAIN.DATASET_FEW list format, as in the reference)" % name)
    return cls(cfg, split)


_LOGGED_K = set()
RN_FRAME_CAP = 2560                  # RN50 tower: frames per launch the default policy goes up to
AUTO_EPISODES_PER_STEP_MAX = 36      # ViT towers: utils/batching.py picks k <= this so that the persistent GEMM grid's rounds come out full


def auto_episodes_per_step(cfg, n_local):
    """Episodes per model call when the config does not say (a reference-shaped config has no TEST.EPISODES_PER_STEP: the reference
    feeds ONE episode per iteration, runs/test_net_few_shot.py:57-64, which leaves a quarter of this tower's throughput on the table).
    The loader collates k episodes -- per-episode results do not depend on k (tests/test_gpu_e2e.py: batch invariance).  ViT towers:
    utils/batching.py::pick_episodes_per_step -- the k that fills the rounds of the persistent GEMM grid best within the frames one tower
    launch may carry (cfg2: 36 episodes = 2 880 frames); RN50: what 2 560 frames hold (32 episodes of 80 frames).  Bounded by the rank's episode count and by a quarter of the device's
    TOTAL HBM over an upper estimate of one episode's footprint (two upload buffers of fp32 frames + the tower's activation workspace):
    deterministic per device and config, logged once."""
    if not (torch.cuda.is_available() and int(getattr(cfg, "NUM_GPUS", 1) or 0) > 0):
        return 1
    from ...utils.batching import FRAME_CAP, pick_episodes_per_step
    arch = synth.ARCHS.get(cfg.VIDEO.HEAD.BACKBONE_NAME, None)
    way = int(getattr(cfg.TRAIN, "WAT_TEST", 0) or cfg.TRAIN.WAY)
    shot = int(getattr(cfg.TRAIN, "SHOT_TEST", getattr(cfg.TRAIN, "SHOT", 1)))
    qpc = int(getattr(cfg.TRAIN, "QUERY_PER_CLASS_TEST", getattr(cfg.TRAIN, "QUERY_PER_CLASS", 1)))
    frames = way * (shot + qpc) * int(cfg.DATA.NUM_INPUT_FRAMES)
    res = int(getattr(cfg.DATA, "TEST_CROP_SIZE", arch["res"] if arch else 224))
    per_frame = 2 * 3 * res * res * 4
    vit = bool(arch) and arch.get("kind") != "rn"
    if vit:
        ntok = (arch["res"] // arch["patch"]) ** 2 + 1
        per_frame += ntok * arch["width"] * 28          # x (two words), qkv, o, u, patches, statistics: < 28 bytes per token-channel
    else:
        per_frame += 12 << 20                           # RN50: NHWC activations of the widest stage, generously
    # from the device's TOTAL memory, not from what happens to be free at the call: the choice must be the same on every rank and in every
    # run (the fp32 tail picks its GEMM kernel by row count: logits are bit-reproducible only at a fixed k)
    try:
        total = torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory
    except Exception:
        return 1
    kmem = int(0.25 * total // max(1, per_frame * frames))
    kcap = max(1, min(AUTO_EPISODES_PER_STEP_MAX, kmem, max(1, int(n_local))))
    if vit:
        launch_frames = min(int(getattr(cfg.VIDEO.HEAD, "MAX_FRAMES_PER_LAUNCH", FRAME_CAP)), (2 ** 32 - 1) // (ntok * 4 * arch["width"] * 2) - 1)
        k = pick_episodes_per_step(frames, ntok, arch["width"], max_frames=max(frames, launch_frames), max_episodes=kcap)
    else:
        # RN50: as many episodes as one tower launch carries inside its 32-bit activation offsets, up to 2 560 frames (32 cfg-rn50 episodes:
        # 635 episodes/s against 609 at 16, profiles/r05_bench_rn50_epoch_size.txt)
        rn_frames = min(RN_FRAME_CAP, (2 ** 32 - 1) // (max(1, res // 2) ** 2 * int(arch["width"] if arch else 64) * 2) - 1)
        k = max(1, min(kcap, rn_frames // max(1, frames)))
    key = (frames, res, k)
    if key not in _LOGGED_K:
        _LOGGED_K.add(key)
        import logging
        # k stays a function of the device's TOTAL memory (same on every rank, in every run); a device that is shared or partly occupied is
        # told so up front instead of running out of memory inside the first step
        try:
            free = torch.cuda.mem_get_info()[0]
            need = int(1.15 * per_frame * frames * k)
            if free < need:
                logging.getLogger(__name__).warning(
                    "TEST.EPISODES_PER_STEP unset: %d episodes per model call need about %.1f GB, the device has %.1f GB free of %.0f GB -- set "
                    "TEST.EPISODES_PER_STEP to a smaller batch if the first step runs out of memory", k, need / 2 ** 30, free / 2 ** 30, total / 2 ** 30)
        except Exception:
            pass
        logging.getLogger(__name__).info("TEST.EPISODES_PER_STEP unset: %d episodes per model call (%d frames each, %.0f GB device memory)",
                                         k, frames, total / 2 ** 30)
    return k


def build_loader(cfg, split):
    assert split in ("test", "val", "train")
    name = getattr(cfg.TEST, "DATASET", "Synthetic_few_shot")
    # TEST.DATASET = Ssv2_few_shot (the reference registers that one class for every few-shot benchmark) builds the video-episode
    # pipeline of datasets/base/ssv2_few_shot.py, which raises when its split list / decoder is missing: synthetic frames are never
    # substituted for a real dataset.
    ds = build_dataset(name, cfg, split)
    idx = du.shard_episodes(len(ds))
    sub = torch.utils.data.Subset(ds, idx)
    bs = int(getattr(cfg.TEST, "EPISODES_PER_STEP", 0) or 0)
    if bs <= 0:
        bs = auto_episodes_per_step(cfg, len(idx))
    # DATA_LOADER.{NUM_WORKERS, PIN_MEMORY} as in the reference (datasets/base/builder.py:83-92; its configs pin): pinned batches are what
    # makes the copy-stream upload of runs/test_net_few_shot.py asynchronous (utils/prefetch.py).  Pinning needs a GPU runtime.
    dl = getattr(cfg, "DATA_LOADER", None)
    workers = int(getattr(dl, "NUM_WORKERS", 0) or 0)
    pin = bool(getattr(dl, "PIN_MEMORY", True)) and torch.cuda.is_available() and int(getattr(cfg, "NUM_GPUS", 1) or 0) > 0
    if bool(getattr(getattr(cfg, "AUGMENTATION", None), "USE_GPU", False)):
        pin = False                 # the dataset's transform runs on the device (preprocess.py): its items are already device tensors
    kw = dict(persistent_workers=True, prefetch_factor=2) if workers > 0 else {}
    return torch.utils.data.DataLoader(sub, batch_size=bs, shuffle=False, num_workers=workers, pin_memory=pin, drop_last=False, **kw)


def shuffle_dataset(loader, cur_epoch):
    pass    # episodes are a fixed seed-indexed list
