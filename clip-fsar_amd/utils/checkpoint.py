"""Checkpoint loading for the test run, with the reference's names and priority order
(reference utils/checkpoint.py:62-96 get_last_checkpoint / has_checkpoint, :279-350 load_checkpoint, :452-530
load_test_checkpoint):

    TEST.CHECKPOINT_FILE_PATH  ->  newest "*checkpoint*" file in OUTPUT_DIR/checkpoints  ->  TRAIN.CHECKPOINT_FILE_PATH
    ->  random initialisation ("only for debugging").

Checkpoints are the reference's ``.pyth`` files: ``torch.save({'epoch': e, 'model_state': model.state_dict(), ...})`` where
the model is ``BaseVideoModel`` -> keys ``head.backbone.*``, ``head.context2.*``, ``head.scale`` (a DDP-saved file carries the
same keys: the reference saves ``model.module.state_dict()``, :118).  Loading is ``strict=False`` and logs the key lists like
the reference (:329-331).

Two departures, both on the side of failing loudly: a checkpoint was GIVEN, so
  * if none of its keys matches the model (wrong prefix, wrong architecture), or
  * if any ``backbone.*`` / ``context2.*`` parameter of the head is missing from it,
loading raises instead of leaving random-init weights in place behind an "INFO: loaded" line.  ``oss://`` paths (the reference's
object-storage branch, :460-476) are out of scope and raise.
"""
import os

import torch

from . import logging as log_utils

logger = log_utils.get_logger(__name__)


def get_checkpoint_dir(path_to_job):
    return os.path.join(path_to_job or "", "checkpoints")


def has_checkpoint(path_to_job):
    d = get_checkpoint_dir(path_to_job)
    files = os.listdir(d) if os.path.isdir(d) else []
    return any("checkpoint" in f for f in files)


def get_last_checkpoint(path_to_job):
    """Newest checkpoint of a job: the lexicographically last file whose name CONTAINS "checkpoint" (reference :69-75)."""
    d = get_checkpoint_dir(path_to_job)
    names = [f for f in (os.listdir(d) if os.path.isdir(d) else []) if "checkpoint" in f]
    if not names:
        return None
    return os.path.join(d, sorted(names)[-1])


def _critical_missing(missing):
    return [k for k in missing if (".backbone." in "." + k or ".context2." in "." + k) and "num_batches_tracked" not in k]


def load_checkpoint(cfg, path_to_checkpoint, model, model_ema=None, data_parallel=False, optimizer=None, inflation=False,
                    pre_process=False):
    """Reference signature (:279-287).  Returns the checkpoint's epoch (-1 if absent), like the reference."""
    if str(path_to_checkpoint).split(":")[0] == "oss":
        raise NotImplementedError("oss:// checkpoints (reference utils/bucket.py) are out of scope; copy the file locally")
    assert os.path.exists(path_to_checkpoint), "Checkpoint '{}' not found".format(path_to_checkpoint)
    if inflation or pre_process or optimizer is not None:
        raise NotImplementedError("2D->3D inflation / checkpoint pre-processing / optimizer state belong to the training path")
    ms = model.module if (data_parallel and hasattr(model, "module")) else model
    with open(path_to_checkpoint, "rb") as f:
        checkpoint = torch.load(f, map_location="cpu")
    if "model_state" not in checkpoint:
        raise KeyError("'%s' is not a reference-format checkpoint: no 'model_state' entry (keys: %s)"
                       % (path_to_checkpoint, sorted(checkpoint)[:8]))
    state = checkpoint["model_state"]
    own = ms.state_dict()
    mismatch = ms.load_state_dict(state, strict=False)
    logger.info("Keys in model not matched: {}".format(mismatch[0]))
    logger.info("Keys in checkpoint not matched: {}".format(mismatch[1]))
    matched = [k for k in state if k in own]
    if not matched:
        raise RuntimeError("checkpoint '%s': none of its %d keys matches the model (first checkpoint key %r, first model key %r)"
                           % (path_to_checkpoint, len(state), next(iter(state), None), next(iter(own), None)))
    crit = _critical_missing(mismatch[0])
    if crit:
        raise RuntimeError("checkpoint '%s' lacks %d backbone / context2 parameters of the model (e.g. %s): refusing to test "
                           "with random-init weights" % (path_to_checkpoint, len(crit), crit[:3]))
    if "model_ema_state" in checkpoint and model_ema is not None:
        (model_ema.module if (data_parallel and hasattr(model_ema, "module")) else model_ema).load_state_dict(
            checkpoint["model_ema_state"], strict=False)
    head = getattr(ms, "head", ms)
    if hasattr(head, "invalidate_engine"):
        head.invalidate_engine()                     # device-side packed weights are rebuilt from the new parameters
    return int(checkpoint.get("epoch", -1))


def load_test_checkpoint(cfg, model, model_ema=None, model_bucket=None):
    """Reference :452-530."""
    dp = int(getattr(cfg, "NUM_GPUS", 1) or 0) * int(getattr(cfg, "NUM_SHARDS", 1) or 1) > 1
    test_path = getattr(getattr(cfg, "TEST", None), "CHECKPOINT_FILE_PATH", "")
    train_path = getattr(getattr(cfg, "TRAIN", None), "CHECKPOINT_FILE_PATH", "")
    out_dir = getattr(cfg, "OUTPUT_DIR", "")
    if test_path != "" and test_path is not None:
        logger.info("Load from given checkpoint file.\nCheckpoint file path: {}".format(test_path))
        return load_checkpoint(cfg, test_path, model, model_ema, dp)
    if out_dir and has_checkpoint(out_dir):
        last = get_last_checkpoint(out_dir)
        epoch = load_checkpoint(cfg, last, model, model_ema, dp)
        logger.info("Load from the last checkpoint file: {}".format(last))
        return epoch
    if train_path != "" and train_path is not None:
        logger.info("Load from given checkpoint file.\nCheckpoint file path: {}".format(train_path))
        return load_checkpoint(cfg, train_path, model, model_ema, dp)
    logger.info("Unknown way of loading checkpoint. Using with random initialization, only for debugging.")
    return None


def save_checkpoint(path_to_job, model, model_ema=None, optimizer=None, epoch=0, cfg=None):
    """The reference's file layout (:102-143): OUTPUT_DIR/checkpoints/checkpoint_epoch_{:05d}.pyth with 'epoch', 'model_state',
    'cfg'.  Provided so that an evaluation job can round-trip weights; training itself is out of scope."""
    d = get_checkpoint_dir(path_to_job)
    os.makedirs(d, exist_ok=True)
    ms = model.module if hasattr(model, "module") else model
    ckpt = {"epoch": int(epoch), "model_state": ms.state_dict(), "cfg": str(cfg) if cfg is not None else ""}
    if model_ema is not None:
        ckpt["model_ema_state"] = (model_ema.module if hasattr(model_ema, "module") else model_ema).state_dict()
    path = os.path.join(d, "checkpoint_epoch_{:05d}.pyth".format(int(epoch) + 1))
    with open(path, "wb") as f:
        torch.save(ckpt, f)
    return path
