"""load_test_checkpoint with the reference's priority order (reference utils/checkpoint.py:452-530):
TEST.CHECKPOINT_FILE_PATH -> newest file in OUTPUT_DIR/checkpoints -> TRAIN.CHECKPOINT_FILE_PATH -> random init.
Checkpoints are the reference's ``.pyth`` dicts ({'model_state': ...}); loading is ``strict=False`` (:329)."""
import os

import torch

from . import logging as log_utils

logger = log_utils.get_logger(__name__)


def get_last_checkpoint(output_dir):
    d = os.path.join(output_dir or "", "checkpoints")
    if not os.path.isdir(d):
        return None
    names = sorted(f for f in os.listdir(d) if f.startswith("checkpoint"))
    return os.path.join(d, names[-1]) if names else None


def load_checkpoint(path, model):
    ckpt = torch.load(path, map_location="cpu")
    state = ckpt.get("model_state", ckpt)
    ms = model.module if hasattr(model, "module") else model
    result = ms.load_state_dict(state, strict=False)
    logger.info("loaded %s (missing %d, unexpected %d keys)" % (path, len(result.missing_keys), len(result.unexpected_keys)))
    return result


def load_test_checkpoint(cfg, model, model_ema=None, model_bucket=None):
    test_path = getattr(getattr(cfg, "TEST", None), "CHECKPOINT_FILE_PATH", "")
    if test_path:
        return load_checkpoint(test_path, model)
    last = get_last_checkpoint(getattr(cfg, "OUTPUT_DIR", ""))
    if last:
        return load_checkpoint(last, model)
    train_path = getattr(getattr(cfg, "TRAIN", None), "CHECKPOINT_FILE_PATH", "")
    if train_path:
        return load_checkpoint(train_path, model)
    logger.info("Unknown way of loading checkpoint. Using with random initialization, only for debugging.")
    return None
