"""build_model(cfg, gpu_id=None) -> (model, model_ema)  (reference models/base/builder.py:19-81).

Same lookup / device-placement / wrapping contract.  Differences, all inference-motivated:
  * EMA copies and SyncBN conversion are training features (out of scope): MODEL.EMA.ENABLE raises;
  * with NUM_GPUS*NUM_SHARDS > 1 the model is NOT wrapped in DistributedDataParallel: episodes shard across
    ranks, every rank builds identical weights, there are no gradients to reduce.
"""
import torch

from .models import BaseVideoModel, MODEL_REGISTRY


def build_model(cfg, gpu_id=None):
    model_cls = MODEL_REGISTRY.get(cfg.MODEL.NAME)
    model = BaseVideoModel(cfg) if model_cls is None else model_cls(cfg)

    if torch.cuda.is_available():
        assert cfg.NUM_GPUS <= torch.cuda.device_count(), "Cannot use more GPU devices than available"
    else:
        assert cfg.NUM_GPUS == 0, "Cuda is not available. Please set `NUM_GPUS: 0 for running on CPUs."

    if cfg.NUM_GPUS:
        cur_device = torch.cuda.current_device() if gpu_id is None else gpu_id
        model = model.cuda(device=cur_device)

    if getattr(getattr(cfg.MODEL, "EMA", None), "ENABLE", False):
        raise NotImplementedError("MODEL.EMA is a training feature; this build is the inference hot path only")
    return model, None
