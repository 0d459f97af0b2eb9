"""BaseVideoModel = backbone + head looked up in the registries (reference models/base/models.py:12-67)."""
import torch.nn as nn

from ...utils.registry import Registry
from .backbone import BACKBONE_REGISTRY
from .base_blocks import HEAD_REGISTRY

MODEL_REGISTRY = Registry("Model")


class BaseVideoModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        backbone_cls = BACKBONE_REGISTRY.get(cfg.VIDEO.BACKBONE.META_ARCH)
        head_cls = HEAD_REGISTRY.get(cfg.VIDEO.HEAD.NAME)
        if backbone_cls is None:
            raise KeyError("backbone meta-arch %r is not registered" % cfg.VIDEO.BACKBONE.META_ARCH)
        if head_cls is None:
            raise KeyError("head %r is not registered" % cfg.VIDEO.HEAD.NAME)
        self.backbone = backbone_cls(cfg=cfg)
        self.head = head_cls(cfg=cfg)

    def forward(self, x):
        return self.head(self.backbone(x))

    def train(self, mode=True):
        """Reference :47-67: norm layers stay frozen in train mode when cfg.BN.FREEZE."""
        self.training = mode
        super().train(mode)
        freeze = bool(getattr(getattr(self.cfg, "BN", None), "FREEZE", False))
        if freeze:
            for m in self.modules():
                if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d, nn.LayerNorm)):
                    m.train(False)
        return self
