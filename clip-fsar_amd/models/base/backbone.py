"""BACKBONE_REGISTRY (reference models/base/backbone.py:17) and the ``Identity`` meta-architecture CLIP-FSAR
configures as its backbone (reference :219-226): the episode dict passes through untouched to the head."""
import torch.nn as nn

from ...utils.registry import Registry

BACKBONE_REGISTRY = Registry("Backbone")


@BACKBONE_REGISTRY.register()
class Identity(nn.Module):
    def __init__(self, cfg):
        super().__init__()

    def forward(self, x):
        return x
