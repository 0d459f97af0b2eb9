"""Registries the reference defines in models/base/base_blocks.py:19-21."""
from ...utils.registry import Registry

STEM_REGISTRY = Registry("Stem")
BRANCH_REGISTRY = Registry("Branch")
HEAD_REGISTRY = Registry("Head")
