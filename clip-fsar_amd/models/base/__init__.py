"""Import side effects register the hot-path classes, like reference models/base/__init__.py:4-8."""
from .. import module_zoo  # noqa: F401
from . import base_blocks, backbone, models, few_shot  # noqa: F401
