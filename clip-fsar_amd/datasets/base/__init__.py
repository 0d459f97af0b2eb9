from .builder import DATASET_REGISTRY, build_dataset, build_loader  # noqa: F401
from . import ssv2_few_shot  # noqa: F401  (registers Ssv2_few_shot)
