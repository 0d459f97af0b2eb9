"""Subset of reference utils/misc.py the few-shot test harness exercises."""


def get_num_gpus(cfg):
    """reference utils/misc.py get_num_gpus: GPUs used by this process group (NUM_GPUS, PAI aside)."""
    return int(getattr(cfg, "NUM_GPUS", 0))
