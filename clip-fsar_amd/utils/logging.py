"""Logging helpers with the reference's names (reference utils/logging.py:32-88): stdout logging on the master
process, one JSON line per stats dict."""
import json
import logging
import os
import sys

_FORMAT = "[%(asctime)s][%(levelname)s] %(name)s: %(lineno)4d: %(message)s"


def setup_logging(cfg=None, log_file=None):
    root = logging.getLogger()
    if getattr(setup_logging, "_done", False):
        return
    root.setLevel(logging.INFO)
    rank = int(os.environ.get("RANK", "0"))
    if rank == 0:
        h = logging.StreamHandler(stream=sys.stdout)
        h.setFormatter(logging.Formatter(_FORMAT, datefmt="%m/%d %H:%M:%S"))
        root.addHandler(h)
        out_dir = getattr(cfg, "OUTPUT_DIR", None) if cfg is not None else None
        if out_dir and log_file:
            os.makedirs(out_dir, exist_ok=True)
            fh = logging.FileHandler(os.path.join(out_dir, log_file))
            fh.setFormatter(logging.Formatter(_FORMAT, datefmt="%m/%d %H:%M:%S"))
            root.addHandler(fh)
    else:
        root.addHandler(logging.NullHandler())
    setup_logging._done = True


def get_logger(name):
    return logging.getLogger(name)


def log_json_stats(stats):
    stats = {k: (("%.5f" % v) if isinstance(v, float) else v) for k, v in stats.items()}
    get_logger(__name__).info("json_stats: %s" % json.dumps(stats, sort_keys=True))
