"""ValMeter with the reference's update/log semantics that the few-shot test loop uses
(reference utils/meters.py:676-828): running totals top1_err = sum(err * mb) / sum(mb), windowed medians for the
per-iteration log line, wall-clock iter timer."""
import time
from collections import deque

import numpy as np

from . import logging as log_utils


class ScalarMeter(object):
    def __init__(self, window_size):
        self.deque = deque(maxlen=window_size)
        self.total = 0.0
        self.count = 0

    def reset(self):
        self.deque.clear()
        self.total, self.count = 0.0, 0

    def add_value(self, v):
        self.deque.append(v)
        self.total += v
        self.count += 1

    def get_win_median(self):
        return float(np.median(self.deque))

    def get_global_avg(self):
        return self.total / max(self.count, 1)


class ValMeter(object):
    def __init__(self, max_iter, cfg):
        self._cfg = cfg
        self.max_iter = max_iter
        self.log_period = int(getattr(cfg, "LOG_PERIOD", 50))
        self.mb_top1_err = ScalarMeter(self.log_period)
        self.mb_top5_err = ScalarMeter(self.log_period)
        self.reset()
        self._t0 = time.perf_counter()
        self._dt = 0.0

    def reset(self):
        self.mb_top1_err.reset()
        self.mb_top5_err.reset()
        self.num_top1_mis = 0.0
        self.num_top5_mis = 0.0
        self.num_samples = 0
        self.min_top1_err, self.min_top5_err = 100.0, 100.0
        self.all_preds, self.all_labels = [], []

    def iter_tic(self):
        self._t0 = time.perf_counter()

    def iter_toc(self):
        self._dt = time.perf_counter() - self._t0

    def update_stats(self, top1_err, top5_err, mb_size):
        self.mb_top1_err.add_value(top1_err)
        self.mb_top5_err.add_value(top5_err)
        self.num_top1_mis += top1_err * mb_size
        self.num_top5_mis += top5_err * mb_size
        self.num_samples += mb_size

    def update_predictions(self, preds, labels):
        pass          # the reference keeps them for a TensorBoard plot that is stubbed out (test_net_few_shot.py:283-289)

    def log_iter_stats(self, cur_epoch, cur_iter):
        if (cur_iter + 1) % self.log_period != 0:
            return
        log_utils.log_json_stats({
            "_type": "val_iter", "epoch": "{}".format(cur_epoch + 1), "iter": "{}/{}".format(cur_iter + 1, self.max_iter),
            "time_diff": self._dt, "top1_err": self.mb_top1_err.get_win_median(),
            "top5_err": self.mb_top5_err.get_win_median()})

    def log_epoch_stats(self, cur_epoch):
        top1 = self.num_top1_mis / max(self.num_samples, 1)
        top5 = self.num_top5_mis / max(self.num_samples, 1)
        self.min_top1_err = min(self.min_top1_err, top1)
        self.min_top5_err = min(self.min_top5_err, top5)
        stats = {"_type": "val_epoch", "epoch": "{}".format(cur_epoch + 1), "top1_err": top1, "top5_err": top5,
                 "top1_acc": 100.0 - top1, "min_top1_err": self.min_top1_err, "min_top5_err": self.min_top5_err}
        log_utils.log_json_stats(stats)
        return stats
