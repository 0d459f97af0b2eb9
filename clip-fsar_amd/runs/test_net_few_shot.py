"""test_epoch / test_few_shot with the reference's names and flow (reference runs/test_net_few_shot.py:36-305),
re-designed for sharded inference:

  * the loader yields `EPISODES_PER_STEP` episodes per step for this rank's static shard of the episode list;
  * `model(task_dict)` is the same call as reference :109 (BaseVideoModel -> Identity -> CNN_OTAM_CLIPFSAR);
  * loss / top-1 / top-5 per episode stay on the GPU; nothing is synchronised per episode.  The reference's three
    scalar all-reduces + `.item()` per episode (:168-178) become ONE all-gather of the [episodes, 3] stats matrix
    after the loop (utils.distributed.gather_episode_stats), then the ValMeter is fed in episode order;
  * per-class accuracy tallies (:148-160) are computed from the gathered predictions on the master.
"""
import numpy as np
import torch
import torch.nn.functional as F

from ..datasets.base.builder import build_loader
from ..models.base.builder import build_model
from ..utils import checkpoint as cu
from ..utils import distributed as du
from ..utils import logging
from ..utils import metrics
from ..utils import misc
from ..utils.meters import ValMeter
from ..utils.prefetch import DevicePrefetcher

logger = logging.get_logger(__name__)


@torch.no_grad()
def test_epoch(val_loader, model, val_meter, cur_epoch, cfg, writer=None):
    model.eval()
    val_meter.iter_tic()
    dev = torch.device("cuda", torch.cuda.current_device()) if misc.get_num_gpus(cfg) else torch.device("cpu")
    stats, preds_all, real_all, lab_all = [], [], [], []
    n_local = 0
    batches = val_loader
    if misc.get_num_gpus(cfg):
        # Host -> device ingestion: the reference uploads with `.cuda(non_blocking=True)` from its pinned loader (:59-62).  Here the
        # upload of step i + 1 runs on a copy stream into a second device buffer while step i computes (utils/prefetch.py).  The head
        # checks the label vectors on the host (class ids inside the text table, `way` distinct labels per episode): that happens on the
        # loader's CPU tensors BEFORE the upload, so the forward itself needs no device -> host copy -- a copy there waits for the
        # previous episode's kernels and stalls the launch queue once per step.
        head = getattr(getattr(model, "module", model), "head", None)

        def check_on_host(host):
            if head is not None and hasattr(head, "validate_labels_host") and not host["support_labels"].is_cuda:
                return {"_labels_validated_way": head.validate_labels_host(host["support_labels"], host["real_support_labels"])}
            return None
        # A loader that yields ONE episode per item (the reference's own: TEST.BATCH_SIZE / NUM_GPUS = 1, :57-64) under a config that
        # does not set TEST.EPISODES_PER_STEP: the prefetcher collates k items per model call (k <= 36, fixed per device and config: auto_episodes_per_step); per-episode
        # results do not depend on k.  build_loader's own loaders already come batched (datasets/base/builder.py).
        k = 1
        if not int(getattr(cfg.TEST, "EPISODES_PER_STEP", 0) or 0) and int(getattr(val_loader, "batch_size", 0) or 0) == 1:
            from ..datasets.base.builder import auto_episodes_per_step
            k = auto_episodes_per_step(cfg, len(val_loader))
        batches = DevicePrefetcher(val_loader, dev, pre_upload=check_on_host, collate=k)
    for cur_iter, task_dict in enumerate(batches):
        if n_local >= cfg.TRAIN.NUM_TEST_TASKS:
            break
        model_dict = model(task_dict)
        logits = model_dict["logits"]                                  # [B, Q, way]
        labels = task_dict["target_labels"]
        B, Q, way = logits.shape
        flat, flab = logits.reshape(B * Q, way), labels.reshape(B * Q).long()
        loss = F.cross_entropy(flat, flab, reduction="none").reshape(B, Q).mean(1) / cfg.TRAIN.BATCH_SIZE
        _, idx = torch.topk(flat, min(5, way), dim=1)
        hit = idx.eq(flab.unsqueeze(1))
        top1 = (1.0 - hit[:, :1].any(1).float().reshape(B, Q).mean(1)) * 100.0
        top5 = (1.0 - hit.any(1).float().reshape(B, Q).mean(1)) * 100.0
        stats.append(torch.stack([loss, top1, top5], dim=1))
        preds_all.append(idx[:, 0].reshape(B, Q))
        # copies, not views: the prefetcher's tensors are prefix views of two device buffers that later uploads overwrite
        real_all.append(task_dict["real_target_labels"].reshape(B, Q).clone())
        lab_all.append(labels.reshape(B, Q).clone())
        n_local += B
    total = min(int(cfg.TRAIN.NUM_TEST_TASKS), du.get_world_size() * max(n_local, 0) if du.get_world_size() > 1 else n_local)
    local = torch.cat(stats) if stats else torch.zeros(0, 3, device=dev)
    extra = torch.cat([torch.cat(preds_all).float(), torch.cat(real_all).float(), torch.cat(lab_all).float()], dim=1) \
        if stats else torch.zeros(0, 3, device=dev)
    num_eps = int(getattr(val_loader.dataset, "dataset", val_loader.dataset).__len__()) if du.get_world_size() > 1 else n_local
    num_eps = min(num_eps, int(cfg.TRAIN.NUM_TEST_TASKS))
    allstats = du.gather_episode_stats(torch.cat([local, extra], dim=1), num_eps).cpu()     # the ONE collective
    val_meter.iter_toc()
    Q = (allstats.shape[1] - 3) // 3
    top1_per_class, num_per_class = {}, {}
    for e in range(allstats.shape[0]):
        val_meter.update_stats(float(allstats[e, 1]), float(allstats[e, 2]), 1)
        pred, real, lab = allstats[e, 3:3 + Q], allstats[e, 3 + Q:3 + 2 * Q], allstats[e, 3 + 2 * Q:3 + 3 * Q]
        for qi in range(Q):
            key = str(float(real[qi]))
            num_per_class[key] = num_per_class.get(key, 0) + 1
            top1_per_class[key] = top1_per_class.get(key, 0) + int(pred[qi] == lab[qi])
        val_meter.log_iter_stats(cur_epoch, e)
    epoch_stats = val_meter.log_epoch_stats(cur_epoch)
    if du.is_master_proc():
        for c in sorted(top1_per_class):
            logger.info("class: {}, acc: {}".format(c, top1_per_class[c] / num_per_class[c]))
    acc = 100.0 - allstats[:, 1]
    result = {"episodes": int(allstats.shape[0]), "top1_acc": float(acc.mean()) if len(acc) else float("nan"),
              "top1_acc_ci95": float(1.96 * acc.std(unbiased=False) / max(len(acc), 1) ** 0.5) if len(acc) else float("nan"),
              "loss": float(allstats[:, 0].mean()) if len(acc) else float("nan"), "epoch_stats": epoch_stats,
              "top1_per_class": {c: (top1_per_class[c], num_per_class[c]) for c in sorted(top1_per_class)}}
    val_meter.reset()
    return result


@torch.no_grad()
def eval_epoch(val_loader, model, val_meter, cur_epoch, cfg, writer=None):
    """The validation pass the reference's training script runs between epochs (reference runs/train_net_few_shot.py:279-451,
    few-shot branch :355-420): same inputs, same meter updates and log lines, same per-episode statistics as ``test_epoch``
    (the two loops compute identical quantities there; ``test_epoch`` adds the per-class tallies).  Here it IS ``test_epoch``
    on the sharded loader, returning the epoch's (top1_err, top5_err, loss) so a training driver can track the best epoch
    (reference :452-458 reads ``val_meter.min_top1_err``)."""
    result = test_epoch(val_loader, model, val_meter, cur_epoch, cfg, writer)
    if writer is not None and du.is_master_proc():
        writer.add_scalars({"Val/Top1_err": 100.0 - result["top1_acc"]}, global_step=cur_epoch)
    return result


def test_few_shot(cfg):
    du.init_distributed_training(cfg)
    np.random.seed(cfg.RANDOM_SEED)
    torch.manual_seed(cfg.RANDOM_SEED)
    logging.setup_logging(cfg, getattr(cfg.TEST, "LOG_FILE", None))
    model, model_ema = build_model(cfg)
    cu.load_test_checkpoint(cfg, model, model_ema, None)
    val_loader = build_loader(cfg, "test")
    val_meter = ValMeter(len(val_loader), cfg)
    return test_epoch(val_loader, model, val_meter, 0, cfg, None)
