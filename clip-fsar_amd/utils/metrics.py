"""top-k correctness as the reference's harness computes it (reference utils/metrics.py:100-138)."""
import torch


def topks_correct(preds, labels, ks):
    """preds [n, classes], labels [n] -> list of 0-dim float tensors: number of top-k hits for each k in ks."""
    assert preds.size(0) == labels.size(0), "Batch dim of predictions and labels must match"
    _, idx = torch.topk(preds, max(ks), dim=1, largest=True, sorted=True)          # needs classes >= max(ks)
    hit = idx.t().eq(labels.view(1, -1).expand(max(ks), -1))
    return [hit[:k, :].reshape(-1).float().sum() for k in ks]


def topk_errors(preds, labels, ks):
    return [(1.0 - x / preds.size(0)) * 100.0 for x in topks_correct(preds, labels, ks)]
