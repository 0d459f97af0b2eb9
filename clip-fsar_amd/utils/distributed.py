"""torch.distributed helpers with the reference's names (reference utils/distributed.py:19-57,98,262-277).

One process per GPU; backend "nccl" on PyTorch-ROCm is RCCL (xGMI inside a node); "gloo" is used by the CPU tests.
The eval path's traffic is ONE all-gather of the per-episode accuracy vector at the end of the run
(``gather_episode_stats``) instead of the reference's three scalar all-reduces per episode
(reference runs/test_net_few_shot.py:168-171) -- SURVEY.md 8(e).
"""
import os

import torch
import torch.distributed as dist


def is_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_initialized() else 1


def get_rank():
    return dist.get_rank() if is_initialized() else 0


def is_master_proc(num_gpus=8):
    return get_rank() % max(int(num_gpus), 1) == 0 if is_initialized() else True


def init_distributed_training(cfg):
    """Reference :262-277 creates one sub-group per machine; single-node runs need none.  If the launcher
    (torch.distributed.run) exported WORLD_SIZE > 1 and no group exists yet, create it here."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_gpu = torch.cuda.is_available() and int(getattr(cfg, "NUM_GPUS", 1) or 0) > 0
    if use_gpu:
        # one process per GPU: bind this rank to ITS device before any allocation or collective (the reference does it in
        # utils/launcher.py:84-86 `torch.cuda.set_device`); build_model / test_epoch use the current device
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if local_rank >= torch.cuda.device_count():
            raise RuntimeError("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)
    if world > 1 and not is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_gpu:
            backend = getattr(cfg, "DIST_BACKEND", "nccl")
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", torch.cuda.current_device())
                                    if backend == "nccl" else None)
        else:
            dist.init_process_group(backend="gloo")


def all_reduce(tensors, average=True):
    """Reference :41-57: in-place sum (then mean) of every tensor over the world."""
    if get_world_size() == 1:
        return tensors
    for t in tensors:
        dist.all_reduce(t, async_op=False)
    if average:
        w = get_world_size()
        for t in tensors:
            t.mul_(1.0 / w)
    return tensors


def all_gather(tensors):
    """Reference :19-38: gather each tensor from all ranks and concatenate along dim 0."""
    if get_world_size() == 1:
        return tensors
    out = []
    for t in tensors:
        bucket = [torch.ones_like(t) for _ in range(get_world_size())]
        dist.all_gather(bucket, t, async_op=False)
        out.append(torch.cat(bucket, dim=0))
    return out


def shard_episodes(num_episodes, rank=None, world=None):
    """Static partition of the seed-indexed episode list: rank r takes episodes {e : e % world == r}."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    return list(range(rank, num_episodes, world))


def gather_episode_stats(local_stats: torch.Tensor, num_episodes: int):
    """ONE collective: all-gather the per-rank fp32 stats matrix [n_local, k] (rows = this rank's episodes in
    shard_episodes order, padded to ceil(num_episodes / world)) and return them in global episode order
    [num_episodes, k] on every rank."""
    world, rank = get_world_size(), get_rank()
    if local_stats.dim() == 1:
        local_stats = local_stats.unsqueeze(1)
    if world == 1 and not is_initialized():
        return local_stats[:num_episodes]
    # (a one-rank GROUP still issues the collective: the RCCL path is then exercised by the single-GPU tests as well)
    per = (num_episodes + world - 1) // world
    k = local_stats.shape[1]
    pad = torch.zeros(per, k, device=local_stats.device, dtype=torch.float32)
    pad[:local_stats.shape[0]] = local_stats.float()
    out = torch.empty(world * per, k, device=local_stats.device, dtype=torch.float32)
    dist.all_gather_into_tensor(out, pad)
    # rank r's i-th row is global episode r + i*world
    out = out.reshape(world, per, k).transpose(0, 1).reshape(world * per, k)
    return out[:num_episodes]
