"""CPU, world_size 2 over gloo: the N>1 path -- static episode partition, the ONE all-gather of per-episode stats, and
the sharded test loop (runs/test_net_few_shot.test_epoch) with a stand-in model that needs no GPU."""
import os
from types import SimpleNamespace as NS

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _cfg(n):
    return NS(VIDEO=NS(HEAD=NS(NAME="CNN_OTAM_CLIPFSAR", BACKBONE_NAME="ViT-test/16"), BACKBONE=NS(META_ARCH="Identity")),
              TRAIN=NS(CLASS_NAME=["c"] * 64, WAY=5, SHOT=1, QUERY_PER_CLASS=1, NUM_TEST_TASKS=n, BATCH_SIZE=1),
              TEST=NS(CLASS_NAME=["t"] * 24, DATASET="Synthetic_few_shot", EPISODES_PER_STEP=2),
              DATA=NS(NUM_INPUT_FRAMES=2, TEST_CROP_SIZE=32), NUM_GPUS=0, NUM_SHARDS=1, RANDOM_SEED=18, LOG_PERIOD=100)


class _FakeModel(torch.nn.Module):
    """logits that depend only on the episode content: class c gets +1 when the query's label is c, except that
    queries whose real class id is divisible by 3 are mis-predicted -> accuracy is a known function of the episodes."""

    def forward(self, task):
        lab, real = task["target_labels"], task["real_target_labels"]
        B, Q = lab.shape
        logits = torch.zeros(B, Q, 5)
        pred = torch.where(real.long() % 3 == 0, (lab.long() + 1) % 5, lab.long())
        logits.scatter_(2, pred.unsqueeze(-1), 1.0)
        return {"logits": logits, "class_logits": None}


def _expected(n):
    import clip_fsar_amd.synth as synth
    accs = []
    for e in range(n):
        ep = synth.make_episode(5, 1, 1, 2, 32, 24, e, 18)
        accs.append(float(((ep["real_target_labels"].astype(int) % 3) != 0).mean()) * 100.0)
    return accs


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from clip_fsar_amd.utils import distributed as du
    from clip_fsar_amd.utils.meters import ValMeter
    from clip_fsar_amd.datasets.base.builder import build_loader
    from clip_fsar_amd.runs.test_net_few_shot import test_epoch
    # 1. partition
    mine = du.shard_episodes(n)
    assert mine == list(range(rank, n, world))
    # 2. the single collective restores global episode order on every rank
    local = torch.tensor([[float(e), float(e) * 2] for e in mine])
    allst = du.gather_episode_stats(local, n)
    assert allst.shape == (n, 2) and torch.equal(allst[:, 0], torch.arange(n).float())
    # 3. sharded test loop
    cfg = _cfg(n)
    loader = build_loader(cfg, "test")
    res = test_epoch(loader, _FakeModel(), ValMeter(len(loader), cfg), 0, cfg)
    q.put((rank, res["episodes"], res["top1_acc"]))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    n, world = 7, 2                      # odd count: ranks get 4 and 3 episodes (ragged shard + padding)
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = _expected(n)
    for rank, episodes, acc in out:
        assert episodes == n
        assert abs(acc - sum(exp) / n) < 1e-4, (rank, acc, exp)


def test_world_size_8_gloo_ragged():
    """Eight ranks (the node the reference's launcher fills, utils/launcher.py:29-34) with an episode count that does not divide:
    19 episodes -> three ranks hold 3, five hold 2; EPISODES_PER_STEP = 2 makes the last step of the 3-episode ranks ragged too."""
    n, world = 19, 8
    port = 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp = _expected(n)
    assert sorted(r for r, _, _ in out) == list(range(world))
    for rank, episodes, acc in out:
        assert episodes == n
        assert abs(acc - sum(exp) / n) < 1e-4, (rank, acc, exp)


def test_single_process_matches():
    from clip_fsar_amd.utils.meters import ValMeter
    from clip_fsar_amd.datasets.base.builder import build_loader
    from clip_fsar_amd.runs.test_net_few_shot import test_epoch
    n = 5
    cfg = _cfg(n)
    loader = build_loader(cfg, "test")
    res = test_epoch(loader, _FakeModel(), ValMeter(len(loader), cfg), 0, cfg)
    exp = _expected(n)
    assert res["episodes"] == n and abs(res["top1_acc"] - sum(exp) / n) < 1e-4
