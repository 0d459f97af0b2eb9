"""Host logic: how many episodes one model call carries (clip_fsar_amd/utils/batching.py).  No reference counterpart -- the reference feeds one
episode per iteration (runs/test_net_few_shot.py:57-64); the rule exists because the tower's GEMMs are persistent 256-workgroup kernels."""
import math

import clip_fsar_amd  # noqa: F401  (registers the package alias)
from clip_fsar_amd.utils.batching import CUS, FRAME_CAP, TILE, grid_fill, pick_episodes_per_step


def test_grid_fill_counts_whole_rounds():
    # 16 cfg2 episodes: 985 bands -> 34.6 / 11.5 / 46.2 / 11.5 rounds paid as 35 / 12 / 47 / 12; 18 and 36 episodes come out (almost) integral
    assert abs(grid_fill(16 * 80 * 197, 768) - 0.9756) < 1e-4
    assert abs(grid_fill(18 * 80 * 197, 768) - 0.9989) < 1e-4
    assert abs(grid_fill(36 * 80 * 197, 768) - 0.9989) < 1e-4
    # by hand for one GEMM-dominated case: a single band of a width-256 "tower" = 3 + 1 + 4 + 1 tiles, each launch one round of 256
    rows = TILE
    expect = (3 * 3 / CUS + 1 * 1 / CUS + 4 * 4 / CUS + 4 * 1 / CUS) / 12
    assert math.isclose(grid_fill(rows, 256), expect)
    assert all(0 < grid_fill(r, 1024) <= 1 for r in (1, 255, 257, 100000, 524143))


def test_pick_for_the_baseline_configs():
    assert FRAME_CAP == 2880
    assert pick_episodes_per_step(80, 197, 768) == 36                          # cfg2: 5-way 1-shot, 8 frames, ViT-B/16
    assert pick_episodes_per_step(240, 197, 768) == 12                         # cfg3: 5-way 5-shot
    assert pick_episodes_per_step(160, 257, 1024, max_frames=2039) == 11       # cfg4: ViT-L/14, 16 frames, the 32-bit offset limit
    # bounded by what the rank has; small counts are taken whole
    assert pick_episodes_per_step(80, 197, 768, max_episodes=20) == 18
    assert [pick_episodes_per_step(80, 197, 768, max_episodes=k) for k in (1, 5, 7)] == [1, 5, 7]
    # never more frames than one launch may carry, never zero
    for fpe in (8, 80, 240, 400, 5000):
        k = pick_episodes_per_step(fpe, 197, 768)
        assert k >= 1 and (k * fpe <= FRAME_CAP or k == 1)
    # the chosen k is at least half of what fits (the rule trades at most a factor two of batch for full rounds)
    for fpe in (40, 80, 120, 160, 240):
        kmax = FRAME_CAP // fpe
        assert pick_episodes_per_step(fpe, 197, 768) > kmax // 2 - 1
