"""How many episodes one model call should carry on an MI355X (host logic; no reference counterpart: the reference feeds ONE episode per
iteration, runs/test_net_few_shot.py:57-64).

The tower's four block GEMMs run as persistent kernels: one workgroup per CU walks 256 x 256 output tiles (csrc/gemm_vit.hip), so a launch costs
ceil(tiles / 256) rounds whether the last round is full or not.  With M = episodes x frames x tokens rows the tile counts of the four GEMMs are
bands x {3D, D, 4D, D} / 256, bands = ceil(M / 256): at 16 ViT-B/16 episodes (985 bands) the launches are 34.6 / 11.5 / 46.2 / 11.5 rounds and pay
for 35 / 12 / 47 / 12 (97.6 % full); at 18 or 36 episodes (1 109 / 2 217 bands) they are 39.0 / 13.0 / 52.0 / 13.0 rounds or twice that - 99.9 % full.
Measured on one box (profiles/r05_bench_epoch_size.txt): 346.6 (16) -> 349.8 (18) -> 352.4 (36) episodes/s in bf16, 293.9 -> 294.9 -> 303.2 in fp16
(the larger batch also halves the launches per episode)."""
import math

CUS = 256                       # MI355X: 8 XCDs x 32 CUs; the persistent GEMM grid
TILE = 256                      # output tile edge of csrc/gemm_vit.hip at batch scale
FRAME_CAP = 2880                # frames per tower launch the default policy goes up to (36 cfg2 episodes; 7.5 GB of workspace)


def grid_fill(rows: int, width: int, cus: int = CUS) -> float:
    """Useful fraction of the tile rounds the four block GEMMs of a ViT of `width` pay for at M = rows (FLOP-weighted: QKV 3, out_proj 1, c_fc 4,
    c_proj 4 parts of 12)."""
    bands = math.ceil(rows / TILE)
    tot = 0.0
    for n, w in ((3 * width, 3), (width, 1), (4 * width, 4), (width, 4)):
        tn = math.ceil(n / TILE)
        rounds = math.ceil(bands * tn / cus)
        tot += w * (rows / TILE * tn) / (rounds * cus)
    return tot / 12.0


def pick_episodes_per_step(frames_per_episode: int, tokens: int, width: int, max_frames: int = FRAME_CAP, max_episodes: int = 0) -> int:
    """Episodes per model call: the k in (kmax / 2, kmax] that fills the persistent grid's rounds best, kmax = what `max_frames` (and
    `max_episodes`, e.g. the rank's episode count) allows; a fixed per-call cost of a fifth of an episode's time (one episode per call runs at
    0.82 x the batched rate) weighs against small batches, and below 8 episodes the answer is simply kmax.  cfg2 (80 frames, ViT-B/16): 36;
    cfg3 (240 frames): 12; cfg4 (160 frames, ViT-L/14, 32-bit offsets cap it at 2 039 frames): 11."""
    kmax = max(1, max_frames // max(1, frames_per_episode))
    if max_episodes > 0:
        kmax = min(kmax, max_episodes)
    if kmax < 8:
        return kmax
    best, best_score = kmax, -1.0
    for k in range(max(1, (kmax + 1) // 2), kmax + 1):
        score = grid_fill(k * frames_per_episode * tokens, width) * (1.0 - 0.2 / k)
        if score >= best_score:
            best, best_score = k, score
    return best
