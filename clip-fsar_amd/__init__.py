"""clip-fsar_amd: MI355X-native CLIP-FSAR episodic-inference hot path.

Hand-written HIP (gfx950) kernels behind a C-ABI shared library (``csrc/`` ->
``libclipfsar_hip.so``, declared in ``include/clipfsar_hip.h``) and the host-side
Python mirror of the reference's ``models/base`` builder / registry surface.
"""
__version__ = "0.5.0"

# Logits tolerances against the reference's fp32 path on identical frames.  NORTH_STAR_TOLERANCE is BASELINE.json's bound.
#
# What each numerics mode guarantees on the FULL-SIZE ViT configurations (cfg2 / cfg3 / cfg4), measured against 13-episode reference goldens
# per configuration at the generator's standard contrast (logits spread ~1) AND at high contrast (spread 3-4.5, as trained CLIP features give):
# profiles/r05_parity_table.md, tests/test_gpu_e2e.py::test_modes_against_multi_episode_reference_goldens:
#   fp32 -- a HARD per-episode bound: every logit of every episode within NORTH_STAR_TOLERANCE (measured <= 7.6e-6);
#   fp16 -- a STATISTIC, not a bound: rms <= 4e-4 and 99th percentile <= 1e-3 of |dlogits| over >= 64 logit rows per configuration
#           (measured rms 1.9-3.5e-4, p99 4.6-9.0e-4; ViT-B/16: the same at spread 0.9 and 3.1; ViT-L/14 at spread 4.4 is the upper end).  An
#           episode's LARGEST deviation exceeds 1e-3 in about 1 episode of 13-60 (max seen: FP16_TAIL_MAX_SEEN): LOGITS_TOLERANCE["fp16"] is that tail's
#           regression bound, 1.5e-3.  A caller that needs every episode inside 1e-3 uses "fp32";
#   bf16 -- the throughput mode: rms 2.3-3.9e-3, p99 6.5-8.5e-3, max 1.0e-2 over the same episodes; 1 argmax flip in 325 rows (a near-tie of
#           ViT-L/14 at standard contrast); regression bounds rms 5e-3, p99 1.2e-2, max 1.5e-2.  NOT inside the north star.
# One constant per mode for the head's warning, bench.py's `parity` object and the tests (per-case bounds of the small cases: tests/_cases.py).
NORTH_STAR_TOLERANCE = 1e-3
FP16_TAIL_MAX_SEEN = 1.17e-3          # fp16 mode: the largest |dlogits| of any episode measured so far (profiles/r05_parity_fresh64.txt, 64 fresh cfg2 episodes); ONE place
LOGITS_TOLERANCE = {"fp32": 1e-3, "fp16_strict": 1e-3, "fp16": 1.5e-3, "bf16": 1.5e-2}
LOGITS_STATISTIC = {"fp16_strict": {"rms": 3.2e-4, "p99": 8.5e-4}, "fp16": {"rms": 4e-4, "p99": 1e-3}, "bf16": {"rms": 5e-3, "p99": 1.2e-2}}
# CLIP RN50 tower (N3): three equal error sources and no coherent term to remove (profiles/r04_rn50_fp16.md) -- over 13 reference episodes (8 frames,
# high contrast; profiles/r05_parity_table.md) fp16 mode rms 9.4e-4 / p99 2.7e-3 / max 3.3e-3, bf16 7.5e-3 / 2.1e-2 / 2.5e-2; bounds ~ 1.5-2 x measured.
# "fp32" (6e-6) is RN50's ONLY mode for the 1e-3 contract.
LOGITS_TOLERANCE_RN50 = {"fp32": 1e-3, "fp16": 5e-3, "bf16": 3.5e-2}
LOGITS_STATISTIC_RN50 = {"fp16": {"rms": 1.6e-3, "p99": 5e-3}, "bf16": {"rms": 1.2e-2, "p99": 3.5e-2}}


def install_as_reference_modules():
    """Expose this package's mirror of the reference module surface under the reference's own top-level names
    (``utils.registry``, ``models.base.builder`` ...), so a harness written against the reference -- e.g. its
    ``runs/test_net_few_shot.py`` (imports at :15-29) -- resolves ``from models.base.builder import build_model``
    to this implementation.  See INTEGRATION.md."""
    import importlib
    import sys

    pkg = __name__
    for short in ("utils", "utils.registry", "utils.metrics", "utils.meters", "utils.distributed", "utils.misc",
                  "utils.logging", "utils.checkpoint", "models", "models.module_zoo", "models.base",
                  "models.base.base_blocks", "models.base.backbone", "models.base.models", "models.base.builder",
                  "models.base.few_shot", "datasets", "datasets.base", "datasets.base.builder", "runs",
                  "runs.test_net_few_shot"):
        try:
            sys.modules[short] = importlib.import_module(pkg + "." + short)
        except ModuleNotFoundError:
            pass
