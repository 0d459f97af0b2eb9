"""clip-fsar_amd: MI355X-native CLIP-FSAR episodic-inference hot path.

Hand-written HIP (gfx950) kernels behind a C-ABI shared library (``csrc/`` ->
``libclipfsar_hip.so``, declared in ``include/clipfsar_hip.h``) and the host-side
Python mirror of the reference's ``models/base`` builder / registry surface.
"""
__version__ = "0.4.0"

# Logits tolerances against the reference's fp32 path on identical frames.  NORTH_STAR_TOLERANCE is BASELINE.json's bound.
# LOGITS_TOLERANCE[mode] is each numerics mode's own bound on the FULL-SIZE configurations (cfg2 / cfg3 / cfg4 / RN50) -- one constant
# for the head's warning, bench.py's `parity` object and tests/test_gpu_e2e.py (per-case bounds of the small cases: tests/_cases.py):
#   fp32 -- the north star; measured <= 7.6e-6;
#   fp16 -- the north star on ALL THREE ViT configurations since round 4 (measured 3.6e-4 / 2.8e-4 / 5.7e-4 on cfg2 / cfg3 / cfg4,
#           profiles/r04_parity_table.md; round 3's fp16 mode met it on cfg2 only: 5.1e-4 / 1.06e-3 / 1.78e-3);
#   bf16 -- the throughput mode's regression bound, about 2 x its measured 3e-3 ... 5.5e-3: it does NOT meet the north star.
NORTH_STAR_TOLERANCE = 1e-3
LOGITS_TOLERANCE = {"fp32": 1e-3, "fp16": 1e-3, "bf16": 1.2e-2}


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
