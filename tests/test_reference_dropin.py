"""CPU, dev container only (skipped where /root/reference is absent): INTEGRATION.md option A -- our head registered
in the REAL reference's HEAD_REGISTRY is what the reference's own build_model(cfg) constructs, and a state dict saved
from the reference's head loads into it key-for-key."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/models/base"), reason="reference tree not present")
def test_reference_build_model_constructs_hip_head():
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import torch
        import ref_harness as rh
        import clip_fsar_amd.synth as synth
        fs = rh.import_reference()                                   # real reference modules: models.*, utils.*
        from models.base.base_blocks import HEAD_REGISTRY            # the reference's registry
        from models.base.builder import build_model                  # the reference's builder
        import clip_fsar_amd.models.base.few_shot as hip_fs
        arch = "ViT-test/16"; a = synth.ARCHS[arch]
        cfg = rh.make_cfg(arch)
        # reference head (for its state dict), built through the harness
        sd = synth.head_state_dict(arch)
        ref_head = rh.build_reference_head(cfg, a, sd, synth.text_features(64, a["embed"], "train"),
                                           synth.text_features(24, a["embed"], "test"))
        HEAD_REGISTRY._entry_map["CNN_OTAM_CLIPFSAR"] = hip_fs.CNN_OTAM_CLIPFSAR
        model, ema = build_model(cfg)                                # reference code path, NUM_GPUS = 0
        assert type(model).__module__ == "models.base.models", type(model).__module__
        assert isinstance(model.head, hip_fs.CNN_OTAM_CLIPFSAR)
        res = model.head.load_state_dict(ref_head.state_dict(), strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        w = dict(model.named_parameters())["head.backbone.transformer.resblocks.1.attn.in_proj_weight"]
        assert torch.equal(w, ref_head.state_dict()["backbone.transformer.resblocks.1.attn.in_proj_weight"])
        print("DROPIN_OK")
    """) % (ROOT, os.path.join(ROOT, "oracle"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "DROPIN_OK" in out.stdout, out.stdout + out.stderr
