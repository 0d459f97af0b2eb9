"""CPU tests around compiler-invisible register loads (inline-asm `global_load_dword` with a VGPR destination).  hipcc treats an asm load's
destination as written at the end of the statement (cdna_hip_programming.md 5.7): a spill or copy placed before the data lands stores a stale
register -- silently, on some waves of some launches.  Round 4 found exactly that in a build WITH packed-fp32 VALU instructions (wrong
lanes / a memory fault, profiles/r04_fault_audit.md) and made the loads visible to the compiler; these tests keep it that way."""
import importlib.util

import pytest
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _compile_and_audit(tmp_path, defs=()):
    build = _load(os.path.join(ROOT, "clip-fsar_amd", "build.py"), "cfsar_build")
    audit = _load(os.path.join(ROOT, "tools", "asm_load_audit.py"), "cfsar_asm_audit")
    if not os.path.exists(build.HIPCC):
        import pytest
        pytest.skip("no hipcc")
    src = os.path.join(build.CSRC, "gemm_vit.hip")
    out = str(tmp_path / "gemm_vit.s")
    flags = [f for f in build.FLAGS if not f.startswith("-Rpass") and f not in ("-fPIC",)] + build.SOURCE_FLAGS["gemm_vit.hip"] + list(defs)
    subprocess.run([build.HIPCC] + flags + ["--cuda-device-only", "-S", "-o", out, "-c", src], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    rc = audit.main(out)
    return rc, audit.main.last_count


def test_product_build_has_no_compiler_invisible_register_loads(tmp_path):
    """Round 4: the tail operands / fused statistics of the ViT GEMM are plain loads the compiler can see (it waits before it spills or reuses
    their destinations); the only inline-asm memory operations left are LDS-DMA (no register destination) and stores."""
    rc, n = _compile_and_audit(tmp_path)
    assert rc == 0 and n == 0, (rc, n)


@pytest.mark.skipif(os.environ.get("CFSAR_AUDIT_HIDDEN_FORM", "0") != "1", reason="A/B form only (CFSAR_AUDIT_HIDDEN_FORM=1): a second 100 s compile")
def test_hidden_load_form_is_clean_where_it_is_still_compiled(tmp_path):
    """-DCFSAR_HIDDEN_TAIL_LOADS (the rounds 2-3 form, kept for A/B): hundreds of hidden loads, none touched by compiler code before its wait
    in the build WITHOUT packed-fp32 instructions -- the property that used to be the product's only protection."""
    rc, n = _compile_and_audit(tmp_path, ["-DCFSAR_HIDDEN_TAIL_LOADS"])
    assert n > 100, n
    assert rc == 0, "compiler code touches the destination of a compiler-invisible load before its wait: python tools/asm_load_audit.py <file.s>"
