"""CPU test: the product build's compiler-invisible loads (inline-asm `global_load_dword` in csrc/gemm_vit.hip: the tail operands, the fused
LayerNorm statistics, the per-frame correction) are not read, copied, spilled or overwritten by compiler-generated code before the wait that
covers them.  hipcc treats an asm load's destination as written at the end of the statement (cdna_hip_programming.md 5.7): a spill or copy
placed before the data lands stores a stale register -- silently, on some waves of some launches.  Round 4 found exactly that in a build
WITH packed-fp32 VALU instructions (profiles/r04_fault_audit.md); the product build is clean and this test keeps it so."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_compiler_invisible_loads_are_not_touched_before_their_wait(tmp_path):
    build = _load(os.path.join(ROOT, "clip-fsar_amd", "build.py"), "cfsar_build")
    audit = _load(os.path.join(ROOT, "tools", "asm_load_audit.py"), "cfsar_asm_audit")
    if not os.path.exists(build.HIPCC):
        import pytest
        pytest.skip("no hipcc")
    src = os.path.join(build.CSRC, "gemm_vit.hip")
    out = str(tmp_path / "gemm_vit.s")
    flags = [f for f in build.FLAGS if not f.startswith("-Rpass") and f not in ("-fPIC",)] + build.SOURCE_FLAGS["gemm_vit.hip"]
    cmd = [build.HIPCC] + flags + ["--cuda-device-only", "-S", "-o", out, "-c", src]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    n_loads = open(out).read().count("global_load_dword v")
    assert n_loads > 100, "the audit found no asm loads: has the kernel changed?"
    rc = audit.main(out)
    assert rc == 0, "compiler code touches the destination of a compiler-invisible load before its wait: python tools/asm_load_audit.py <file.s>"
