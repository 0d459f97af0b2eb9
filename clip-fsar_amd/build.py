"""Build libclipfsar_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python clip-fsar_amd/build.py [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libclipfsar_hip.so")
SOURCES = ["runtime.hip", "gemm.hip", "gemm_vit.hip", "frame_gemm.hip", "rowops.hip", "attention.hip", "tail.hip", "conv.hip"]
# developer library only: the two alternative GEMM forms measured in round 5 (profiles/r05_gemm_forms.md); the product library does not carry them
DEV_ONLY_SOURCES = ["gemm_vit4.hip", "gemm_vit1w.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# Developer build (`python clip-fsar_amd/build.py --dev`): the same sources with -DCFSAR_DEV -- ablation switches and the
# cfsar_debug_* hooks of include/clipfsar_hip_dev.h -- as a SEPARATE library, libclipfsar_hip_dev.so, which clip_fsar_amd.hip
# loads only when CFSAR_DEV_LIB=1 (tools/*.py set it).  The product library contains none of it; tests/test_abi.py checks that.
DEV_LIB = os.path.join(HERE, "libclipfsar_hip_dev.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage"]      # per-kernel VGPR / scratch report -> build/resource_usage.json
# Every source is compiled WITHOUT packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32).  Round 2: builds of the
# LN-folded GEMM that scaled accumulators with v_pk_mul_f32 occasionally returned stale values in lanes 48-63 of the HIGH register of one
# packed pair (once per ~100 launches, more often with a second kernel on the chip; docs/history/design_r01-r03.md "A fault worth recording").  Neither the
# round-2 microtests nor round 3's tools/ubench/pk_trans_waw.hip (transcendental -> packed WAW / RAW under a transcendental- or
# MFMA-heavy partner wave) reproduce it, so it is FENCED, not explained: with scalar fp32 VALU code it has not been seen (0 of 1 500
# stress launches against 44 of 150), and the fence now covers gemm.hip, attention.hip, tail.hip, conv.hip and rowops.hip as well (the same
# epilogue patterns live there).  Round 4 found one mechanism of this class -- the compiler spilling / reusing the destination of an inline-asm
# load before its data had landed, in exactly such a packed build -- and made those loads compiler-visible (profiles/r04_fault_audit.md);
# the fence stays because it costs nothing.  Same-box A/B: no measurable cost (profiles/r03_gemm_anatomy.md); MI355X_MICROARCH.md lists packed fp32
# beside MFMAs as an anti-lever anyway.  `-DCFSAR_PACKED_FP32` in CFSAR_BUILD_DEFS (developer builds) switches the instructions back on.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
SOURCE_FLAGS = {src: NO_PACKED_FP32 for src in SOURCES + DEV_ONLY_SOURCES}
USAGE = os.path.join(HERE, "build", "resource_usage.json")


def _parse_usage(text: str) -> dict:
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: {vgprs, agprs, scratch, spills, occupancy}}"""
    import re
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r"remark:\s+VGPRs: (\d+)"), ("agprs", r"remark:\s+AGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("spills", r"VGPRs Spill: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return out


def _stale(lib=LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(os.path.dirname(HERE), "include", "clipfsar_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


# `python clip-fsar_amd/build.py --packed`: the PRODUCT sources and flags minus the fence, i.e. WITH packed-fp32 VALU instructions, as
# libclipfsar_hip_packed.so -- the build in which tools/asm_load_audit.py finds a compiler spill of a hidden load's destination before its
# wait (profiles/r04_fault_audit.md).  Loaded only through CFSAR_LIB_PATH (clip_fsar_amd.hip); exists to reproduce that finding on hardware.
PACKED_LIB = os.path.join(HERE, "libclipfsar_hip_packed.so")


def build(force: bool = False, verbose: bool = True, dev: bool = False, packed: bool = False, variant: str = "", defs=()) -> str:
    """variant / defs (developer A/B): the product build with extra -D flags as libclipfsar_hip_<variant>.so (loaded through CFSAR_LIB_PATH)"""
    LIB_OUT = os.path.join(HERE, "libclipfsar_hip_%s.so" % variant) if variant else (PACKED_LIB if packed else (DEV_LIB if dev else LIB))
    if not force and not _stale(LIB_OUT):
        return LIB_OUT
    objs = []
    procs = []
    bdir = os.path.join(HERE, "build", variant or "packed") if (packed or variant) else (os.path.join(HERE, "build", "dev") if dev else os.path.join(HERE, "build"))
    os.makedirs(bdir, exist_ok=True)
    for src in SOURCES + (DEV_ONLY_SOURCES if dev else []):
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        extra = (os.environ.get("CFSAR_BUILD_DEFS", "").split() if dev else []) + list(defs)      # developer A/B builds only
        if packed or "-DCFSAR_PACKED_FP32" in extra:                                # A/B: compile with the packed instructions
            extra = [e for e in extra if e != "-DCFSAR_PACKED_FP32"]
        else:
            extra = extra + SOURCE_FLAGS.get(src, [])
        cmd = [HIPCC] + FLAGS + (["-DCFSAR_DEV"] if dev else []) + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    usage = {}
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
        usage.update(_parse_usage(out))
        rest = "\n".join(l for l in out.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l)
        if verbose and rest.strip():
            print(rest)
    import json
    with open(USAGE if not (dev or packed or variant) else os.path.join(bdir, "resource_usage.json"), "w") as f:
        json.dump(usage, f, indent=0, sort_keys=True)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_OUT


if __name__ == "__main__":
    _v = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    _d = [a for a in sys.argv[1:] if a.startswith("-D")]
    print(build(force="--force" in sys.argv, dev="--dev" in sys.argv, packed="--packed" in sys.argv, variant=_v, defs=_d))
