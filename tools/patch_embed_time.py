"""The patch embedding (few_shot.py:672-676) as one launch (cfsar_patch_embed: rows gathered from the fp32 frames inside the GEMM) against the
three-launch form (cfsar_im2col_patches + cfsar_gemm with the scattering epilogue + cfsar_cls_rows), interleaved, and the bench leg with and
without it.  usage: python tools/patch_embed_time.py [frames ...]      (default 1280 2880)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from clip_fsar_amd import hip  # noqa: E402

D, dev = 768, "cuda"
for F_ in [int(a) for a in sys.argv[1:]] or [1280, 2880]:
    frames = torch.randn(F_, 3, 224, 224, device=dev)
    w = (torch.randn(D, 768, device=dev) * 768 ** -0.5).to(torch.bfloat16)
    pos, cls = torch.randn(197, D, device=dev) * 0.3, torch.randn(D, device=dev)
    x = torch.empty(F_ * 197, D, device=dev, dtype=torch.float16)
    patches = torch.empty(F_ * 196, 768, device=dev, dtype=torch.bfloat16)

    def fused():
        hip.patch_embed(frames, w, pos, cls, x)

    def three():
        hip.im2col_patches(frames, patches, 16)
        hip.gemm(patches, w, x, residual=pos, M=F_ * 196, N=D, K=768, ldo=D, ldr=D, row_group=196, row_gap=1, row_off=1, res_mod=196, res_off=1)
        hip.cls_rows(x, cls, pos, F_, 197, D)

    t = {"fused": [], "three launches": []}
    for r in range(7):
        for name, fn in (("fused", fused), ("three launches", three)):
            fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                fn()
            e.record()
            torch.cuda.synchronize()
            if r:
                t[name].append(s.elapsed_time(e) / 5 * 1e3)
    fl = 2.0 * F_ * 196 * 768 * D
    for name, v in t.items():
        med = statistics.median(v)
        print("%5d frames  %-15s median %7.1f us  (%5.0f TF/s on the GEMM's FLOPs)  min %7.1f" % (F_, name, med, fl / med / 1e6, min(v)))
    del frames, x, patches
    torch.cuda.empty_cache()

devc = torch.device("cuda:0")
res = {"fused": [], "three launches": []}
for r in range(3):
    for name, vo in (("fused", None), ("three launches", {"fused_patch": False})):
        leg = bench.timed_leg("cfg2", "bf16", 36, 8, devc, None, vit_options=vo)
        res[name].append(leg["value"])
        print(r, name, leg["value"], leg["parity"].get("max_abs_dlogits"), flush=True)
for n, v in res.items():
    print("bench leg (cfg2, bf16, 36 episodes per step) %-15s median %.1f episodes/s  %s" % (n, sorted(v)[len(v) // 2], v))
