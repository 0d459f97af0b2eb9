"""Dev tool (GPU box): the RN50 narrow 3x3 convs (stem conv2 / conv3, layer1 conv2) -- direct kernel (csrc/conv.hip) against the
implicit GEMM (csrc/gemm.hip p3 CONV NARROW), interleaved, medians.  usage: python tools/rn_conv_ab.py [B=16]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CFSAR_DEV_LIB"] = "1"
import ctypes, torch
from clip_fsar_amd import hip
L = hip.lib()
L.cfsar_debug_set_direct_conv.argtypes = [ctypes.c_int]; L.cfsar_debug_set_direct_conv.restype = None
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Fn = 80 * B
def bench(fn, rounds=5, iters=4):
    ts = []
    for _ in range(rounds):
        fn(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / iters * 1e3)
    return statistics.median(ts)
for tag, C, Co, H in (("stem conv2", 32, 32, 112), ("stem conv3", 32, 64, 112), ("layer1 conv2", 64, 64, 56)):
    x = torch.randn(Fn * H * H, C, device="cuda").to(torch.bfloat16)
    kpad = -(-9 * C // 64) * 64
    w = (torch.randn(Co, kpad, device="cuda") * (9 * C) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Co, device="cuda")
    out = torch.empty(Fn * H * H, Co, device="cuda", dtype=torch.bfloat16)
    res = {}
    for on in (1, 0, 1, 0):
        L.cfsar_debug_set_direct_conv(on)
        res.setdefault(on, []).append(bench(lambda: hip.conv3x3(x, w, out, Fn, H, H, C, bias=b, relu=True)))
    L.cfsar_debug_set_direct_conv(1)
    M = Fn * H * H
    gb = M * (C + Co) * 2 / 1e9
    fl = 2.0 * M * 9 * C * Co
    td, ti = min(res[1]), min(res[0])
    abl = {}
    for name, bits in (("no stores", 1), ("nt stores", 16), ("no MFMA", 4), ("no barrier", 8), ("no MFMA, no stores", 5), ("only DMA + barrier", 7)):
        L.cfsar_debug_set_direct_conv(1 | (bits << 8))
        abl[name] = bench(lambda: hip.conv3x3(x, w, out, Fn, H, H, C, bias=b, relu=True), rounds=3)
    L.cfsar_debug_set_direct_conv(1)
    print("   ablations (us): " + ", ".join("%s %.0f" % kv for kv in abl.items()))
    print("%-13s direct %7.1f us (%.2f TB/s, %4.0f TFLOP/s) | implicit GEMM %7.1f us | x%.2f" % (tag, td, gb / td * 1e3, fl / td / 1e6, ti, ti / td))
