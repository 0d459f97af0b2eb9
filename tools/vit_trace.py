"""Dev tool (GPU box): per-tile time stamps of the persistent ViT GEMM (csrc/gemm_vit.hip, CFSAR_TRACE) -- how long a tile's K loop
and epilogue take on each CU and how synchronised the CUs are -- with and without a start-time stagger (dbg bit 128).
usage: python tools/vit_trace.py [B=16] [shape=qkv|out|fc|proj] [variant=26] [stagger units ...]"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CFSAR_DEV_LIB"] = "1"
import torch
from clip_fsar_amd import hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shape = sys.argv[2] if len(sys.argv) > 2 else "qkv"
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 26
staggers = [int(x) for x in sys.argv[4:]] or [0, 12, 23, 46]
XDBG = int(os.environ.get("TRACE_DBG", "0"))          # extra ablation bits (16 = no stores, 4 = no epilogue)
L = hip.lib()
L.cfsar_debug_set_vit_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.cfsar_debug_set_vit_trace.restype = None
D = 768
M = 80 * 197 * B
n, k = {"qkv": (3 * D, D), "out": (D, D), "fc": (4 * D, D), "proj": (D, 4 * D)}[shape]
dev = "cuda"
A = torch.randn(M, k, device=dev).to(torch.bfloat16)
W = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
bias = torch.randn(n, device=dev)
res_mode = shape in ("out", "proj")
out = torch.empty(M, n, device=dev, dtype=torch.float16 if res_mode else torch.bfloat16)
act = hip.ACT_QUICKGELU if shape == "fc" else hip.ACT_NONE
GRID, MAXT = int(os.environ.get("TRACE_GRID", "256")), 64          # TRACE_GRID=512: the two-workgroups-per-CU kernel (variants 36 / 38)
trace = torch.zeros(GRID * MAXT * 4, dtype=torch.int64, device=dev)


def run(dbg, unit, traced):
    L.cfsar_debug_set_gemm_variant(variant, dbg)
    L.cfsar_debug_set_vit_trace(ctypes.c_void_p(trace.data_ptr() if traced else 0), unit)
    for _ in range(2):
        hip.gemm(A, W, out, bias=bias, residual=out if res_mode else None, act=act)
    torch.cuda.synchronize()
    if traced:
        trace.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    iters = 1 if traced else 8
    for _ in range(iters):
        hip.gemm(A, W, out, bias=bias, residual=out if res_mode else None, act=act)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for unit in staggers:
    dbg = (128 if unit else 0) | XDBG
    t_plain = statistics.median(run(dbg, unit, False) for _ in range(5))
    t_tr = run(dbg, unit, True)
    tr = trace.cpu().view(GRID, MAXT, 4)
    ntile = int((tr[:, :, 2] > 0).sum(1).min())
    kl = (tr[:, :ntile, 1] - tr[:, :ntile, 0]).double() * 10.0          # ns (100 MHz)
    ep = (tr[:, :ntile, 2] - tr[:, :ntile, 1]).double() * 10.0
    per = (tr[:, 1:ntile, 0] - tr[:, :ntile - 1, 0]).double() * 10.0
    # synchronisation: spread over the CUs of the K-loop-end stamp of the same tile ordinal, relative to the tile period
    spread = []
    for t in (2, ntile // 2, ntile - 2):
        x = tr[:, t, 1].double() * 10.0
        spread.append(float(x.std()))
    # how many CUs are inside their epilogue at the same time (sampled at the epilogue midpoints of CU 0)
    conc = []
    for t in range(2, ntile - 1):
        mid = (tr[0, t, 1] + tr[0, t, 2]) // 2
        inside = ((tr[:, :ntile, 1] <= mid) & (tr[:, :ntile, 2] >= mid)).any(1).sum()
        conc.append(int(inside))
    st0 = tr[:, 0, 0].double() * 10.0                  # residency: first-tile start stamps of all workgroups (co-resident <=> all within a few us)
    print("   first-tile starts: max - min %.2f us, workgroups starting > 5 us after the first: %d of %d" % (
        float(st0.max() - st0.min()) / 1e3, int((st0 > st0.min() + 5000.0).sum()), GRID))
    if dbg & 32:                                       # half of the workgroups of each XCD skip their stores: compare the two populations
        odd = ((torch.arange(GRID) >> 3) & 1).bool()
        print("   K loop of storing workgroups %.2f us (p90 %.2f), of non-storing workgroups %.2f us (p90 %.2f); epilogue %.2f / %.2f us" % (
            kl[~odd].mean() / 1e3, kl[~odd].flatten().quantile(0.9) / 1e3, kl[odd].mean() / 1e3, kl[odd].flatten().quantile(0.9) / 1e3,
            ep[~odd].mean() / 1e3, ep[odd].mean() / 1e3))
    print("%s variant %d dbg %d stagger unit %3d: launch %7.1f us (traced %7.1f) | tiles/CU %d | tile period %6.2f us | K loop %6.2f us (p10 %5.2f p90 %5.2f) | "
          "epilogue %5.2f us (p10 %5.2f p90 %5.2f) | std of K-loop-end over CUs (tile 2 / mid / last) %5.2f %5.2f %5.2f us | CUs in epilogue with CU 0: median %d"
          % (shape, variant, dbg, unit, t_plain, t_tr, ntile, per.mean() / 1e3, kl.mean() / 1e3, kl.flatten().quantile(0.1) / 1e3, kl.flatten().quantile(0.9) / 1e3,
             ep.mean() / 1e3, ep.flatten().quantile(0.1) / 1e3, ep.flatten().quantile(0.9) / 1e3, spread[0] / 1e3, spread[1] / 1e3, spread[2] / 1e3,
             statistics.median(conc)))
L.cfsar_debug_set_gemm_variant(0, 0)
L.cfsar_debug_set_vit_trace(ctypes.c_void_p(0), 0)
