"""Dev tool (GPU box): forced vit_gemm_kernel variants must give bit-identical outputs (same MFMA order, different operand schedule).
usage: python tools/vit_variant_check.py B ref_variant variant [variant ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CFSAR_DEV_LIB"] = "1"
import torch
from clip_fsar_amd import hip
B = int(sys.argv[1]); ref = int(sys.argv[2]); vs = [int(x) for x in sys.argv[3:]]
L = hip.lib()
D = 768
M = 80 * 197 * B + 37                      # ragged last row band
bad = 0
for tag, n, k in [("qkv", 3 * D, D), ("out", D, D), ("fc", 4 * D, D), ("proj", D, 4 * D), ("k128", 2 * D, 128)]:
    torch.manual_seed(1)
    A = torch.randn(M, k, device="cuda").to(torch.bfloat16)
    W = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device="cuda")
    resm = tag in ("out", "proj")
    x0 = torch.randn(M, n, device="cuda").to(torch.float16) if resm else None
    act = hip.ACT_QUICKGELU if tag == "fc" else hip.ACT_NONE
    outs = {}
    for v in [ref] + vs:
        L.cfsar_debug_set_gemm_variant(v, 0)
        out = x0.clone() if resm else torch.zeros(M, n, device="cuda", dtype=torch.bfloat16)
        for rep in range(3):                                  # repeated launches: races show up as run-to-run differences
            o = x0.clone() if resm else torch.zeros(M, n, device="cuda", dtype=torch.bfloat16)
            hip.gemm(A, W, o, bias=bias, residual=o if resm else None, act=act)
            torch.cuda.synchronize()
            if rep and not torch.equal(o.view(torch.int16), out.view(torch.int16)):
                print("  %s variant %d: run-to-run difference" % (tag, v)); bad += 1
            out = o
        outs[v] = out
    r32 = (A.float() @ W.float().t() + bias)
    if act == hip.ACT_QUICKGELU: r32 = r32 * torch.sigmoid(1.702 * r32)
    if resm: r32 = r32 + x0.float()
    for v in vs:
        same = torch.equal(outs[v].view(torch.int16), outs[ref].view(torch.int16))
        err = (outs[v].float() - r32).abs().max().item()
        print("%-5s M=%d N=%d K=%d variant %d vs %d: %s   max |err| vs fp32 %.4f" % (tag, M, n, k, v, ref, "bit-identical" if same else "DIFFERENT", err))
        bad += 0 if same else 1
L.cfsar_debug_set_gemm_variant(0, 0)
print("FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
