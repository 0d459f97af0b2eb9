"""Dev tool: time the individual HIP kernels on cfg2-shaped problems (GPU box only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_fsar_amd import hip


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    F_, N, D = 80 * B, 197, 768
    M = F_ * N
    dev = "cuda"
    print("frames", F_, "M", M)
    for td, name in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
        for (n, k, tag) in ((2304, 768, "qkv"), (768, 768, "out"), (3072, 768, "fc"), (768, 3072, "proj")):
            if name == "f32" and B > 1:
                continue
            A = torch.randn(M, k, device=dev).to(td)
            W = (torch.randn(n, k, device=dev) * k ** -0.5).to(td)
            bias = torch.randn(n, device=dev)
            out = torch.empty(M, n, device=dev, dtype=td if tag in ("qkv", "fc") else torch.float32)
            res = out if out.dtype == torch.float32 else None
            act = hip.ACT_QUICKGELU if tag == "fc" else hip.ACT_NONE
            us = timeit(lambda: hip.gemm(A, W, out, bias=bias, residual=res, act=act))
            print("gemm %-4s %-4s M=%d N=%d K=%d  %.1f us  %.1f TFLOP/s" % (name, tag, M, n, k, us, 2.0 * M * n * k / us / 1e6))
    x = torch.randn(M, D, device=dev)
    h = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    w = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
    us = timeit(lambda: hip.layernorm(x, h, w, b, M, D))
    print("layernorm f32->bf16 rows=%d  %.1f us  %.1f GB/s" % (M, us, M * D * 6 / us / 1e3))
    xh = x.to(torch.float16)
    us = timeit(lambda: hip.layernorm(xh, h, w, b, M, D))
    print("layernorm f16->bf16 rows=%d  %.1f us  %.1f GB/s" % (M, us, M * D * 4 / us / 1e3))
    qkv = torch.randn(M, 3 * D, device=dev).to(torch.bfloat16)
    o = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    us = timeit(lambda: hip.vit_attention(qkv, o, F_, N, D, 12))
    print("attention bf16 F=%d  %.1f us  %.1f TFLOP/s" % (F_, us, F_ * 12 * 4.0 * N * N * 64 / us / 1e6))
    if B == 1:
        qkv32 = qkv.float(); o32 = torch.empty(M, D, device=dev)
        us = timeit(lambda: hip.vit_attention(qkv32, o32, F_, N, D, 12), iters=3, warm=1)
        print("attention f32 F=%d  %.1f us" % (F_, us))
    frames = torch.randn(F_, 3, 224, 224, device=dev)
    p = torch.empty(F_ * 196, 768, device=dev, dtype=torch.bfloat16)
    us = timeit(lambda: hip.im2col_patches(frames, p, 16))
    print("im2col F=%d  %.1f us  %.1f GB/s" % (F_, us, (frames.numel() * 4 + p.numel() * 2) / us / 1e3))


if __name__ == "__main__":
    main()
