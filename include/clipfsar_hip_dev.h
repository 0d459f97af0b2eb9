/*
 * clipfsar_hip_dev.h -- DEVELOPER hooks of libclipfsar_hip.so.  NOT part of the product ABI.
 *
 * These symbols exist only in a library built with `CFSAR_DEV=1 python clip-fsar_amd/build.py` (-DCFSAR_DEV).  The default
 * (product) build contains neither the hooks nor the ablation branches they switch: `nm -D libclipfsar_hip.so` shows the entry
 * points of clipfsar_hip.h only (tests/test_abi.py).  Users: tools/gemm_ab.py and friends (in-process, interleaved A/B of
 * kernel variants on the GPU box).
 */
#ifndef CLIPFSAR_HIP_DEV_H
#define CLIPFSAR_HIP_DEV_H
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* Force the kernel cfsar_gemm / cfsar_gemm_ex dispatch to (variant) and set the ablation bits (dbg) of the next launches;
 * (0, 0) restores the product policy.  Variants: see the dispatch comment in csrc/gemm.hip and csrc/gemm_vit.hip. */
void cfsar_debug_set_gemm_variant(int variant, int dbg);

/* Operand path (0 register-staged, 1 LDS-DMA) and store policy (0 default, 1 nt, 2 sc1) of cfsar_gemm_lnfold /
 * cfsar_gemm_residual_stats; -1 = the product policy. */
void cfsar_debug_set_vit_paths(int opath, int store);
/* ablation bits of cfsar_gemm_lnfold (32 = bf16 MFMA instruction on the fp16 bits: timing A/B only) */
void cfsar_debug_set_vit_dbg(int dbg);
/* per-tile time stamps of the persistent ViT GEMM (device buffer [grid][64][4] of int64: s_memrealtime at tile start, K-loop end,
 * epilogue end; NULL = off) and the unit (x 64 cycles) of the start-time stagger that dbg bit 128 switches on */
void cfsar_debug_set_vit_trace(void* trace, int stagger_unit);
/* workgroup shape of the bf16 attention kernel at 197 tokens (csrc/attention.hip); 0 = product */
void cfsar_debug_set_attn_variant(int v);
/* bit 0 clear: cfsar_conv3x3_nhwc routes Cin, Cout in {32, 64} through the implicit GEMM again instead of the direct kernel
 * (csrc/conv.hip); bits 8+: ablation bits of the direct kernel (1 no stores, 2 first tile's tap addresses, 4 no MFMAs, 8 no barrier) */
void cfsar_debug_set_direct_conv(int on);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* CLIPFSAR_HIP_DEV_H */
