// out = act(A . W^T + bias) + residual  on the CDNA4 matrix cores.
//
// Replaces nn.Linear / MultiheadAttention.in_proj,out_proj / conv1-as-GEMM / `@ proj` of the reference
// (few_shot.py:623,626-628,635,672,686,1046-1053,1646-1650).
//
// Structure (gfx950):
//   * 128(M) x 128(N) output tile per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles);
//   * K streamed in 128-byte slices per row (64 bf16 / 32 f32) through a 2-stage LDS ring filled by
//     LDS-DMA (`global_load_lds_dwordx4`, 16 B per lane, no VGPR round trip); one barrier per K-slice;
//   * LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row>>1)&7 so that the
//     `ds_read_b128` fragment reads (32 rows x one chunk) are bank-conflict free.  Because LDS-DMA writes
//     lane-linearly, the swizzle is applied to the per-lane *global source* address and to the read address;
//   * operands are swapped (W feeds MFMA "A", activations feed "B"), so a lane ends up holding 4 consecutive
//     output columns of ONE token row -> 8/16-byte epilogue stores and float4 bias/residual loads;
//   * bf16 inputs: v_mfma_f32_32x32x16_bf16; f32 inputs: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, the
//     validation mode); fp32 accumulation in both;
//   * workgroup id -> tile mapping is XCD-aware (bijective remap: each XCD's L2 sees a contiguous band of tiles).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_vit.h"
#ifdef CFSAR_DEV
#include "../../include/clipfsar_hip_dev.h"
#endif

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int ROWB = 128;                       // bytes of K per tile row per stage
constexpr int STAGE_BYTES = (BM + BN) * ROWB;   // 32 KiB
constexpr int NTHREADS = 256;

// Ablation switches exist only in dev builds (`CFSAR_DEV=1 python clip-fsar_amd/build.py`, include/clipfsar_hip_dev.h); in the
// product build GDBG() is the constant 0 and every branch on it folds away.
#ifdef CFSAR_DEV
#define GDBG(p) ((p).dbg)
#else
#define GDBG(p) 0
#endif

struct GemmArgs {
    const char* A;
    const char* W;
    void* out;
    const float* bias;
    const void* res;      // residual: fp32 (res_kind 0), bf16 (1) or fp16 (2)
    int res_kind;
    int relu;             // ReLU applied last (after bias / activation / residual): conv+BN(+identity)+ReLU of the RN50 tower
    int M, N, K;
    int lda, ldw, ldo, ldr;
    int act;
    int row_group, row_gap, row_off, res_mod, res_off;
    int tiles_n;
    int ntiles;    // persistent p6: total number of 256x256 tiles
#ifdef CFSAR_DEV
    int dbg;   // ablation bits (dev builds only, cfsar_debug_set_gemm_variant): 1 = no in-loop DMA, 2 = no in-loop barrier, 4 = no epilogue
#endif
    int conv_H, conv_W, conv_lgC;   // implicit 3x3 / pad 1 / stride 1 conv (p10 CONV): A = NHWC input [F,H,W,C], C = 1 << conv_lgC
};

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ void glds16(const char* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == CFSAR_ACT_QUICKGELU)     // x * sigmoid(1.702 x); v_exp_f32 + v_rcp_f32 (1 ulp each)
        return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * v));
    if (act == CFSAR_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}

__device__ __forceinline__ float4 load_res4(const void* res, int kind, size_t elem_off) {
    if (kind == 1) {
        const bf16x4 r = *reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(res) + elem_off);
        return make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
    }
    if (kind == 2) {
        const f16x4 r = *reinterpret_cast<const f16x4*>(static_cast<const _Float16*>(res) + elem_off);
        return make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
    }
    return *reinterpret_cast<const float4*>(static_cast<const float*>(res) + elem_off);
}

// ---- epilogue: D[n][m] layout -> lane owns token row m = mbase+32mi+(lane&31) and columns 8g+4hi..+3 of each
// 32-column tile; bias / activation / residual fused; 8-byte (bf16) or 16-byte (f32) stores.
template <typename TO>
__device__ __forceinline__ void epilogue(f32x16 (&acc)[2][2], const GemmArgs& p, int mbase, int nbase, int lane) {
    const int lr = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = mbase + mi * 32 + lr;
        if (m >= p.M) continue;
        int orow = m + p.row_off;
        if (p.row_group > 0) orow += (m / p.row_group) * p.row_gap;
        const int rrow = p.res_mod > 0 ? (m % p.res_mod) + p.res_off : orow;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nbase + ni * 32 + 8 * g + 4 * hi;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mi][ni][4 * g + j];
                if (p.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (p.act != CFSAR_ACT_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], p.act);
                }
                if (p.res) {
                    const float4 rv = load_res4(p.res, p.res_kind, (size_t)rrow * p.ldr + n);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
                if (p.relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
                }
                if constexpr (sizeof(TO) == 2) {
                    typename Vec2B<TO>::v4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (TO)v[j];
                    *reinterpret_cast<typename Vec2B<TO>::v4*>(reinterpret_cast<TO*>(p.out) + (size_t)orow * p.ldo + n) = o;
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)orow * p.ldo + n) =
                        make_float4(v[0], v[1], v[2], v[3]);
                }
            }
    }
}

// ---- one 128-byte K slice of a wave's 64x64 sub-tile: 16 ds_read_b128 + 16 (bf16) / 64 (f32) MFMAs
template <typename TI>
__device__ __forceinline__ void mma_slice(f32x16 (&acc)[2][2], const char* sX, const char* sW, const int (&offX)[2],
                                          const int (&offW)[2], const int (&sxX)[2], const int (&sxW)[2], int hi) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 2 * s + hi;
        uint4 xf[2], wf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            xf[i] = *reinterpret_cast<const uint4*>(sX + offX[i] + ((c ^ sxX[i]) << 4));
            wf[i] = *reinterpret_cast<const uint4*>(sW + offW[i] + ((c ^ sxW[i]) << 4));
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if constexpr (sizeof(TI) == 2) {
                    acc[mi][ni] = cfsar_mfma_32x32x16<TI>(wf[ni], xf[mi], acc[mi][ni]);
                } else {
                    const f32x4 a = __builtin_bit_cast(f32x4, wf[ni]);
                    const f32x4 bb = __builtin_bit_cast(f32x4, xf[mi]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bb[j], acc[mi][ni], 0, 0, 0);
                }
            }
    }
}


// ---- epilogue through LDS (p3 kernel): each wave transposes its 64x64 fp32 sub-tile through a private LDS region so
// that global traffic is whole contiguous rows: a lane ends up with 4 consecutive columns of a row, 16 lanes cover one
// 64-column row segment (128 B bf16 / 256 B f32 stores, 256 B residual loads) instead of 32 rows x 16 B per instruction.
constexpr int EPI_RS = 272;                     // staged row: 64 fp32 + 16 B pad (conflict-free b128 writes)
constexpr int EPI_WAVE_BYTES = 64 * EPI_RS;     // 17408 B per wave
// bf16 output, 16 bytes per lane: after the same fp32 LDS transpose a lane owns EIGHT consecutive columns of a row, so the
// residual load (bf16: one dwordx4) and the store (one dwordx4) move whole 128-byte row segments per 8 lanes, 8 rows per
// wave-instruction -- half the VMEM instructions of the 4-column form.  The RN50 bottleneck tails (K = 64...512, bf16
// residual + ReLU) are bound by exactly those instructions.
template <typename T2, int ACT, bool HAS_RES, bool FULL, int NMI>
__device__ __forceinline__ void epilogue_lds_x8(f32x16 (*acc)[2], const GemmArgs& p, int mbase, int nbase, int lane,
                                                char* wbuf) {
    typedef typename Vec2B<T2>::v8 T2x8;
    typedef typename Vec2B<T2>::v4 T2x4;
    const int lr = lane & 31, hi = lane >> 5;
    const int rsub = lane >> 3, cc = lane & 7;
    const int M = p.M, N = p.N, ldo = p.ldo, ldr = p.ldr, row_off = p.row_off;
    const int n = nbase + cc * 8;
    const int nc = n + 8 <= N ? n : N - 8;                   // clamped column for loads; stores are predicated
    const int nvalid = FULL ? 8 : (n + 8 <= N ? 8 : (n + 4 <= N ? 4 : 0));
    const void* resp = p.res;
    const int res_kind = p.res_kind, relu = p.relu;
    T2* outp = reinterpret_cast<T2*>(p.out);
    constexpr int NIT = 4 * NMI;
    uint4 rb[HAS_RES ? NIT : 1];                              // 2-byte residual: 8 values; fp32 residual: loaded in the row loop
    size_t ooff[NIT], roff[HAS_RES ? NIT : 1];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int m = mbase + it * 8 + rsub;
        const int mc = m < M ? m : M - 1;
        ooff[it] = (size_t)(mc + row_off) * ldo + nc;
        if constexpr (HAS_RES) {
            roff[it] = (size_t)(mc + row_off) * ldr + nc;
            if (res_kind) rb[it] = *reinterpret_cast<const uint4*>(static_cast<const char*>(resp) + roff[it] * 2);
        }
    }
    float4 bv0 = make_float4(0.f, 0.f, 0.f, 0.f), bv1 = bv0;
    if (p.bias) {
        bv0 = *reinterpret_cast<const float4*>(p.bias + nc);
        bv1 = *reinterpret_cast<const float4*>(p.bias + nc + 4);
    }
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(wbuf + (mi * 32 + lr) * EPI_RS + (ni * 32 + 8 * g + 4 * hi) * 4) =
                    make_float4(acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = it * 8 + rsub;
        // the staged row holds the tile's columns [nbase, nbase+64); a clamped (ragged) lane re-reads valid columns
        const int cl = nc - nbase;
        const float4 a0 = *reinterpret_cast<const float4*>(wbuf + row * EPI_RS + cl * 4);
        const float4 a1 = *reinterpret_cast<const float4*>(wbuf + row * EPI_RS + cl * 4 + 16);
        float v[8] = {a0.x + bv0.x, a0.y + bv0.y, a0.z + bv0.z, a0.w + bv0.w, a1.x + bv1.x, a1.y + bv1.y, a1.z + bv1.z, a1.w + bv1.w};
        if constexpr (ACT != CFSAR_ACT_NONE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j], ACT);
        }
        if constexpr (HAS_RES) {
            if (res_kind == 1) {
                const bf16x8 r8 = __builtin_bit_cast(bf16x8, rb[it]);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)r8[j];
            } else if (res_kind == 2) {
                const f16x8 r8 = __builtin_bit_cast(f16x8, rb[it]);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += (float)r8[j];
            } else {
                const float4 r0 = *reinterpret_cast<const float4*>(static_cast<const float*>(resp) + roff[it]);
                const float4 r1 = *reinterpret_cast<const float4*>(static_cast<const float*>(resp) + roff[it] + 4);
                v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
            }
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.0f);
        }
        T2x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T2)v[j];
        if (FULL || (nvalid == 8 && mbase + row < M)) {
            *reinterpret_cast<T2x8*>(outp + ooff[it]) = o;
        } else if (nvalid == 4 && mbase + row < M) {          // ragged right edge (N % 8 == 4): the clamped lane owns columns
            T2x4 o4;                                          // [N-8, N); its last four are the tile's last four
#pragma unroll
            for (int j = 0; j < 4; ++j) o4[j] = o[4 + j];
            *reinterpret_cast<T2x4*>(outp + ooff[it] + 4) = o4;
        }
    }
}

// X8 = false: the one-wave-per-SIMD kernel (p10) holds 256 accumulator registers through the epilogue and spills with
// the wider form
template <typename TO, int ACT, bool HAS_RES, bool REMAP, bool FULL, int NMI = 2, bool X8 = true>
__device__ __forceinline__ void epilogue_lds(f32x16 (*acc)[2], const GemmArgs& p, int mbase, int nbase, int lane,
                                             char* wbuf) {
    if constexpr (sizeof(TO) == 2 && !REMAP && X8) {
        if (!(GDBG(p) & 128) && (p.ldo & 7) == 0 && (!HAS_RES || (p.ldr & 7) == 0) && p.N >= 8) {
            epilogue_lds_x8<TO, ACT, HAS_RES, FULL, NMI>(acc, p, mbase, nbase, lane, wbuf);
            return;
        }
    }
    const int lr = lane & 31, hi = lane >> 5;
    const int rsub = lane >> 4, cc = lane & 15;
    const int M = p.M, N = p.N, ldo = p.ldo, ldr = p.ldr, row_off = p.row_off;
    const int n = nbase + cc * 4;
    const bool nvalid = n < N;
    const int nc = nvalid ? n : N - 4;                       // clamped column: loads stay in bounds, stores are predicated
    const void* resp = p.res;
    const int res_kind = p.res_kind, relu = p.relu;
    TO* outp = reinterpret_cast<TO*>(p.out);
    // (1) issue every residual load of this lane up front (16 x 16 B, whole 256-byte row segments per 16 lanes)
    float4 rv[8 * NMI];
    size_t ooff[8 * NMI];
#pragma unroll
    for (int it = 0; it < 8 * NMI; ++it) {
        const int m = mbase + it * 4 + rsub;
        const int mc = m < M ? m : M - 1;
        int orow = mc + row_off;
        int rrow = orow;
        if constexpr (REMAP) {
            if (p.row_group > 0) orow += (mc / p.row_group) * p.row_gap;
            rrow = p.res_mod > 0 ? (mc % p.res_mod) + p.res_off : orow;
        }
        ooff[it] = (size_t)orow * ldo + nc;
        if constexpr (HAS_RES) rv[it] = load_res4(resp, res_kind, (size_t)rrow * ldr + nc);
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + nc);
    // (2) transpose the accumulators through the wave-private LDS region
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(wbuf + (mi * 32 + lr) * EPI_RS + (ni * 32 + 8 * g + 4 * hi) * 4) =
                    make_float4(acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private region: in-order DS + this wait suffice
    // (3) row-contiguous epilogue math + stores
#pragma unroll
    for (int it = 0; it < 8 * NMI; ++it) {
        const int row = it * 4 + rsub;
        const float4 a = *reinterpret_cast<const float4*>(wbuf + row * EPI_RS + cc * 16);
        float v[4] = {a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w};
        if constexpr (ACT != CFSAR_ACT_NONE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], ACT);
        }
        if constexpr (HAS_RES) {
            v[0] += rv[it].x; v[1] += rv[it].y; v[2] += rv[it].z; v[3] += rv[it].w;
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
        }
        if (FULL || (nvalid && mbase + row < M)) {       // FULL: straight-line code, counted vmcnt waits
            if constexpr (sizeof(TO) == 2) {
                typename Vec2B<TO>::v4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (TO)v[j];
                *reinterpret_cast<typename Vec2B<TO>::v4*>(outp + ooff[it]) = o;
            } else {
                *reinterpret_cast<float4*>(outp + ooff[it]) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// ---- bf16-output epilogue without residual (QKV, c_fc): bias + activation are applied in registers (the lane owns 4
// consecutive columns), the result is packed to bf16 BEFORE the LDS transpose (half the LDS traffic), and each lane
// then stores 16 bytes: 8 lanes cover one 128-byte row segment, 8 rows per wave-instruction.
// Measured on MI355X (r01): not faster than the fp32-staged path (QKV 326 vs 322 us, c_fc 434 vs 415 us at M = 63040):
// the epilogue is bound by the store drain, not by LDS traffic.  Kept for the next round's persistent-tile kernel.
constexpr bool kBf16PackedEpilogue = false;
constexpr int EPI_RS16 = 144;                    // staged row: 64 bf16 + 16 B pad (keeps ds_read_b128 16-byte aligned)
template <int ACT, bool FULL>
__device__ __forceinline__ void epilogue_lds_bf16(f32x16 (*acc)[2], const GemmArgs& p, int mbase, int nbase, int lane,
                                                  char* wbuf) {
    const int lr = lane & 31, hi = lane >> 5;
    const int M = p.M, N = p.N, ldo = p.ldo;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = ni * 32 + 8 * g + 4 * hi;                 // column inside the 64-wide wave tile
            int n = nbase + cl;
            n = n < N ? n : N - 4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                float v[4] = {acc[mi][ni][4 * g] + bv.x, acc[mi][ni][4 * g + 1] + bv.y, acc[mi][ni][4 * g + 2] + bv.z,
                              acc[mi][ni][4 * g + 3] + bv.w};
                if constexpr (ACT != CFSAR_ACT_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], ACT);
                }
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
                *reinterpret_cast<bf16x4*>(wbuf + (mi * 32 + lr) * EPI_RS16 + cl * 2) = o;
            }
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int rsub = lane >> 3, cc = lane & 7;
    const int n = nbase + cc * 8;
    __bf16* outp = reinterpret_cast<__bf16*>(p.out);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + rsub;
        const int m = mbase + row;
        const uint4 a = *reinterpret_cast<const uint4*>(wbuf + row * EPI_RS16 + cc * 16);
        if (FULL || (m < M && n + 8 <= N))
            *reinterpret_cast<uint4*>(outp + (size_t)(m + p.row_off) * ldo + n) = a;
        else if (m < M) {                                            // ragged right edge (N % 8 == 4)
            const unsigned w[4] = {a.x, a.y, a.z, a.w};
            for (int j = 0; j < 8; ++j)
                if (n + j < N) reinterpret_cast<unsigned short*>(outp)[(size_t)(m + p.row_off) * ldo + n + j] =
                    (unsigned short)(w[j >> 1] >> ((j & 1) * 16));
        }
    }
}

// ---- K slice with register double-buffered fragments: the ds_read_b128 of sub-step s+1 are in flight while the
// MFMAs of sub-step s execute.  MI = 32-row tiles of the wave in M (2: 64x64 wave tile, 1: 32x64).
template <typename TI, int MI = 2>
__device__ __forceinline__ void mma_slice_db(f32x16 (&acc)[MI][2], const char* sX, const char* sW, const int (&offX)[MI],
                                             const int (&offW)[2], const int (&sxX)[MI], const int (&sxW)[2], int hi) {
    uint4 xf[2][MI], wf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i < MI) xf[0][i] = *reinterpret_cast<const uint4*>(sX + offX[i] + ((hi ^ sxX[i]) << 4));
        wf[0][i] = *reinterpret_cast<const uint4*>(sW + offW[i] + ((hi ^ sxW[i]) << 4));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s < 3) {
            const int c = 2 * (s + 1) + hi;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (i < MI) xf[nxt][i] = *reinterpret_cast<const uint4*>(sX + offX[i] + ((c ^ sxX[i]) << 4));
                wf[nxt][i] = *reinterpret_cast<const uint4*>(sW + offW[i] + ((c ^ sxW[i]) << 4));
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the next sub-step's reads AHEAD of this sub-step's MFMAs
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if constexpr (sizeof(TI) == 2) {
                    acc[mi][ni] = cfsar_mfma_32x32x16<TI>(wf[cur][ni], xf[cur][mi], acc[mi][ni]);
                } else {
                    const f32x4 a = __builtin_bit_cast(f32x4, wf[cur][ni]);
                    const f32x4 bb = __builtin_bit_cast(f32x4, xf[cur][mi]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bb[j], acc[mi][ni], 0, 0, 0);
                }
            }
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(GemmArgs p) {
    // Two separate LDS objects (not one array): hipcc tracks in-flight LDS-DMA per LDS object, so fragment reads of
    // one stage do not wait (vmcnt(0)) for the DMA that is filling the other stage.
    __shared__ __attribute__((aligned(16))) char stage0[STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) char stage1[STAGE_BYTES];
    constexpr bool kBf16 = sizeof(TI) == 2;
    constexpr int BK = ROWB / (int)sizeof(TI);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- XCD-aware bijective block remap (blocks b, b+8, ... share an XCD / L2)
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int tm = lin / p.tiles_n, tn = lin - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- LDS-DMA staging geometry: instruction i of this wave fills rows (4i+wave)*8 .. +8 of each operand tile
    const char* srcX[4];
    const char* srcW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);          // logical chunk that lands in physical slot lane&7
        int gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        srcX[i] = p.A + ((size_t)gm * p.lda) * sizeof(TI) + chunk * 16;
        srcW[i] = p.W + ((size_t)gn * p.ldw) * sizeof(TI) + chunk * 16;
    }
    auto stage_load = [&](char* stage, int kt) {
        char* dst = stage + wave * 1024;
        const size_t koff = (size_t)kt * ROWB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(srcX[i] + koff, dst + i * 4096);
            glds16(srcW[i] + koff, dst + BM * ROWB + i * 4096);
        }
    };

    // ---- MFMA fragment geometry
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, hi = lane >> 5;
    int offX[2], offW[2], sxX[2], sxW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rx = wm * 64 + i * 32 + lr;
        const int rw = wn * 64 + i * 32 + lr;
        offX[i] = rx * ROWB;
        sxX[i] = swz(rx);
        offW[i] = BM * ROWB + rw * ROWB;
        sxW[i] = swz(rw);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    auto compute = [&](const char* base) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c = 2 * s + hi;
            uint4 xf[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                xf[i] = *reinterpret_cast<const uint4*>(base + offX[i] + ((c ^ sxX[i]) << 4));
                wf[i] = *reinterpret_cast<const uint4*>(base + offW[i] + ((c ^ sxW[i]) << 4));
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if constexpr (kBf16) {
                        acc[mi][ni] = cfsar_mfma_32x32x16<TI>(wf[ni], xf[mi], acc[mi][ni]);
                    } else {
                        const f32x4 a = __builtin_bit_cast(f32x4, wf[ni]);
                        const f32x4 bb = __builtin_bit_cast(f32x4, xf[mi]);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bb[j], acc[mi][ni], 0, 0, 0);
                    }
                }
        }
    };

    // ---- main loop: 2-stage ring, one barrier per K slice
    const int nk = p.K / BK;
    stage_load(stage0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0;;) {
        if (kt + 1 < nk) stage_load(stage1, kt + 1);     // prefetch next slice, then compute this one
        __builtin_amdgcn_sched_barrier(0);
        compute(stage0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (++kt >= nk) break;
        if (kt + 1 < nk) stage_load(stage0, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(stage1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (++kt >= nk) break;
    }

    epilogue<TO>(acc, p, m0 + wm * 64, n0 + wn * 64, lane);
}

template <typename TI, typename TO>
int launch(const GemmArgs& a, hipStream_t s) {
    const int tiles_m = (a.M + BM - 1) / BM;
    hipLaunchKernelGGL((gemm_kernel<TI, TO>), dim3(tiles_m * a.tiles_n), dim3(NTHREADS), 0, s, a);
    return cfsar_check_launch("cfsar_gemm");
}


// ============================================================================================================
// v2 ("p3"): 256(M) x 128(N) tile, 512 threads (8 waves as 4x2, each 64x64), THREE-stage LDS ring (144 KiB) with the
// LDS-DMA issued from inline asm, so hipcc does not see it: no compiler-inserted vmcnt(0) at the barrier or before
// fragment reads.  Loads run two K-slices ahead and stay in flight ACROSS the (single) barrier per slice; each wave
// drains only the slice it is about to read with a counted `s_waitcnt vmcnt(6)`.
// ============================================================================================================
constexpr int BM2 = 256;
constexpr int BN2 = 128;
constexpr int STAGE2 = (BM2 + BN2) * ROWB;   // 48 KiB
constexpr int NSTAGE2 = 3;
constexpr int NTHREADS2 = 512;

// LDS-DMA, 16 B per lane: LDS[m0 + lane*16 .. +16] = *(src).  M0 is compiler-reserved: save/restore inside the statement.
__device__ __forceinline__ void glds16_asm(const char* src, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_addr)
        : "memory");
}

// CONV (bf16): the X operand is gathered on the fly from an NHWC tensor (implicit GEMM for nn.Conv2d(3, padding=1), see
// gemm_kernel_p10): per K slice every lane picks the pixel of its chunk's tap, or a 16-byte block of zeros outside the image
// (LDS-DMA takes per-lane source addresses).  This is the conv kernel for Cout <= 128 (256x128 tile; p10 for wider ones).
__device__ const uint4 g_zero_chunk[2] = {};
// NARROW: 256 x 64 tile (8 waves stacked in M, each 32 x 64) for Cout <= 64: the RN50 stem / layer1 convs are bound by the
// MFMA work wasted on padding columns, not by bytes.
template <typename TI, typename TO, int ACT, bool HAS_RES, bool REMAP, bool CONV = false, bool NARROW = false>
__global__ __launch_bounds__(NTHREADS2) void gemm_kernel_p3(GemmArgs p) {
    constexpr int BNT = NARROW ? 64 : BN2;
    constexpr int MIW = NARROW ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = ROWB / (int)sizeof(TI);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int tm = lin / p.tiles_n, tn = lin - tm * p.tiles_n;
    const int m0 = tm * BM2, n0 = tn * BNT;

    // staging: X tile = 32 instructions of 8 rows, W tile = 16; wave w issues X instr {w, w+8, w+16, w+24}, W {w, w+8}
    const char* srcX[4];
    const char* srcW[2];
    unsigned cmask[CONV ? 2 : 1] = {};     // CONV: two 16-bit tap-validity masks per register
    // CONV: swz(row) = (wave*4 + lane/16) & 7 for every piece of a lane -> one chunk position / tap per lane and K slice
    const int cchunk = ((lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7)) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);
        int gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        if constexpr (CONV) {
            const int x = gm % p.conv_W, y = (gm / p.conv_W) % p.conv_H;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy >= 0 && yy < p.conv_H && xx >= 0 && xx < p.conv_W) mk |= 1u << t;
            }
            cmask[i >> 1] |= mk << ((i & 1) * 16);
            srcX[i] = p.A + (((size_t)gm << p.conv_lgC) * 2);              // centre pixel, channel 0
        } else {
            srcX[i] = p.A + ((size_t)gm * p.lda) * sizeof(TI) + chunk * 16;
        }
    }
#pragma unroll
    for (int i = 0; i < (NARROW ? 1 : 2); ++i) {
        const int row = (i * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        srcW[i] = p.W + ((size_t)gn * p.ldw) * sizeof(TI) + chunk * 16;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    auto issue = [&](int stage, int kt) {
        const unsigned base = __builtin_amdgcn_readfirstlane(ldsw + (unsigned)stage * (unsigned)STAGE2);
        const size_t koff = (size_t)kt * ROWB;
        if constexpr (CONV) {
            const int kk = kt * 64 + cchunk;
            const int tap = kk >> p.conv_lgC, cc = kk & ((1 << p.conv_lgC) - 1);
            const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;                           // tap / 3, tap % 3 for tap < 16
            const long long delta = (long long)(((((dy - 1) * p.conv_W + (dx - 1)) << p.conv_lgC) + cc) * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = (cmask[i >> 1] >> ((i & 1) * 16 + tap)) & 1u;
                const char* src = ok ? srcX[i] + delta : reinterpret_cast<const char*>(g_zero_chunk);
                glds16_asm(src, base + i * 8192);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16_asm(srcX[i] + koff, base + i * 8192);
        }
#pragma unroll
        for (int i = 0; i < (NARROW ? 1 : 2); ++i) glds16_asm(srcW[i] + koff, base + BM2 * ROWB + i * 8192);
    };

    const int wm = NARROW ? wave : wave >> 1, wn = NARROW ? 0 : wave & 1;
    const int lr = lane & 31, hi = lane >> 5;
    int offX[MIW], offW[2], sxX[MIW], sxW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rx = wm * (32 * MIW) + i * 32 + lr;
        const int rw = wn * 64 + i * 32 + lr;
        if (i < MIW) {
            offX[i] = rx * ROWB;
            sxX[i] = swz(rx);
        }
        offW[i] = rw * ROWB;
        sxW[i] = swz(rw);
    }
    f32x16 acc[MIW][2];
#pragma unroll
    for (int i = 0; i < MIW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = p.K / BK;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int st = 0;                      // stage holding slice kt
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {                                                  // slice kt landed, kt+1 may still fly
            if constexpr (NARROW) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(GDBG(p) & 2)) __syncthreads();   // everyone's part of slice kt is in LDS; everyone is done reading slice kt-1
        int st2 = st + 2;
        st2 = st2 >= NSTAGE2 ? st2 - NSTAGE2 : st2;
        if (kt + 2 < nk && !(GDBG(p) & 1)) issue(st2, kt + 2);
        const char* sX = smem + st * STAGE2;
        mma_slice_db<TI, MIW>(acc, sX, sX + BM2 * ROWB, offX, offW, sxX, sxW, hi);
        st = st + 1 >= NSTAGE2 ? 0 : st + 1;
    }
    __syncthreads();                 // every wave is done with the ring: reuse it as per-wave transpose buffers
    const int mb = m0 + wm * (32 * MIW), nb = n0 + wn * 64;
    f32x16 (*accp)[2] = acc;
    if (GDBG(p) & 4) {
        if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.out)[0] = acc[MIW - 1][1][3];
        return;
    }
    const bool full = mb + 32 * MIW <= p.M && nb + 64 <= p.N;
    if constexpr (NARROW) {
        if (full) epilogue_lds<TO, ACT, HAS_RES, REMAP, true, 1>(accp, p, mb, nb, lane, smem + wave * EPI_WAVE_BYTES);
        else epilogue_lds<TO, ACT, HAS_RES, REMAP, false, 1>(accp, p, mb, nb, lane, smem + wave * EPI_WAVE_BYTES);
        return;
    }
    if constexpr (kBf16PackedEpilogue && sizeof(TO) == 2 && !HAS_RES && !REMAP) {
        if (full) epilogue_lds_bf16<ACT, true>(accp, p, mb, nb, lane, smem + wave * EPI_WAVE_BYTES);
        else epilogue_lds_bf16<ACT, false>(accp, p, mb, nb, lane, smem + wave * EPI_WAVE_BYTES);
    } else {
        if (full) epilogue_lds<TO, ACT, HAS_RES, REMAP, true>(accp, p, mb, nb, lane, smem + wave * EPI_WAVE_BYTES);
        else epilogue_lds<TO, ACT, HAS_RES, REMAP, false>(accp, p, mb, nb, lane, smem + wave * EPI_WAVE_BYTES);
    }
}

template <typename TI, typename TO, int ACT, bool HAS_RES, bool REMAP, bool CONV = false, bool NARROW = false>
int launch_p3_inst(const GemmArgs& a, hipStream_t s) {
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&gemm_kernel_p3<TI, TO, ACT, HAS_RES, REMAP, CONV, NARROW>), NSTAGE2 * STAGE2, "cfsar_gemm")) return rc;
    const int tiles_m = (a.M + BM2 - 1) / BM2;
    hipLaunchKernelGGL((gemm_kernel_p3<TI, TO, ACT, HAS_RES, REMAP, CONV, NARROW>), dim3(tiles_m * a.tiles_n), dim3(NTHREADS2),
                       NSTAGE2 * STAGE2, s, a);
    return cfsar_check_launch("cfsar_gemm(p3)");
}

// instantiated combinations: {no act, QuickGELU, GELU} x {residual or not}; the row remap (patch-embed scatter) only
// exists with a residual and no activation.
template <typename TI, typename TO>
int launch_p3(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_n = (a.N + BN2 - 1) / BN2;
    const bool r = a.res != nullptr;
    const bool remap = a.row_group > 0 || a.res_mod > 0;
    if (remap) {
        if (a.act != CFSAR_ACT_NONE || !r) return -2;        // caller falls back to the generic v1 kernel
        return launch_p3_inst<TI, TO, CFSAR_ACT_NONE, true, true>(a, s);
    }
    switch (a.act) {
        case CFSAR_ACT_QUICKGELU:
            return r ? launch_p3_inst<TI, TO, CFSAR_ACT_QUICKGELU, true, false>(a, s)
                     : launch_p3_inst<TI, TO, CFSAR_ACT_QUICKGELU, false, false>(a, s);
        case CFSAR_ACT_GELU_ERF:
            return r ? launch_p3_inst<TI, TO, CFSAR_ACT_GELU_ERF, true, false>(a, s)
                     : launch_p3_inst<TI, TO, CFSAR_ACT_GELU_ERF, false, false>(a, s);
        default:
            return r ? launch_p3_inst<TI, TO, CFSAR_ACT_NONE, true, false>(a, s)
                     : launch_p3_inst<TI, TO, CFSAR_ACT_NONE, false, false>(a, s);
    }
}

// compile-time loop: indices are constants in the AST, so the register arrays of the kernels below are split by the FIRST SROA pass
// (a `#pragma unroll` loop index is still dynamic there and leaves them to the size-limited alloca promotion -> scratch)
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// ============================================================================================================
// K1 (few_shot.py:659, 672-676): the patch embedding in ONE launch -- conv1 as a GEMM whose X operand is gathered straight from the
// fp32 NCHW frames (no im2col matrix: at 36 episodes that was 0.87 GB written and read back, two launches of 240 us), + pos[1 + p], rows scattered
// behind each frame's class token, and the class-token rows cls + pos[0] themselves.  P = 16: a 64-wide K slice is 4 image rows x 16 pixels of
// one channel of a patch, a lane's 16-byte LDS chunk = 8 consecutive pixels = two 16-byte fp32 loads, rounded to the operand type in registers
// and written where the LDS-DMA of gemm_kernel_p3 would have put them -- same tile (256 x 128), same ring, same fragment reads and MFMA
// order: an output element is bit-identical to the im2col + cfsar_gemm form.  W still arrives by LDS-DMA.
// Pipeline: the fp32 loads of slice s are issued three iterations ahead (two register sets alternate), written to stage s % 3 one iteration
// ahead; W of slice s is DMA'd two iterations ahead.  The register loads are plain C++ loads (no compiler-invisible register destinations:
// profiles/r04_fault_audit.md) and the K loop is FULLY unrolled (K = 768 = 12 slices): in straight-line code hipcc's wait for a register set
// is exact in its own loads -- vmcnt(8): the next set may still be in flight -- where, across a loop's back edge, it drained to vmcnt(0) at
// every use (one iteration of latency cover instead of two).  Its count ignores the asm DMA instructions in between, so the wait is at
// least as strict as the exact one, and it also covers W of the slice the barrier behind it publishes (issued before the set it waits for).
// ============================================================================================================
struct PatchArgs {
    GemmArgs g;               // M = F * npatch rows, N = D, K = 768, W / ldw, out / ldo, res = pos (fp32), remap fields set
    const float* frames;      // [F, 3, H, W]
    const float* cls;         // [D]
    int H, Wd, gw, npatch, F;
};

// P = 14 (ViT-L/14; round 6): 14 pixels are 56 bytes -- a patch row is 8-byte aligned only and 64 K slots do not hold whole rows.  The K axis is
// therefore laid out in PADDED rows: k' = (c 14 + dy) 16 + dx with dx = 14, 15 zero, 42 rows = 672 slots, padded to 704 = 11 slices of 4 rows; the weight
// matrix is handed over in the same layout (host side, once).  A lane's chunk is then px 0..7 (four 8-byte loads) or px 8..13 + two zeros (three loads
// + one re-load of a valid address whose value is discarded: every lane issues the same loads per slice, no divergent control flow in the K loop).
template <typename TI, typename TO, int P = 16>
__global__ __launch_bounds__(NTHREADS2) void patch_embed_kernel(PatchArgs q) {
    static_assert(P == 16 || P == 14, "patch size 16 or 14");
    const GemmArgs& p = q.g;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, b = blockIdx.x;
    const int qq = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int lin = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (b >> 3);
    const int tm = lin / p.tiles_n, tn = lin - tm * p.tiles_n;
    const int m0 = tm * BM2, n0 = tn * BN2;

    // class-token rows of the frames whose first patch row lies in this tile (few_shot.py:675-676), this tile's columns
    if (tid < BN2 && n0 + tid < p.N) {
        const int ntok = q.npatch + 1;
        for (int f = (m0 + q.npatch - 1) / q.npatch; f < q.F && f * q.npatch < m0 + BM2; ++f)
            reinterpret_cast<TO*>(p.out)[(size_t)f * ntok * p.ldo + n0 + tid] = (TO)(q.cls[n0 + tid] + reinterpret_cast<const float*>(p.res)[n0 + tid]);
    }

    // X pieces: piece i of this wave = rows (8 i + wave) 8 + lane / 8 of the tile; every piece of a lane has the same chunk (swz: see CONV above)
    const int chunk = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);
    const float* srcX[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int gm = m0 + (i * 8 + wave) * 8 + (lane >> 3);
        gm = gm < p.M ? gm : p.M - 1;
        const int f = gm / q.npatch, pp = gm - f * q.npatch;
        const int py = pp / q.gw, px = pp - py * q.gw;
        if constexpr (P == 16) srcX[i] = q.frames + ((size_t)f * 3 * q.H + py * 16 + (chunk >> 1)) * q.Wd + px * 16 + (chunk & 1) * 8;
        else srcX[i] = q.frames + ((size_t)f * 3 * q.H + py * 14) * q.Wd + px * 14 + (chunk & 1) * 8;      // + the slice's (channel, row) offset: load_x
    }
    const char* srcW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 8 + wave) * 8 + (lane >> 3);
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        srcW[i] = p.W + ((size_t)gn * p.ldw) * sizeof(TI) + ((lane & 7) ^ swz(row)) * 16;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    char* const wrX = smem + wave * 1024 + lane * 16;                       // + stage * STAGE2 + i * 8192: where the DMA form writes
    auto issue_w = [&](int stage, int kt) __attribute__((always_inline)) {
        const unsigned base = __builtin_amdgcn_readfirstlane(ldsw + (unsigned)stage * (unsigned)STAGE2);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_asm(srcW[i] + (size_t)kt * ROWB, base + BM2 * ROWB + i * 8192);
    };
    // slice kt = channel kt / 4, image rows 4 (kt % 4) .. + 3 of the patch
    auto load_x = [&](int kt, f32x4 (&xr)[8]) __attribute__((always_inline)) {
        if constexpr (P == 16) {
            const size_t off = ((size_t)(kt >> 2) * q.H + (kt & 3) * 4) * q.Wd;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xr[2 * i] = *reinterpret_cast<const f32x4*>(srcX[i] + off);
                xr[2 * i + 1] = *reinterpret_cast<const f32x4*>(srcX[i] + off + 4);
            }
        } else {
            // padded row r = 4 kt + j (j = the lane's row inside the slice): channel r / 14, image row r % 14 of the patch; r >= 42: zeros
            const int j = chunk >> 1;
            const bool hi8 = chunk & 1;                    // px 8 .. 13 (+ two zero slots)
            size_t offj[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int r = 4 * kt + jj;
                const int rr = r < 42 ? r : 0;             // (a valid address; the values are discarded)
                offj[jj] = ((size_t)(rr / 14) * q.H + (rr % 14)) * q.Wd;
            }
            const size_t off = j == 0 ? offj[0] : (j == 1 ? offj[1] : (j == 2 ? offj[2] : offj[3]));
            const bool valid = 4 * kt + j < 42;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* src = srcX[i] + off;
                const f32x2 a0 = *reinterpret_cast<const f32x2*>(src), a1 = *reinterpret_cast<const f32x2*>(src + 2);
                const f32x2 a2 = *reinterpret_cast<const f32x2*>(src + 4), a3 = *reinterpret_cast<const f32x2*>(hi8 ? src : src + 6);
                xr[2 * i] = f32x4{valid ? a0[0] : 0.f, valid ? a0[1] : 0.f, valid ? a1[0] : 0.f, valid ? a1[1] : 0.f};
                xr[2 * i + 1] = f32x4{valid ? a2[0] : 0.f, valid ? a2[1] : 0.f, (valid && !hi8) ? a3[0] : 0.f, (valid && !hi8) ? a3[1] : 0.f};
            }
        }
    };
    auto write_x = [&](int stage, const f32x4 (&xr)[8]) __attribute__((always_inline)) {
        typedef typename Vec2B<TI>::v8 TI8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 a = xr[2 * i], c = xr[2 * i + 1];
            TI8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (TI)a[j];
                o[4 + j] = (TI)c[j];
            }
            *reinterpret_cast<TI8*>(wrX + stage * STAGE2 + i * 8192) = o;
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, hi = lane >> 5;
    int offX[2], offW[2], sxX[2], sxW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rx = wm * 64 + i * 32 + lr;
        const int rw = wn * 64 + i * 32 + lr;
        offX[i] = rx * ROWB;
        sxX[i] = swz(rx);
        offW[i] = rw * ROWB;
        sxW[i] = swz(rw);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    constexpr int nk = P == 16 ? 12 : 11;        // K = 768 | 704 padded-row slots (launcher)
    // register-load INSTRUCTIONS of one slice per lane, as a lower bound (the explicit wait below may only name operations that were really issued behind
    // W(kt)): P = 16 exactly eight 16-byte loads; P = 14 at least eight -- hipcc (ROCm 7.2) emits twelve: one 16-byte + two 8-byte per piece
    constexpr int NXL = 8;
    f32x4 xa[8], xb[8];                          // even / odd slices
    issue_w(0, 0);
    issue_w(1, 1);
    load_x(0, xa);
    load_x(1, xb);
    write_x(0, xa);
    load_x(2, xa);
    // iteration kt: X of slice kt + 1 goes from its registers into stage (kt + 1) % 3 (free since the barrier of iteration kt - 1), barrier
    // (slice kt complete, slice kt - 1 read by everyone), W of slice kt + 2 and the registers of slice kt + 3 are requested, slice kt is computed
    static_for<nk>([&](auto kt_c) __attribute__((always_inline)) {
        constexpr int kt = decltype(kt_c)::value;
        constexpr int st = kt % NSTAGE2, st1 = (kt + 1) % NSTAGE2, st2 = (kt + 2) % NSTAGE2;
        f32x4 (&xnext)[8] = (kt & 1) ? xa : xb;                             // registers of slice kt + 1, then of slice kt + 3
        if constexpr (kt + 1 < nk) write_x(st1, xnext);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // last slice: its W
        // The barrier publishes W of slice kt (two asm LDS-DMA pieces per wave, issued two iterations ago).  hipcc's own wait in front of write_x
        // (for the registers of slice kt + 1, requested AFTER that DMA) already covers it -- but hipcc does not count the asm DMA, so the
        // requirement is also stated explicitly: behind W(kt) this wave issued X(kt + 1): NXL loads, W(kt + 1): 2, X(kt + 2): NXL.
        if constexpr (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NXL + 2 + (kt + 2 < nk ? NXL : 0)) : "memory");
        __syncthreads();
        if constexpr (kt + 2 < nk) issue_w(st2, kt + 2);
        if constexpr (kt + 3 < nk) load_x(kt + 3, xnext);
        const char* sX = smem + st * STAGE2;
        mma_slice_db<TI, 2>(acc, sX, sX + BM2 * ROWB, offX, offW, sxX, sxW, hi);
    });
    __syncthreads();                 // every wave is done with the ring: reuse it as per-wave transpose buffers
    const int mb = m0 + wm * 64, nb = n0 + wn * 64;
    f32x16 (*accp)[2] = acc;
    const bool full = mb + 64 <= p.M && nb + 64 <= p.N;
    if (full) epilogue_lds<TO, CFSAR_ACT_NONE, true, true, true>(accp, p, mb, nb, lane, smem + wave * EPI_WAVE_BYTES);
    else epilogue_lds<TO, CFSAR_ACT_NONE, true, true, false>(accp, p, mb, nb, lane, smem + wave * EPI_WAVE_BYTES);
}

template <typename TI, typename TO, int P = 16>
int launch_patch_embed(const PatchArgs& a, hipStream_t s) {
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&patch_embed_kernel<TI, TO, P>), NSTAGE2 * STAGE2, "cfsar_patch_embed")) return rc;
    const int tiles_m = (a.g.M + BM2 - 1) / BM2;
    hipLaunchKernelGGL((patch_embed_kernel<TI, TO, P>), dim3(tiles_m * a.g.tiles_n), dim3(NTHREADS2), NSTAGE2 * STAGE2, s, a);
    return cfsar_check_launch("cfsar_patch_embed");
}

// fp16 output (the bf16 mode's residual stream): no activation + residual, with or without the patch-embed row remap
int launch_p3_f16(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_n = (a.N + BN2 - 1) / BN2;
    if (a.act != CFSAR_ACT_NONE || !a.res) return -2;
    if (a.row_group > 0 || a.res_mod > 0) return launch_p3_inst<__bf16, _Float16, CFSAR_ACT_NONE, true, true>(a, s);
    return launch_p3_inst<__bf16, _Float16, CFSAR_ACT_NONE, true, false>(a, s);
}

// ============================================================================================================
// 256(M) x 256(N) tiles: shared constants and the tile walk of the p10 / p12 kernels below.  (Round 1 also carried p4 / p5 /
// p6 / p8 / p9 siblings -- LDS-DMA operand paths, 64-byte K slices, ping-pong groups; measured within 3 % of each other and
// behind p10 / p12, tables in profiles/r01_gemm_ablation.md -- all removed from the source.)
// ============================================================================================================
constexpr int BM4 = 256;
constexpr int BN4 = 256;
constexpr int LDS4 = 8 * EPI_WAVE_BYTES;     // 139264 B: the 8 wave-private epilogue regions (>= the two 64 KiB operand stages)

// linear tile index -> (row band, column) walking `group` row bands before the next column of tiles (group <= 1: row-major)
__device__ __forceinline__ void tile_of(int lin, int tiles_m, int tiles_n, int group, int& tm, int& tn) {
    if (group <= 1) {
        tm = lin / tiles_n;
        tn = lin - tm * tiles_n;
        return;
    }
    const int per = group * tiles_n;
    const int gid = lin / per, first = gid * group;
    const int gsz = tiles_m - first < group ? tiles_m - first : group;
    const int rem = lin - gid * per;
    tm = first + rem % gsz;
    tn = rem / gsz;
}
// short-K GEMMs (QKV, out_proj, c_fc): groups of 8 row bands (+2...+4 % measured); dbg 512 / 1024: 4 / 16; 2048: off
__device__ __forceinline__ int tile_group(const GemmArgs& p, int ntiles) {
    if (GDBG(p) & 2048) return 1;
    if (GDBG(p) & 512) return 4;
    if (GDBG(p) & 1024) return 16;
    return ((GDBG(p) & 256) || (p.K <= 1024 && ntiles >= 512)) ? 8 : 1;
}


#ifdef CFSAR_DEV
static int g_variant_override = 0, g_dbg_override = 0;   // dev tool: in-process A/B (tools/gemm_ab.py)
#endif
// product policy of the gemm_vit.hip kernel (measured, M = 252 160: profiles/r02_gemm_ab.md): the LDS-DMA operand path for the
// short-K GEMMs (QKV 859 vs 871 us register-staged vs 890 p12; c_fc 1 279 vs 1 335 vs 1 292), the register-staged path (loads two
// K tiles ahead) for K > 1 024 (c_proj 1 093 vs 1 156 us with DMA), write-through (sc1) stores where the output is not read back
// by the same launch (QKV 850 vs 859, c_fc 1 273 vs 1 279; the in-place residual update is slower with it: 365 vs 350)
constexpr bool kUseVitKernel = true;
constexpr int kVitGroup = 8, kVitColfast = 0;

// ---- helpers shared by the register-staged kernels (p10, p12)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vector: loads/stores stay SSA values (no memcpy)

// ============================================================================================================
// v4 ("p10"): ONE wave per SIMD (256 threads, 2 x 2 waves, 128 x 128 wave tiles, 256 accumulators in AGPRs), register-staged
// operands, WHOLE CACHE LINES per request.  Measured (tools/ubench/vgpr_l2.hip): the L2 -> CU path delivers
// ~100 GB/s per CU for 8 rows x 128 B per wave-instruction but only ~61 GB/s for the 16 rows x 64 B pieces that 64-byte K
// slices imply (the L2 serves requests, not bytes): every 64-byte-slice kernel above runs its operand stream at > 50 %
// of that ceiling.  Here K advances in 128-byte tiles (64 bf16): TWO 64 KiB LDS stages of 128-byte rows
// (chunk ^= (row >> 1) & 7), register staging with ONE 64-register set split into an X half and a W half:
//     sub-step 0: MFMA(k 0-15)   | reads(k 16-31) | ds_write X(kt+1)
//     sub-step 1: MFMA(k 16-31)  | reads(k 32-47) | global_load X(kt+2)
//     sub-step 2: MFMA(k 32-47)  | reads(k 48-63) | ds_write W(kt+1)
//     sub-step 3: MFMA(k 48-63)  | global_load W(kt+2) | barrier | reads(kt+1, k 0-15)
// Each half lives 3 sub-steps (~1 500 cycles) in registers between its load and its LDS write; ONE barrier per 128-byte
// K tile orders both hazards (RAW: stage kt+1 complete before its first fragment read; WAR: stage kt's last fragment
// reads (sub-step 2) before X(kt+2) overwrites it in the next tile's sub-step 0).  Any K that is a multiple of 64.
// ============================================================================================================
constexpr int ROWB10 = 128;
constexpr int STAGE10 = (BM4 + BN4) * ROWB10;       // 64 KiB
// CONV: the X operand is gathered on the fly from an NHWC activation tensor (implicit GEMM for nn.Conv2d(3, padding=1),
// few_shot.py:196): row m = output pixel (f, y, x), K index kk = tap * C + c with tap = ky * 3 + kx; a lane's 16-byte chunk
// (8 channels) of K tile kt comes from pixel (y + ky - 1, x + kx - 1) or is zero outside the image -- no im2col matrix.
template <typename TO, int ACT, bool HAS_RES, bool PERSIST, bool CONV = false, typename TI = __bf16>
__global__ __launch_bounds__(256) void gemm_kernel_p10(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // PERSIST: one workgroup per CU walks the virtual block ids b, b + grid, ... (same XCD every time: grid % 8 == 0)
    const int nwg = PERSIST ? p.ntiles : (int)gridDim.x;
    const int q = nwg >> 3, r = nwg & 7;
    for (int b = blockIdx.x; b < nwg; b += PERSIST ? (int)gridDim.x : nwg) {
    const int xcd = b & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    int tm, tn;
    tile_of(lin, nwg / p.tiles_n, p.tiles_n, tile_group(p, nwg), tm, tn);
    const int m0 = tm * BM4, n0 = tn * BN4;

    // staging: an operand tile = 32 pieces of 8 rows x 128 B (1 KiB per wave-instruction); wave w owns pieces
    // {w, w+4, ..., w+28}; a lane fetches global chunk (lane&7) ^ swz(row) and writes LDS lane-linearly
    unsigned offX[8], offWg[8];
    unsigned cmask[CONV ? 4 : 1] = {};     // CONV: two 16-bit masks per register; bit t = tap t reads inside the image
    // CONV: first K element (0, 8, ..., 56) of this lane's chunk inside a K tile -- swz(row) = (wave*4 + lane/16) & 7 is
    // the same for all 8 pieces of a lane, so tap and channel offset of a K tile are per lane, not per piece
    const int cchunk = ((lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7)) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (i * 4 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);
        int gm = ((GDBG(p) & 8) ? 0 : m0) + row;
        gm = gm < p.M ? gm : p.M - 1;
        int gn = ((GDBG(p) & 8) ? 0 : n0) + row;
        gn = gn < p.N ? gn : p.N - 1;
        if constexpr (CONV) {
            const int x = gm % p.conv_W, y = (gm / p.conv_W) % p.conv_H;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                if (yy >= 0 && yy < p.conv_H && xx >= 0 && xx < p.conv_W) mk |= 1u << t;
            }
            cmask[i >> 1] |= mk << ((i & 1) * 16);
            offX[i] = ((unsigned)gm << p.conv_lgC) * 2u;                   // centre pixel, channel 0
        } else {
            offX[i] = (unsigned)gm * (unsigned)p.lda * 2u + chunk * 16;
        }
        offWg[i] = (unsigned)gn * (unsigned)p.ldw * 2u + chunk * 16;
    }
    const int wr_off = wave * 1024 + lane * 16;          // + piece i * 4096 (+ BM4*ROWB10 for W) inside a stage

    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, hi = lane >> 5;
    int rdX[4], rdW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rx = wm * 128 + i * 32 + lr;
        rdX[i] = rx * ROWB10 + ((hi ^ swz(rx)) << 4);                      // sub-step ss: ^ (ss << 5)
        const int rw = wn * 128 + i * 32 + lr;
        rdW[i] = BM4 * ROWB10 + rw * ROWB10 + ((hi ^ swz(rw)) << 4);
    }
    f32x16 acc[2][4][2];                                                   // [n half][mi][ni within the half]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[h][i][j][e] = 0.0f;

    const int nk = p.K / 64;
    u32x4 GX[8], GW[8];
    uint4 xfA[4], wfA[4], xfB[4], wfB[4];
    auto gloadX = [&](int kt, auto J) {
        constexpr int j = decltype(J)::value;
        if constexpr (CONV) {
            const int kk = kt * 64 + cchunk;                                             // (hipcc hoists this block per K tile)
            const int tap = kk >> p.conv_lgC, cc = kk & ((1 << p.conv_lgC) - 1);
            const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;                           // tap / 3, tap % 3 for tap < 16
            const int delta = ((((dy - 1) * p.conv_W + (dx - 1)) << p.conv_lgC) + cc) * 2;
            const bool ok = (cmask[j >> 1] >> ((j & 1) * 16 + tap)) & 1u;
            const unsigned off = ok ? offX[j] + (unsigned)delta : 0u;
            u32x4 v = *reinterpret_cast<const u32x4*>(p.A + off);
            const unsigned keep = ok ? 0xffffffffu : 0u;
            GX[j] = v & keep;
        } else {
            GX[j] = *reinterpret_cast<const u32x4*>(p.A + (size_t)kt * ROWB10 + offX[j]);
        }
    };
    auto gloadW = [&](int kt, auto J) {
        GW[decltype(J)::value] = *reinterpret_cast<const u32x4*>(p.W + (size_t)kt * ROWB10 + offWg[decltype(J)::value]);
    };
    auto swriteX = [&](int stage, auto J) {
        *reinterpret_cast<u32x4*>(smem + stage * STAGE10 + wr_off + decltype(J)::value * 4096) = GX[decltype(J)::value];
    };
    auto swriteW = [&](int stage, auto J) {
        *reinterpret_cast<u32x4*>(smem + stage * STAGE10 + BM4 * ROWB10 + wr_off + decltype(J)::value * 4096) = GW[decltype(J)::value];
    };
    // read order = order of first use by the MFMA sequence (ni-major): x0 w0 x1 x2 x3 w1 w2 w3
    auto load_one = [&](int stage, int ss, auto J, uint4 (&xf)[4], uint4 (&wf)[4]) {
        constexpr int j = decltype(J)::value;
        const char* base = smem + stage * STAGE10;
        const int x2 = ss << 5;
        constexpr int isx[8] = {1, 0, 1, 1, 1, 0, 0, 0};
        constexpr int idx[8] = {0, 0, 1, 2, 3, 1, 2, 3};
        if constexpr (isx[j] != 0) xf[idx[j]] = *reinterpret_cast<const uint4*>(base + (rdX[idx[j]] ^ x2));
        else wf[idx[j]] = *reinterpret_cast<const uint4*>(base + (rdW[idx[j]] ^ x2));
    };
    auto mfma_one = [&](auto J, uint4 (&xf)[4], uint4 (&wf)[4]) {
        constexpr int j = decltype(J)::value;
        constexpr int ni = j >> 2, mi = j & 3;
        acc[ni >> 1][mi][ni & 1] = cfsar_mfma_32x32x16<TI>(wf[ni], xf[mi], acc[ni >> 1][mi][ni & 1]);
    };
    // one 128-byte K tile; cur / nxt = LDS stage of tile kt / kt+1.  WRITE: tile kt+1 exists (its registers are written
    // to LDS and its first fragments are prefetched); LOAD: tile kt+2 exists
    auto tile = [&](int kt, int cur, int nxt, auto LOAD, auto WRITE) {
        constexpr bool load = decltype(LOAD)::value, write = decltype(WRITE)::value;
        static_for<16>([&](auto J) {                                    // sub-step 0
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfA, wfA);
            if constexpr (j < 8) load_one(cur, 1, J, xfB, wfB);
            else if constexpr (write) swriteX(nxt, std::integral_constant<int, j - 8>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<16>([&](auto J) {                                    // sub-step 1
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfB, wfB);
            if constexpr (j < 8) load_one(cur, 2, J, xfA, wfA);
            else if constexpr (load) gloadX(kt + 2, std::integral_constant<int, j - 8>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<16>([&](auto J) {                                    // sub-step 2
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfA, wfA);
            if constexpr (j < 8) load_one(cur, 3, J, xfB, wfB);
            else if constexpr (write) swriteW(nxt, std::integral_constant<int, j - 8>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<16>([&](auto J) {                                    // sub-step 3
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfB, wfB);
            if constexpr (j < 8) {
                if constexpr (load) gloadW(kt + 2, J);
            } else if constexpr (write) load_one(nxt, 0, std::integral_constant<int, j - 8>{}, xfA, wfA);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j == 7 && write) {
                if (!(GDBG(p) & 2)) __syncthreads();                      // hipcc adds lgkmcnt(0): this wave's ds_writes
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    // prologue: tile 0 -> stage 0 through the registers, tile 1 -> registers
    static_for<8>([&](auto J) { gloadX(0, J); });
    static_for<8>([&](auto J) { gloadW(0, J); });
    static_for<8>([&](auto J) { swriteX(0, J); });
    static_for<8>([&](auto J) { swriteW(0, J); });
    if (nk > 1) {
        static_for<8>([&](auto J) { gloadX(1, J); });
        static_for<8>([&](auto J) { gloadW(1, J); });
    }
    __syncthreads();
    static_for<8>([&](auto J) { load_one(0, 0, J, xfA, wfA); });
    int kt = 0;
    for (; kt < nk - 2; ++kt) tile(kt, kt & 1, (kt + 1) & 1, T_{}, T_{});
    if (nk > 1) {
        tile(kt, kt & 1, (kt + 1) & 1, F_{}, T_{});
        ++kt;
    }
    tile(kt, kt & 1, (kt + 1) & 1, F_{}, F_{});
    __syncthreads();
    if (GDBG(p) & 4) {
        if (acc[0][0][0][0] == 123.456f) reinterpret_cast<float*>(p.out)[0] = acc[1][1][1][3] + acc[0][3][1][2] + acc[1][2][0][1];
        if constexpr (PERSIST) continue;
        else return;
    }
    char* wbuf = smem + wave * EPI_WAVE_BYTES;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int mb = m0 + wm * 128 + half * 64, nb = n0 + wn * 128 + nh * 64;
            const bool full = mb + 64 <= p.M && nb + 64 <= p.N;
            if constexpr (sizeof(TO) == 2 && !HAS_RES) {
                if ((GDBG(p) & 64) && !p.relu) {                          // A/B: packed bf16 staging + 16-byte stores
                    if (full) epilogue_lds_bf16<ACT, true>(&acc[nh][2 * half], p, mb, nb, lane, wbuf);
                    else epilogue_lds_bf16<ACT, false>(&acc[nh][2 * half], p, mb, nb, lane, wbuf);
                    continue;
                }
            }
            if constexpr (ACT != CFSAR_ACT_NONE) {                     // activation temporaries: 32 rows per pass (no spills)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int mq = mb + q2 * 32;
                    if (mq + 32 <= p.M && nb + 64 <= p.N)
                        epilogue_lds<TO, ACT, HAS_RES, false, true, 1, false>(&acc[nh][2 * half + q2], p, mq, nb, lane, wbuf);
                    else
                        epilogue_lds<TO, ACT, HAS_RES, false, false, 1, false>(&acc[nh][2 * half + q2], p, mq, nb, lane, wbuf);
                }
                continue;
            }
            if (full) epilogue_lds<TO, ACT, HAS_RES, false, true, 2, false>(&acc[nh][2 * half], p, mb, nb, lane, wbuf);
            else epilogue_lds<TO, ACT, HAS_RES, false, false, 2, false>(&acc[nh][2 * half], p, mb, nb, lane, wbuf);
        }
    if constexpr (PERSIST) __syncthreads();            // the epilogue staging aliases LDS stage 0 of the next tile
    }
}

// persistent kernels: one workgroup per CU, rounded down to a multiple of the 8 XCDs (the block -> XCD walk assumes it)
static inline int persistent_grid() {
    const int n = cfsar_num_cus() & ~7;
    return n >= 8 ? n : 8;
}

template <typename TO, int ACT, bool HAS_RES, bool PERSIST, bool CONV = false, typename TI = __bf16>
int launch_p10_inst(const GemmArgs& a, hipStream_t s) {
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&gemm_kernel_p10<TO, ACT, HAS_RES, PERSIST, CONV, TI>), LDS4, "cfsar_gemm")) return rc;
    hipLaunchKernelGGL((gemm_kernel_p10<TO, ACT, HAS_RES, PERSIST, CONV, TI>), dim3(PERSIST ? persistent_grid() : a.ntiles), dim3(256), LDS4, s, a);
    return cfsar_check_launch("cfsar_gemm(p10)");
}

template <typename TO, bool PERSIST>
int launch_p10(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_n = (a.N + BN4 - 1) / BN4;
    a.ntiles = ((a.M + BM4 - 1) / BM4) * a.tiles_n;
    const bool r = a.res != nullptr;
    if (a.row_group > 0 || a.res_mod > 0 || a.act == CFSAR_ACT_GELU_ERF || a.K % 64 != 0) return -2;
    if ((size_t)a.M * a.lda * 2 >= (1ull << 32) || (size_t)a.N * a.ldw * 2 >= (1ull << 32)) return -2;   // 32-bit offsets
    if (a.act == CFSAR_ACT_QUICKGELU) return r ? -2 : launch_p10_inst<TO, CFSAR_ACT_QUICKGELU, false, PERSIST>(a, s);
    return r ? launch_p10_inst<TO, CFSAR_ACT_NONE, true, PERSIST>(a, s) : launch_p10_inst<TO, CFSAR_ACT_NONE, false, PERSIST>(a, s);
}

// ============================================================================================================
// v5 ("p12"): p10's operand path (128-byte K tiles, whole-line requests, register staging, two 64 KiB LDS stages, one barrier
// per K tile) with TWO waves per SIMD: 512 threads, 8 waves as 2(M) x 4(N), wave tile 128 x 64 (128 accumulators).  For the
// QuickGELU epilogue (c_fc) the second wave per SIMD overlaps the 2-transcendentals-per-element VALU work and the stores of
// one wave with those of the other, which is what p10 lacks; the main loop keeps p10's request efficiency, which p6 lacks.
// Per wave and K tile: 32 MFMAs, 24 fragment reads, 4 + 4 global loads, 4 + 4 LDS writes.
// ============================================================================================================
// PERSIST: one workgroup per CU walks the virtual block ids b, b + grid, ... (grid % 8 == 0 keeps a workgroup on its XCD's band)
template <typename TO, int ACT, bool HAS_RES, bool PERSIST = false, typename TI = __bf16>
__global__ __launch_bounds__(512, 2) void gemm_kernel_p12(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nwg = PERSIST ? p.ntiles : (int)gridDim.x;
    const int q = nwg >> 3, r = nwg & 7;
    for (int b = blockIdx.x; b < nwg; b += PERSIST ? (int)gridDim.x : nwg) {
    const int xcd = b & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    int tm, tn;
    tile_of(lin, nwg / p.tiles_n, p.tiles_n, tile_group(p, nwg), tm, tn);
    const int m0 = tm * BM4, n0 = tn * BN4;

    // staging: an operand tile = 32 pieces of 8 rows x 128 B; wave w owns pieces {w, w+8, w+16, w+24}
    unsigned offX[4], offWg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);
        int gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        offX[i] = (unsigned)gm * (unsigned)p.lda * 2u + chunk * 16;
        offWg[i] = (unsigned)gn * (unsigned)p.ldw * 2u + chunk * 16;
    }
    const int wr_off = wave * 1024 + lane * 16;          // + piece i * 8192 (+ BM4*ROWB10 for W) inside a stage

    const int wm = wave >> 2, wn = wave & 3;
    const int lr = lane & 31, hi = lane >> 5;
    int rdX[4], rdW[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rx = wm * 128 + i * 32 + lr;
        rdX[i] = rx * ROWB10 + ((hi ^ swz(rx)) << 4);                      // sub-step ss: ^ (ss << 5)
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rw = wn * 64 + i * 32 + lr;
        rdW[i] = BM4 * ROWB10 + rw * ROWB10 + ((hi ^ swz(rw)) << 4);
    }
    f32x16 acc[4][2];                                                      // [mi][ni]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = p.K / 64;
    u32x4 GX[4], GW[4];
    uint4 xfA[4], wfA[2], xfB[4], wfB[2];
    auto gloadX = [&](int kt, auto J) {
        GX[decltype(J)::value] = *reinterpret_cast<const u32x4*>(p.A + (size_t)kt * ROWB10 + offX[decltype(J)::value]);
    };
    auto gloadW = [&](int kt, auto J) {
        GW[decltype(J)::value] = *reinterpret_cast<const u32x4*>(p.W + (size_t)kt * ROWB10 + offWg[decltype(J)::value]);
    };
    auto swriteX = [&](int stage, auto J) {
        *reinterpret_cast<u32x4*>(smem + stage * STAGE10 + wr_off + decltype(J)::value * 8192) = GX[decltype(J)::value];
    };
    auto swriteW = [&](int stage, auto J) {
        *reinterpret_cast<u32x4*>(smem + stage * STAGE10 + BM4 * ROWB10 + wr_off + decltype(J)::value * 8192) = GW[decltype(J)::value];
    };
    // read order = order of first use by the MFMA sequence (ni-major): x0 w0 x1 x2 x3 w1
    auto load_one = [&](int stage, int ss, auto J, uint4 (&xf)[4], uint4 (&wf)[2]) {
        constexpr int j = decltype(J)::value;
        const char* base = smem + stage * STAGE10;
        const int x2 = ss << 5;
        constexpr int isx[6] = {1, 0, 1, 1, 1, 0};
        constexpr int idx[6] = {0, 0, 1, 2, 3, 1};
        if constexpr (isx[j] != 0) xf[idx[j]] = *reinterpret_cast<const uint4*>(base + (rdX[idx[j]] ^ x2));
        else wf[idx[j]] = *reinterpret_cast<const uint4*>(base + (rdW[idx[j]] ^ x2));
    };
    auto mfma_one = [&](auto J, uint4 (&xf)[4], uint4 (&wf)[2]) {
        constexpr int j = decltype(J)::value;
        constexpr int ni = j >> 2, mi = j & 3;
        acc[mi][ni] = cfsar_mfma_32x32x16<TI>(wf[ni], xf[mi], acc[mi][ni]);
    };
    // one 128-byte K tile = 4 sub-steps of 8 MFMAs; after MFMA j: a fragment read (j < 6) and, for j >= 4, one memory filler
    auto tile = [&](int kt, int cur, int nxt, auto LOAD, auto WRITE) {
        constexpr bool load = decltype(LOAD)::value, write = decltype(WRITE)::value;
        static_for<8>([&](auto J) {                                     // sub-step 0: reads(ss 1), ds_write X(kt+1)
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfA, wfA);
            if constexpr (j < 6) load_one(cur, 1, J, xfB, wfB);
            if constexpr (j >= 4 && write) swriteX(nxt, std::integral_constant<int, j - 4>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<8>([&](auto J) {                                     // sub-step 1: reads(ss 2), global_load X(kt+2)
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfB, wfB);
            if constexpr (j < 6) load_one(cur, 2, J, xfA, wfA);
            if constexpr (j >= 4 && load) gloadX(kt + 2, std::integral_constant<int, j - 4>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<8>([&](auto J) {                                     // sub-step 2: reads(ss 3), ds_write W(kt+1)
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfA, wfA);
            if constexpr (j < 6) load_one(cur, 3, J, xfB, wfB);
            if constexpr (j >= 4 && write) swriteW(nxt, std::integral_constant<int, j - 4>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<8>([&](auto J) {                                     // sub-step 3: global_load W(kt+2) | barrier | reads(kt+1, ss 0)
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfB, wfB);
            if constexpr (j < 2) {
                if constexpr (load) {
                    gloadW(kt + 2, std::integral_constant<int, 2 * j>{});
                    gloadW(kt + 2, std::integral_constant<int, 2 * j + 1>{});
                }
            } else if constexpr (write) load_one(nxt, 0, std::integral_constant<int, j - 2>{}, xfA, wfA);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j == 1 && write) {
                __syncthreads();                                        // hipcc adds lgkmcnt(0): this wave's ds_writes
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    static_for<4>([&](auto J) { gloadX(0, J); });
    static_for<4>([&](auto J) { gloadW(0, J); });
    static_for<4>([&](auto J) { swriteX(0, J); });
    static_for<4>([&](auto J) { swriteW(0, J); });
    if (nk > 1) {
        static_for<4>([&](auto J) { gloadX(1, J); });
        static_for<4>([&](auto J) { gloadW(1, J); });
    }
    __syncthreads();
    static_for<6>([&](auto J) { load_one(0, 0, J, xfA, wfA); });
    int kt = 0;
    for (; kt < nk - 2; ++kt) tile(kt, kt & 1, (kt + 1) & 1, T_{}, T_{});
    if (nk > 1) {
        tile(kt, kt & 1, (kt + 1) & 1, F_{}, T_{});
        ++kt;
    }
    tile(kt, kt & 1, (kt + 1) & 1, F_{}, F_{});
    __syncthreads();
    if (GDBG(p) & 4) {
        if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.out)[0] = acc[1][1][3] + acc[3][1][2] + acc[2][0][1];
        if constexpr (PERSIST) continue;
        else return;
    }
    char* wbuf = smem + wave * EPI_WAVE_BYTES;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int mb = m0 + wm * 128 + half * 64, nb = n0 + wn * 64;
        const bool full = mb + 64 <= p.M && nb + 64 <= p.N;
        if (full) epilogue_lds<TO, ACT, HAS_RES, false, true>(&acc[2 * half], p, mb, nb, lane, wbuf);
        else epilogue_lds<TO, ACT, HAS_RES, false, false>(&acc[2 * half], p, mb, nb, lane, wbuf);
    }
    if constexpr (PERSIST) __syncthreads();            // the epilogue staging aliases the LDS stages of the next tile
    }
}

template <typename TO, int ACT, bool HAS_RES, bool PERSIST = false, typename TI = __bf16>
int launch_p12_inst(const GemmArgs& a, hipStream_t s) {
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&gemm_kernel_p12<TO, ACT, HAS_RES, PERSIST, TI>), LDS4, "cfsar_gemm")) return rc;
    hipLaunchKernelGGL((gemm_kernel_p12<TO, ACT, HAS_RES, PERSIST, TI>), dim3(PERSIST ? persistent_grid() : a.ntiles), dim3(512), LDS4, s, a);
    return cfsar_check_launch("cfsar_gemm(p12)");
}

template <typename TO, bool PERSIST = false>
int launch_p12(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_n = (a.N + BN4 - 1) / BN4;
    a.ntiles = ((a.M + BM4 - 1) / BM4) * a.tiles_n;
    const bool r = a.res != nullptr;
    if (a.row_group > 0 || a.res_mod > 0 || a.act == CFSAR_ACT_GELU_ERF || a.K % 64 != 0) return -2;
    if ((size_t)a.M * a.lda * 2 >= (1ull << 32) || (size_t)a.N * a.ldw * 2 >= (1ull << 32)) return -2;   // 32-bit offsets
    if (a.act == CFSAR_ACT_QUICKGELU) return r ? -2 : launch_p12_inst<TO, CFSAR_ACT_QUICKGELU, false, PERSIST>(a, s);
    return r ? launch_p12_inst<TO, CFSAR_ACT_NONE, true, PERSIST>(a, s) : launch_p12_inst<TO, CFSAR_ACT_NONE, false, PERSIST>(a, s);
}

// fp16 output (the bf16 mode's residual stream): only the no-activation + residual form exists
static int launch_p12_f16(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_n = (a.N + BN4 - 1) / BN4;
    a.ntiles = ((a.M + BM4 - 1) / BM4) * a.tiles_n;
    if (a.row_group > 0 || a.res_mod > 0 || a.act != CFSAR_ACT_NONE || !a.res || a.K % 64 != 0) return -2;
    if ((size_t)a.M * a.lda * 2 >= (1ull << 32) || (size_t)a.N * a.ldw * 2 >= (1ull << 32)) return -2;
    return launch_p12_inst<_Float16, CFSAR_ACT_NONE, true, true>(a, s);
}

// fp16 operands (the RN50 tower's fp16 mode: 1x1 convs and the attention pool's k / v GEMM): bias [+ fp16 residual] [+ ReLU, a runtime
// flag of the epilogue] -> fp16, or bias -> fp32
template <bool PERSIST>
static int launch_p12_h(const GemmArgs& a0, int out_dtype, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_n = (a.N + BN4 - 1) / BN4;
    a.ntiles = ((a.M + BM4 - 1) / BM4) * a.tiles_n;
    if (a.row_group > 0 || a.res_mod > 0 || a.act != CFSAR_ACT_NONE || a.K % 64 != 0) return -2;
    if ((size_t)a.M * a.lda * 2 >= (1ull << 32) || (size_t)a.N * a.ldw * 2 >= (1ull << 32)) return -2;
    if (out_dtype == CFSAR_F32) return a.res ? -2 : launch_p12_inst<float, CFSAR_ACT_NONE, false, PERSIST, _Float16>(a, s);
    return a.res ? launch_p12_inst<_Float16, CFSAR_ACT_NONE, true, PERSIST, _Float16>(a, s)
                 : launch_p12_inst<_Float16, CFSAR_ACT_NONE, false, PERSIST, _Float16>(a, s);
}
static int launch_p3_h(const GemmArgs& a0, int out_dtype, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_n = (a.N + BN2 - 1) / BN2;
    if (a.row_group > 0 || a.res_mod > 0 || a.act != CFSAR_ACT_NONE) return -2;
    if (out_dtype == CFSAR_F32) return a.res ? -2 : launch_p3_inst<_Float16, float, CFSAR_ACT_NONE, false, false>(a, s);
    return a.res ? launch_p3_inst<_Float16, _Float16, CFSAR_ACT_NONE, true, false>(a, s)
                 : launch_p3_inst<_Float16, _Float16, CFSAR_ACT_NONE, false, false>(a, s);
}

// ============================================================================================================
// Skinny fp32 GEMM (M <= 256 rows), first form (any K % 32 == 0; the second form below serves K % 128 == 0): the temporal head of ONE episode is (S+Q)*T + S = 85 rows against 512...2048-wide weights,
// and the final ViT projection of one episode's support / query set is 40 rows.  An MFMA tiling gives such a problem 4-16
// workgroups with a long serial K loop (66 us per launch); a thread-per-column FMA loop straight from global memory is bound
// by the latency of its K / 4 dependent loads (60 us).  Here a workgroup owns 8 output columns for all rows and streams K in
// chunks of KC through a 3-stage LDS ring filled by LDS-DMA (global_load_lds_dwordx4, two chunks in flight, one barrier per
// chunk): 256 threads = 8 columns x 32 row groups, thread (c, g) accumulates rows g, g + 32, ... with plain fp32 FMAs from
// conflict-free ds_read_b128s (the 16-byte chunks of LDS row r sit at position q ^ (r & (CH - 1)); the swizzle is applied on
// the DMA source side).  N / 8 workgroups (64 ... 256) spread over the chip.  Same epilogue as cfsar_gemm: bias, QuickGELU /
// GELU(erf), fp32 residual, ReLU, output-row remap.
// ============================================================================================================
template <int RPT, int KC>
__global__ __launch_bounds__(256) void skinny_gemm_f32_kernel(GemmArgs p) {
    constexpr int CH = KC / 4;                     // 16-byte chunks per LDS row
    constexpr int RB = KC * 4;                     // bytes per LDS row
    constexpr int RPI = 1024 / RB;                 // LDS rows filled by one DMA instruction
    constexpr int AR = 32 * RPT;                   // A rows of a stage; then 8 W rows; padded so that 4 waves share evenly
    constexpr int R = AR + 4 * RPI;
    constexpr int NI = R / RPI / 4;                // DMA instructions per wave per chunk
    constexpr int STG = R * RB;
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int c = t & 7, g = t >> 3;
    const int n0 = blockIdx.x * 8;
    const char* src[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = (i * 4 + wave) * RPI + lane / CH;
        const int q = (lane % CH) ^ (row & (CH - 1));
        const char* base;
        if (row < AR) base = p.A + (size_t)(row < p.M ? row : p.M - 1) * p.lda * 4;
        else if (row < AR + 8) base = p.W + (size_t)(n0 + row - AR < p.N ? n0 + row - AR : p.N - 1) * p.ldw * 4;
        else base = p.A;                           // padding rows: any valid address, never read back
        src[i] = base + q * 16;
    }
    auto issue = [&](int chunk, int stage) __attribute__((always_inline)) {
        char* dst = sk_smem + stage * STG + wave * 1024;
#pragma unroll
        for (int i = 0; i < NI; ++i) glds16(src[i] + (size_t)chunk * RB, dst + i * 4096);
    };
    const int nch = p.K / KC;
    issue(0, 0);
    if (nch > 1) issue(1, 1);
    float acc[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) acc[i] = 0.f;
    const int gsw = g & (CH - 1);
    int stage = 0;
    for (int j = 0; j < nch; ++j) {
        if (j + 1 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");     // chunk j landed, chunk j + 1 in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                           // ... for every wave; and every wave is done with chunk j - 1
        if (j + 2 < nch) issue(j + 2, stage >= 1 ? stage - 1 : 2);                      // into the stage of chunk j - 1
        const char* sa = sk_smem + stage * STG;
        const char* sw = sa + (AR + c) * RB;
#pragma unroll
        for (int s4 = 0; s4 < CH; ++s4) {
            const float4 w4 = *reinterpret_cast<const float4*>(sw + ((s4 ^ c) << 4));
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const float4 a4 = *reinterpret_cast<const float4*>(sa + (g + 32 * i) * RB + ((s4 ^ gsw) << 4));
                acc[i] = fmaf(a4.x, w4.x, acc[i]);
                acc[i] = fmaf(a4.y, w4.y, acc[i]);
                acc[i] = fmaf(a4.z, w4.z, acc[i]);
                acc[i] = fmaf(a4.w, w4.w, acc[i]);
            }
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    const int n = n0 + c;
    if (n >= p.N) return;
    const float bv = p.bias ? p.bias[n] : 0.f;
    float* outp = reinterpret_cast<float*>(p.out);
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = g + 32 * i;
        if (r >= p.M) continue;
        int orow = r + p.row_off;
        if (p.row_group > 0) orow += (r / p.row_group) * p.row_gap;
        float v = apply_act(acc[i] + bv, p.act);
        if (p.res) v += static_cast<const float*>(p.res)[(size_t)orow * p.ldr + n];
        if (p.relu) v = fmaxf(v, 0.f);
        outp[(size_t)orow * p.ldo + n] = v;
    }
}

template <int RPT, int KC>
static int launch_skinny_inst(const GemmArgs& a, hipStream_t s) {
    constexpr int lds = 3 * (32 * RPT + 4 * (1024 / (KC * 4))) * KC * 4;
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&skinny_gemm_f32_kernel<RPT, KC>), lds, "cfsar_gemm(skinny f32)")) return rc;
    hipLaunchKernelGGL((skinny_gemm_f32_kernel<RPT, KC>), dim3((unsigned)((a.N + 7) / 8)), dim3(256), lds, s, a);
    return cfsar_check_launch("cfsar_gemm(skinny f32)");
}

// ---- Skinny fp32 GEMM, second form (K % 128 == 0): register tiles + K split over the workgroup's four waves.
// The form above gives every thread ONE output column: each thread re-reads the whole A chunk from LDS (2 MB of ds_read_b128 per
// workgroup for 8 columns) and the launch is a chain of K / 64 barriers on N / 8 workgroups: 20 us at K = 512, 70 us at K = 2 048
// (tools/skinny_time.py).  Here a workgroup owns an (8 TR) x 32 output tile, wave w the K range [w K / 4, (w + 1) K / 4) of it: a
// lane accumulates TR x 4 outputs (one A fragment feeds 4 columns, one W fragment TR rows), streams its quarter of K through a
// wave-private 3-stage LDS ring (LDS-DMA, 32-float chunks: no workgroup barrier inside the K loop), and the four partial tiles are
// summed through LDS in wave order (deterministic).  Same epilogue.
template <int TR>
__global__ __launch_bounds__(256) void skinny2_gemm_f32_kernel(GemmArgs p) {
    constexpr int AR = 8 * TR;                     // A rows of the tile
    constexpr int ROWS = AR + 32;                  // + 32 W rows, 128 bytes (32 floats) each
    constexpr int NI = ROWS / 8;                   // DMA instructions per chunk (8 rows each)
    constexpr int STG = ROWS * 128;
    extern __shared__ __attribute__((aligned(16))) char s2_smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tr = lane >> 3, tc = lane & 7;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * AR;
    const int kq = p.K >> 2;                       // floats of K per wave
    char* ring = s2_smem + wave * (3 * STG);
    const char* src[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int row = i * 8 + (lane >> 3);
        const int q = (lane & 7) ^ ((row >> 1) & 7);
        const char* base;
        if (row < AR) base = p.A + (size_t)(m0 + row < p.M ? m0 + row : p.M - 1) * p.lda * 4;
        else base = p.W + (size_t)(n0 + row - AR < p.N ? n0 + row - AR : p.N - 1) * p.ldw * 4;
        src[i] = base + (size_t)wave * kq * 4 + q * 16;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
    auto issue = [&](int chunk, int stage) __attribute__((always_inline)) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)stage * STG);
#pragma unroll
        for (int i = 0; i < NI; ++i) glds16_asm(src[i] + (size_t)chunk * 128, dst + i * 1024);
    };
    const int nch = kq / 32;
    issue(0, 0);
    if (nch > 1) issue(1, 1);
    float acc[TR][4];
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    // lane (tr, tc): rows tr + 8 i, columns tc + 8 j.  16-byte chunk s4 of LDS row r sits at position s4 ^ ((r >> 1) & 7).
    int offA[TR], swA[TR], offW[4], swW[4];
#pragma unroll
    for (int i = 0; i < TR; ++i) { const int r = tr + 8 * i; offA[i] = r * 128; swA[i] = (r >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int r = AR + tc + 8 * j; offW[j] = r * 128; swW[j] = (r >> 1) & 7; }
    int stage = 0;
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");     // this wave's chunk c landed, c + 1 in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (c + 2 < nch) issue(c + 2, stage >= 1 ? stage - 1 : 2);       // the stage of chunk c - 1: its reads were consumed by the FMAs
        const char* sb = ring + stage * STG;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            float4 a4[TR], w4[4];
#pragma unroll
            for (int i = 0; i < TR; ++i) a4[i] = *reinterpret_cast<const float4*>(sb + offA[i] + ((s4 ^ swA[i]) << 4));
#pragma unroll
            for (int j = 0; j < 4; ++j) w4[j] = *reinterpret_cast<const float4*>(sb + offW[j] + ((s4 ^ swW[j]) << 4));
#pragma unroll
            for (int i = 0; i < TR; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = fmaf(a4[i].x, w4[j].x, acc[i][j]);
                    acc[i][j] = fmaf(a4[i].y, w4[j].y, acc[i][j]);
                    acc[i][j] = fmaf(a4[i].z, w4[j].z, acc[i][j]);
                    acc[i][j] = fmaf(a4[i].w, w4[j].w, acc[i][j]);
                }
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    // partial tiles -> LDS (each wave into its own ring: no hazard with another wave's DMA), then every thread sums 4 waves in order
    float* red = reinterpret_cast<float*>(ring);                          // [AR][32]
#pragma unroll
    for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[(tr + 8 * i) * 32 + tc + 8 * j] = acc[i][j];
    __syncthreads();
    for (int e = t; e < AR * 32; e += 256) {
        const int r = e >> 5, cidx = e & 31;
        const int m = m0 + r, n = n0 + cidx;
        if (m >= p.M || n >= p.N) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += reinterpret_cast<const float*>(s2_smem + w * (3 * STG))[e];
        int orow = m + p.row_off;
        if (p.row_group > 0) orow += (m / p.row_group) * p.row_gap;
        v = apply_act(v + (p.bias ? p.bias[n] : 0.f), p.act);
        if (p.res) v += static_cast<const float*>(p.res)[(size_t)orow * p.ldr + n];
        if (p.relu) v = fmaxf(v, 0.f);
        reinterpret_cast<float*>(p.out)[(size_t)orow * p.ldo + n] = v;
    }
}

template <int TR>
static int launch_skinny2_inst(const GemmArgs& a, hipStream_t s) {
    constexpr int lds = 4 * 3 * (8 * TR + 32) * 128;
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&skinny2_gemm_f32_kernel<TR>), lds, "cfsar_gemm(skinny2 f32)")) return rc;
    hipLaunchKernelGGL((skinny2_gemm_f32_kernel<TR>), dim3((unsigned)((a.N + 31) / 32), (unsigned)((a.M + 8 * TR - 1) / (8 * TR))), dim3(256), lds, s, a);
    return cfsar_check_launch("cfsar_gemm(skinny2 f32)");
}

// M <= 256, K % 32 == 0 (checked by the caller).  64-float chunks while three stages fit the LDS (M <= 128), else 32.
static int launch_skinny_f32(const GemmArgs& a, hipStream_t s) {
    if (a.K % 128 == 0 && !(GDBG(a) & 65536)) {    // register-tiled, K split over the waves; 16-row tiles while they stay within one round of workgroups
        const long wg16 = (long)((a.M + 15) / 16) * ((a.N + 31) / 32);
        return wg16 <= 256 ? launch_skinny2_inst<2>(a, s) : launch_skinny2_inst<4>(a, s);
    }
    const int rpt = (a.M + 31) / 32;
    if (a.K % 64 == 0) {
        switch (rpt) {
            case 1: return launch_skinny_inst<1, 64>(a, s);
            case 2: return launch_skinny_inst<2, 64>(a, s);
            case 3: return launch_skinny_inst<3, 64>(a, s);
            case 4: return launch_skinny_inst<4, 64>(a, s);
            default: break;
        }
    }
    switch (rpt) {
        case 1: case 2: return launch_skinny_inst<2, 32>(a, s);
        case 3: case 4: return launch_skinny_inst<4, 32>(a, s);
        case 5: case 6: return launch_skinny_inst<6, 32>(a, s);
        default: return launch_skinny_inst<8, 32>(a, s);
    }
}

}  // namespace

#ifdef CFSAR_DEV
// dev builds only (include/clipfsar_hip_dev.h): force a kernel variant / ablation bits for in-process A/B; (0, 0) = product behaviour
extern "C" void cfsar_debug_set_gemm_variant(int variant, int dbg) { g_variant_override = variant; g_dbg_override = dbg; }
#endif

extern "C" int cfsar_gemm_ex(const void* A, const void* W, void* out, const float* bias, const void* residual, int M,
                             int N, int K, int lda, int ldw, int ldo, int ldr, int in_dtype, int out_dtype, int act,
                             int row_group, int row_gap, int row_off, int res_mod, int res_off, int res_dtype, int relu,
                             cfsar_stream_t stream) {
    CFSAR_REQUIRE(A && W && out, "cfsar_gemm: null operand");
    CFSAR_REQUIRE(M > 0 && N > 0 && K > 0, "cfsar_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    CFSAR_REQUIRE(in_dtype == CFSAR_F32 || in_dtype == CFSAR_BF16 || in_dtype == CFSAR_F16, "cfsar_gemm: bad in_dtype %d", in_dtype);
    CFSAR_REQUIRE(out_dtype == CFSAR_F32 || out_dtype == CFSAR_BF16 || out_dtype == CFSAR_F16, "cfsar_gemm: bad out_dtype %d", out_dtype);
    // fp16 output = a residual-stream update: 16-bit operands, no activation, a residual.  fp16 OPERANDS (the fp16 mode's patch embedding and
    // small-shape fallbacks; its block GEMMs are cfsar_gemm_lnfold / cfsar_gemm_residual_stats) go to fp16 or fp32 outputs.
    CFSAR_REQUIRE(out_dtype != CFSAR_F16 || (in_dtype != CFSAR_F32 && act == CFSAR_ACT_NONE && (residual || in_dtype == CFSAR_F16)),
                  "cfsar_gemm: fp16 output needs 16-bit operands, no activation and (bf16 operands) a residual");
    CFSAR_REQUIRE(in_dtype != CFSAR_F16 || out_dtype != CFSAR_BF16, "cfsar_gemm: fp16 operands write fp16 or fp32 outputs");
    const int esz = in_dtype == CFSAR_F32 ? 4 : 2;
    const int bk = ROWB / esz;
    CFSAR_REQUIRE(K % bk == 0, "cfsar_gemm: K=%d must be a multiple of %d for this dtype", K, bk);
    CFSAR_REQUIRE(N % 4 == 0, "cfsar_gemm: N=%d must be a multiple of 4", N);
    CFSAR_REQUIRE((lda * esz) % 16 == 0 && (ldw * esz) % 16 == 0, "cfsar_gemm: lda/ldw rows must be 16-byte aligned");
    CFSAR_REQUIRE(lda >= K && ldw >= K && ldo >= N, "cfsar_gemm: leading dimension too small");
    CFSAR_REQUIRE((ldo * (out_dtype == CFSAR_F32 ? 4 : 2)) % 8 == 0, "cfsar_gemm: ldo alignment");
    CFSAR_REQUIRE(!residual || (ldr >= N && ldr % 4 == 0), "cfsar_gemm: bad ldr");
    CFSAR_REQUIRE(act >= 0 && act <= 2, "cfsar_gemm: bad act %d", act);
    CFSAR_REQUIRE(res_dtype == CFSAR_F32 || res_dtype == CFSAR_BF16 || res_dtype == CFSAR_F16, "cfsar_gemm: bad res_dtype %d", res_dtype);
    GemmArgs a;
    a.A = static_cast<const char*>(A);
    a.W = static_cast<const char*>(W);
    a.out = out;
    a.bias = bias;
    a.res = residual;
    a.res_kind = res_dtype == CFSAR_BF16 ? 1 : (res_dtype == CFSAR_F16 ? 2 : 0);
    a.relu = relu;
    a.M = M; a.N = N; a.K = K;
    a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.ldr = ldr;
    a.act = act;
    a.row_group = row_group; a.row_gap = row_gap; a.row_off = row_off;
    a.res_mod = res_mod; a.res_off = res_off;
    a.tiles_n = (N + BN - 1) / BN;
    a.conv_H = a.conv_W = a.conv_lgC = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Kernel choice.  Product builds take the measured policy below (`forced` is the constant 0 and its branches fold away); dev
    // builds (CFSAR_DEV, include/clipfsar_hip_dev.h) can force a variant for in-process A/B: 1 = v1 (128x128, also the fp32 path),
    // 2 = p3 (256x128, asm LDS-DMA), 10 / 11 = p10 (one wave per SIMD) / persistent, 12 / 13 = p12 (two waves per SIMD) /
    // persistent, 14 / 15 = skinny fp32 kernel always / never, 20+ = the gemm_vit.hip kernel (see there).
#ifdef CFSAR_DEV
    a.dbg = g_dbg_override;
    const int forced = g_variant_override;
#else
    constexpr int forced = 0;
#endif
    const long tiles4 = (long)((M + BM4 - 1) / BM4) * ((N + BN4 - 1) / BN4);
    if (in_dtype == CFSAR_F16) {                  // v_mfma_f32_32x32x16_f16: the ViT tower's patch-embed scatter and the RN50 tower's fp16 mode
        if (out_dtype == CFSAR_F16 && M >= 1024 && (row_group > 0 || res_mod > 0) && residual && !relu) {
            GemmArgs b = a;
            b.tiles_n = (N + BN2 - 1) / BN2;
            return launch_p3_inst<_Float16, _Float16, CFSAR_ACT_NONE, true, true>(b, s);
        }
        // same tile policy as the bf16 operands below (the residual forms stay off the ViT-block kernel: its fp16 residual epilogue adds in
        // packed fp16 -- a second rounding the RN50 mode's budget has no room for, tools/numerics_lab_rn.py dr=1)
        if (row_off == 0 && (res_dtype == CFSAR_F16 || !residual) && forced == 0) {
            int rc = -2;
            if (tiles4 >= 512 && N >= 256) rc = launch_p12_h<true>(a, out_dtype, s);
            else if (N >= 256 && (tiles4 >= 240 || (tiles4 >= 128 && K >= 2048))) rc = launch_p12_h<false>(a, out_dtype, s);
            if (rc == -2 && M >= 1024) rc = launch_p3_h(a, out_dtype, s);
            if (rc != -2) return rc;
        }
        return out_dtype == CFSAR_F16 ? launch<_Float16, _Float16>(a, s) : launch<_Float16, float>(a, s);
    }
    // fp32, at most 192 rows: the skinny kernel (the temporal head and the final ViT projection of one or two episodes)
    if (in_dtype == CFSAR_F32 && out_dtype == CFSAR_F32 && res_dtype == CFSAR_F32 && res_mod == 0 && K % 32 == 0 &&
        lda % 4 == 0 && ldw % 4 == 0 && M <= 256 && (forced == 14 || (forced == 0 && M <= 192)))
        return launch_skinny_f32(a, s);
    // fp32 up to a few thousand rows with K % 128 == 0 (the temporal head at batch scale: 16 episodes = 1 360 rows against 512 ... 2 048-wide
    // weights) where the fp32-MFMA kernel gets fewer than 64 of its 128 x 128 tiles (N = 512): the register-tiled VALU kernel again, 43 x 16 workgroups
    if (in_dtype == CFSAR_F32 && out_dtype == CFSAR_F32 && res_dtype == CFSAR_F32 && res_mod == 0 && K % 128 == 0 && lda % 4 == 0 &&
        ldw % 4 == 0 && M <= 4096 && forced == 0 && (long)((M + 127) / 128) * ((N + 127) / 128) < 64)
        return launch_skinny2_inst<4>(a, s);       // (same-process A/B at 1 360 rows, us: N = 512, K = 512: 26 vs 39; K = 2 048: 80 vs 134; 640 x 512 x 768: 24 vs 54;
                                                   //  N >= 1 536 stays on the fp32-MFMA kernel: 41 vs 73-90; skinny_ab.py (archived probe))
    // The ViT-block GEMMs at batch scale (bias [+ QuickGELU] -> bf16, or bias + fp16 residual -> fp16; >= two 256x256 tiles per
    // CU): the persistent kernel of gemm_vit.hip whose operand pipeline runs through the epilogues.  Dev builds: variant
    // 20 + 4 * opath + store forces it on any shape; dbg bit 256 = column-fastest tile walk, bits 9-11 = band group (see below).
    if (in_dtype == CFSAR_BF16 && row_group == 0 && res_mod == 0 && row_off == 0 &&
        ((forced == 0 && kUseVitKernel && tiles4 >= 512) || (forced >= 20 && forced < 44))) {
        VitGemmCall c;
        c.A = A; c.W = W; c.out = out; c.bias = bias; c.res = residual;
        c.rowstats = nullptr; c.cvec = nullptr; c.stats_out = nullptr;
        c.part = nullptr; c.part_slots = 0; c.part_eps = 0.f;
        c.M = M; c.N = N; c.K = K; c.lda = lda; c.ldw = ldw; c.ldo = ldo; c.ldr = ldr;
        c.in_dtype = in_dtype; c.out_dtype = out_dtype; c.res_dtype = res_dtype; c.act = act; c.relu = relu;
        c.opath = cfsar_vit_policy_opath(K);
        c.store = residual ? 0 : 2;
        c.group = kVitGroup; c.colfast = kVitColfast; c.dbg = 0;
        c.hb_tokens = 0; c.hb_heads = 0; c.ha_tokens = 0;
#ifdef CFSAR_DEV
        if (forced >= 20 && forced < 31) { c.opath = (forced - 20) >> 2; c.store = (forced - 20) & 3; }
        else if (forced >= 40) { c.opath = 5; c.store = forced - 40; }      // 40 / 42: one wave per SIMD, 128 x 128 wave tiles (gemm_vit1w.hip)
        else if (forced >= 36) { c.opath = 4; c.store = forced - 36; }      // 36 / 38: the two-workgroups-per-CU kernel (gemm_vit4.hip), plain / write-through stores
        else if (forced >= 31) { c.opath = 2; c.store = forced - 28; }      // 31..34: store-policy A/B on the early-DMA path
        c.dbg = g_dbg_override;
        if (g_dbg_override & 256) c.colfast = 1;
        constexpr int groups[8] = {0, 4, 16, 2, 32, 1, 3, 6};
        if ((g_dbg_override >> 9) & 7) c.group = groups[(g_dbg_override >> 9) & 7];
#endif
        const int rc = cfsar_gemm_vit_try(c, s);
        if (rc != -2) return rc;
    }
    // p12 measured fastest on all four ViT GEMMs (M = 252 160, same box, interleaved: QKV 888 vs 960 us (p10), out_proj 457 vs
    // 476, c_fc 1 288 vs 1 409 (p6), c_proj 1 143 vs 1 178) and on the RN50 1x1 convs with N >= 256 (tools/rn_gemm_ab.py).
    // 13 = the persistent form (one workgroup per CU walking its tiles: no retire -> dispatch gap; +3 % on QKV / c_fc), the auto
    // choice from two tiles per CU on; 12 = one workgroup per tile.
    if (in_dtype == CFSAR_BF16 && (forced == 13 || (forced == 0 && tiles4 >= 512 && N >= 256))) {
        const int rc = out_dtype == CFSAR_BF16 ? launch_p12<__bf16, true>(a, s)
                       : out_dtype == CFSAR_F16 ? launch_p12_f16(a, s) : launch_p12<float, true>(a, s);
        if (rc != -2) return rc;
    }
    // (long-K GEMMs keep the 256x256 tile down to half a round of tiles: c_proj of ONE episode, 186 tiles, 77 vs 97 us on p3)
    if (in_dtype == CFSAR_BF16 && (forced == 12 || (forced == 0 && N >= 256 && (tiles4 >= 240 || (tiles4 >= 128 && K >= 2048))))) {
        const int rc = out_dtype == CFSAR_BF16 ? launch_p12<__bf16>(a, s)
                       : out_dtype == CFSAR_F16 ? launch_p12_f16(a, s) : launch_p12<float>(a, s);
        if (rc != -2) return rc;
    }
    if (out_dtype == CFSAR_F16) {                    // only p12, p3 and v1 are instantiated for the fp16 stream
        if (M >= 1024) {
            const int rc = launch_p3_f16(a, s);
            if (rc != -2) return rc;
        }
        return launch<__bf16, _Float16>(a, s);
    }
    if (in_dtype == CFSAR_BF16 && (forced == 10 || forced == 11)) {      // 11 = p10 persistent (one workgroup per CU)
        const int rc = forced == 10 ? (out_dtype == CFSAR_BF16 ? launch_p10<__bf16, false>(a, s) : launch_p10<float, false>(a, s))
                                    : (out_dtype == CFSAR_BF16 ? launch_p10<__bf16, true>(a, s) : launch_p10<float, true>(a, s));
        if (rc != -2) return rc;
    }
    const bool use_p3 = (forced != 0 && forced != 1) || (forced == 0 && in_dtype == CFSAR_BF16 && M >= 1024);
    if (use_p3) {
        int rc;
        if (in_dtype == CFSAR_BF16)
            rc = out_dtype == CFSAR_BF16 ? launch_p3<__bf16, __bf16>(a, s) : launch_p3<__bf16, float>(a, s);
        else
            rc = out_dtype == CFSAR_BF16 ? launch_p3<float, __bf16>(a, s) : launch_p3<float, float>(a, s);
        if (rc != -2) return rc;
    }
    if (in_dtype == CFSAR_BF16)
        return out_dtype == CFSAR_BF16 ? launch<__bf16, __bf16>(a, s) : launch<__bf16, float>(a, s);
    return out_dtype == CFSAR_BF16 ? launch<float, __bf16>(a, s) : launch<float, float>(a, s);
}

extern "C" int cfsar_patch_embed(const float* frames, const void* W, int w_dtype, const float* pos, const float* cls, void* x, int x_dtype,
                                 int F, int H, int Wd, int P, int D, int ldw, cfsar_stream_t stream) {
    CFSAR_REQUIRE(frames && W && pos && cls && x, "cfsar_patch_embed: null pointer");
    CFSAR_REQUIRE(P == 16 || P == 14, "cfsar_patch_embed: patch size %d (the fused gather exists for 16 x 16 and 14 x 14 patches; other sizes: cfsar_im2col_patches + cfsar_gemm_ex)", P);
    CFSAR_REQUIRE(F > 0 && H > 0 && Wd > 0 && H % P == 0 && Wd % P == 0 && Wd % 2 == 0, "cfsar_patch_embed: bad geometry F=%d H=%d W=%d", F, H, Wd);
    const int Kslots = P == 16 ? 768 : 704;             // P = 14: padded rows, k' = (c 14 + dy) 16 + dx (see the header)
    CFSAR_REQUIRE(D > 0 && D % 4 == 0 && ldw >= Kslots && ldw % 8 == 0, "cfsar_patch_embed: bad D=%d / ldw=%d (>= %d)", D, ldw, Kslots);
    CFSAR_REQUIRE(((size_t)frames & 7) == 0, "cfsar_patch_embed: frames must be 8-byte aligned");
    CFSAR_REQUIRE(w_dtype == CFSAR_BF16 || w_dtype == CFSAR_F16, "cfsar_patch_embed: weights must be bf16 or fp16, got %d", w_dtype);
    CFSAR_REQUIRE(x_dtype == CFSAR_F16, "cfsar_patch_embed: x must be the fp16 residual stream, got dtype %d", x_dtype);
    PatchArgs a;
    a.frames = frames; a.cls = cls; a.H = H; a.Wd = Wd; a.gw = Wd / P; a.npatch = (H / P) * (Wd / P); a.F = F;
    CFSAR_REQUIRE((long long)F * a.npatch < (1ll << 31) / 2 && (long long)F * 3 * H * Wd < (1ll << 40), "cfsar_patch_embed: too many frames");
    GemmArgs& g = a.g;
    g.A = nullptr; g.W = static_cast<const char*>(W); g.out = x; g.bias = nullptr; g.res = pos; g.res_kind = 0; g.relu = 0;
    g.M = F * a.npatch; g.N = D; g.K = Kslots;
    g.lda = Kslots; g.ldw = ldw; g.ldo = D; g.ldr = D;
    g.act = CFSAR_ACT_NONE;
    g.row_group = a.npatch; g.row_gap = 1; g.row_off = 1; g.res_mod = a.npatch; g.res_off = 1;
    g.tiles_n = (D + BN2 - 1) / BN2; g.ntiles = 0;
    g.conv_H = g.conv_W = g.conv_lgC = 0;
#ifdef CFSAR_DEV
    g.dbg = 0;
#endif
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (P == 14) return w_dtype == CFSAR_F16 ? launch_patch_embed<_Float16, _Float16, 14>(a, s) : launch_patch_embed<__bf16, _Float16, 14>(a, s);
    return w_dtype == CFSAR_F16 ? launch_patch_embed<_Float16, _Float16>(a, s) : launch_patch_embed<__bf16, _Float16>(a, s);
}

extern "C" int cfsar_gemm(const void* A, const void* W, void* out, const float* bias, const float* residual, int M,
                          int N, int K, int lda, int ldw, int ldo, int ldr, int in_dtype, int out_dtype, int act,
                          int row_group, int row_gap, int row_off, int res_mod, int res_off, cfsar_stream_t stream) {
    return cfsar_gemm_ex(A, W, out, bias, residual, M, N, K, lda, ldw, ldo, ldr, in_dtype, out_dtype, act, row_group, row_gap,
                         row_off, res_mod, res_off, CFSAR_F32, 0, stream);
}

#ifdef CFSAR_DEV
static int g_no_direct_conv = 0;        // dev A/B: cfsar_debug_set_direct_conv(0) routes the narrow convs through the implicit GEMM again
extern int g_direct_conv_dbg;          // conv.hip
extern "C" void cfsar_debug_set_direct_conv(int on) { g_no_direct_conv = !(on & 1); g_direct_conv_dbg = on >> 8; }
#else
constexpr int g_no_direct_conv = 0;
#endif
// Implicit-GEMM 3x3 convolution (pad 1, stride 1) on NHWC bf16 activations: out[f,y,x,:] = [relu](W . patch(f,y,x) + bias
// (+ residual)); W is [Cout, ldw] tap-major ((ky*3+kx)*C + c), zero-padded to ldw = round_up(9*C, 64).  See the header.
extern "C" int cfsar_conv3x3_nhwc(const void* in, const void* W, void* out, const float* bias, const void* residual, int F,
                                  int H, int Wd, int C, int Cout, int ldw, int ldo, int ldr, int out_dtype, int res_dtype,
                                  int relu, cfsar_stream_t stream) {
    CFSAR_REQUIRE(in && W && out, "cfsar_conv3x3_nhwc: null operand");
    CFSAR_REQUIRE(F > 0 && H > 0 && Wd > 0 && Cout > 0 && Cout % 4 == 0, "cfsar_conv3x3_nhwc: bad shape");
    CFSAR_REQUIRE(C >= 8 && (C & (C - 1)) == 0, "cfsar_conv3x3_nhwc: C=%d must be a power of two >= 8", C);
    CFSAR_REQUIRE(ldw % 64 == 0 && ldw >= 9 * C && ldw < 16 * C + 64, "cfsar_conv3x3_nhwc: ldw=%d must be round_up(9*C, 64)", ldw);
    CFSAR_REQUIRE(out_dtype == CFSAR_F32 || out_dtype == CFSAR_BF16 || out_dtype == CFSAR_F16, "cfsar_conv3x3_nhwc: bad out_dtype %d", out_dtype);
    CFSAR_REQUIRE(res_dtype == CFSAR_F32 || res_dtype == CFSAR_BF16 || res_dtype == CFSAR_F16, "cfsar_conv3x3_nhwc: bad res_dtype %d", res_dtype);
    CFSAR_REQUIRE(ldo >= Cout && (!residual || (ldr >= Cout && ldr % 4 == 0)), "cfsar_conv3x3_nhwc: bad ldo/ldr");
    const long long M = (long long)F * H * Wd;
    CFSAR_REQUIRE(M * C * 2 < (1ll << 32) && (long long)Cout * ldw * 2 < (1ll << 32), "cfsar_conv3x3_nhwc: tensor too large for 32-bit offsets");
    GemmArgs a;
    a.A = static_cast<const char*>(in);
    a.W = static_cast<const char*>(W);
    a.out = out;
    a.bias = bias;
    a.res = residual;
    a.res_kind = res_dtype == CFSAR_BF16 ? 1 : (res_dtype == CFSAR_F16 ? 2 : 0);
    a.relu = relu;
    a.M = (int)M; a.N = Cout; a.K = ldw;
    a.lda = C; a.ldw = ldw; a.ldo = ldo; a.ldr = ldr;
    a.act = CFSAR_ACT_NONE;
    a.row_group = a.row_gap = a.row_off = a.res_mod = a.res_off = 0;
    a.tiles_n = (Cout + BN4 - 1) / BN4;
    a.ntiles = (int)((M + BM4 - 1) / BM4) * a.tiles_n;
#ifdef CFSAR_DEV
    a.dbg = 0;
#endif
    a.conv_H = H; a.conv_W = Wd;
    a.conv_lgC = 0;
    while ((1 << a.conv_lgC) < C) ++a.conv_lgC;
    hipStream_t s = static_cast<hipStream_t>(stream);
    constexpr int conv_variant = 0;     // 3 / 4: force the 256x128 / 256x64 tile (A/B history: tools/rn_gemm_ab.py)
    // Cin, Cout in {32, 64}: the direct kernel (conv.hip: the input ring in LDS, weights in registers) instead of the 9-fold gather
    if (out_dtype != CFSAR_F32 && !residual && conv_variant == 0 && !g_no_direct_conv) {
        const int rc = cfsar_conv3x3_direct(in, W, out, bias, F, H, Wd, C, Cout, ldw, ldo, relu, out_dtype == CFSAR_F16, s);
        if (rc != -2) return rc;
    }
    if (out_dtype == CFSAR_F16) {                  // fp16 NHWC in and out (the RN50 tower's fp16 mode): same tiles by Cout
        if (Cout <= 64) {
            a.tiles_n = (Cout + 63) / 64;
            return residual ? launch_p3_inst<_Float16, _Float16, CFSAR_ACT_NONE, true, false, true, true>(a, s)
                            : launch_p3_inst<_Float16, _Float16, CFSAR_ACT_NONE, false, false, true, true>(a, s);
        }
        if (Cout <= 128) {
            a.tiles_n = (Cout + BN2 - 1) / BN2;
            return residual ? launch_p3_inst<_Float16, _Float16, CFSAR_ACT_NONE, true, false, true>(a, s)
                            : launch_p3_inst<_Float16, _Float16, CFSAR_ACT_NONE, false, false, true>(a, s);
        }
        return residual ? launch_p10_inst<_Float16, CFSAR_ACT_NONE, true, false, true, _Float16>(a, s)
                        : launch_p10_inst<_Float16, CFSAR_ACT_NONE, false, false, true, _Float16>(a, s);
    }
    if (out_dtype == CFSAR_BF16 && (conv_variant == 4 || (conv_variant == 0 && Cout <= 64))) {     // 256x64 tile
        a.tiles_n = (Cout + 63) / 64;
        return residual ? launch_p3_inst<__bf16, __bf16, CFSAR_ACT_NONE, true, false, true, true>(a, s)
                        : launch_p3_inst<__bf16, __bf16, CFSAR_ACT_NONE, false, false, true, true>(a, s);
    }
    if (out_dtype == CFSAR_BF16 && (conv_variant == 3 || (conv_variant == 0 && Cout <= 128))) {   // narrow outputs: 256x128 tile
        a.tiles_n = (Cout + BN2 - 1) / BN2;
        return residual ? launch_p3_inst<__bf16, __bf16, CFSAR_ACT_NONE, true, false, true>(a, s)
                        : launch_p3_inst<__bf16, __bf16, CFSAR_ACT_NONE, false, false, true>(a, s);
    }
    if (out_dtype == CFSAR_BF16)
        return residual ? launch_p10_inst<__bf16, CFSAR_ACT_NONE, true, false, true>(a, s)
                        : launch_p10_inst<__bf16, CFSAR_ACT_NONE, false, false, true>(a, s);
    return residual ? launch_p10_inst<float, CFSAR_ACT_NONE, true, false, true>(a, s)
                    : launch_p10_inst<float, CFSAR_ACT_NONE, false, false, true>(a, s);
}
