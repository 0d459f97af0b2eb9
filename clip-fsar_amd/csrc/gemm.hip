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

#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int ROWB = 128;                       // bytes of K per tile row per stage
constexpr int STAGE_BYTES = (BM + BN) * ROWB;   // 32 KiB
constexpr int NTHREADS = 256;

struct GemmArgs {
    const char* A;
    const char* W;
    void* out;
    const float* bias;
    const void* res;      // residual, fp32 or (res_bf16) bf16
    int res_bf16;
    int relu;             // ReLU applied last (after bias / activation / residual): conv+BN(+identity)+ReLU of the RN50 tower
    int M, N, K;
    int lda, ldw, ldo, ldr;
    int act;
    int row_group, row_gap, row_off, res_mod, res_off;
    int tiles_n;
    int ntiles;    // persistent p6: total number of 256x256 tiles
    int stagger;   // p4: first-round phase offset in units of s_sleep(127) (~4 us)
    unsigned long long* trace;   // dev tool: per-tile phase timestamps (s_memtime), 8 slots per tile; normally NULL
    int dbg;   // ablation bits (env CFSAR_GEMM_DEBUG): 1 = no in-loop DMA, 2 = no in-loop barrier, 4 = no epilogue
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

__device__ __forceinline__ float4 load_res4(const void* res, int is_bf16, size_t elem_off) {
    if (is_bf16) {
        const bf16x4 r = *reinterpret_cast<const bf16x4*>(static_cast<const __bf16*>(res) + elem_off);
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
                    const float4 rv = load_res4(p.res, p.res_bf16, (size_t)rrow * p.ldr + n);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                }
                if (p.relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
                }
                if constexpr (sizeof(TO) == 2) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.out) + (size_t)orow * p.ldo + n) = o;
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
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, wf[ni]), __builtin_bit_cast(bf16x8, xf[mi]), acc[mi][ni], 0, 0, 0);
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
template <typename TO, int ACT, bool HAS_RES, bool REMAP, bool FULL, int NMI = 2>
__device__ __forceinline__ void epilogue_lds(f32x16 (*acc)[2], const GemmArgs& p, int mbase, int nbase, int lane,
                                             char* wbuf) {
    const int lr = lane & 31, hi = lane >> 5;
    const int rsub = lane >> 4, cc = lane & 15;
    const int M = p.M, N = p.N, ldo = p.ldo, ldr = p.ldr, row_off = p.row_off;
    const int n = nbase + cc * 4;
    const bool nvalid = n < N;
    const int nc = nvalid ? n : N - 4;                       // clamped column: loads stay in bounds, stores are predicated
    const void* resp = p.res;
    const int res_bf16 = p.res_bf16, relu = p.relu;
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
        if constexpr (HAS_RES) rv[it] = load_res4(resp, res_bf16, (size_t)rrow * ldr + nc);
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
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
                *reinterpret_cast<bf16x4*>(outp + ooff[it]) = o;
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
// MFMAs of sub-step s execute.
template <typename TI>
__device__ __forceinline__ void mma_slice_db(f32x16 (&acc)[2][2], const char* sX, const char* sW, const int (&offX)[2],
                                             const int (&offW)[2], const int (&sxX)[2], const int (&sxW)[2], int hi) {
    uint4 xf[2][2], wf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        xf[0][i] = *reinterpret_cast<const uint4*>(sX + offX[i] + ((hi ^ sxX[i]) << 4));
        wf[0][i] = *reinterpret_cast<const uint4*>(sW + offW[i] + ((hi ^ sxW[i]) << 4));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s < 3) {
            const int c = 2 * (s + 1) + hi;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                xf[nxt][i] = *reinterpret_cast<const uint4*>(sX + offX[i] + ((c ^ sxX[i]) << 4));
                wf[nxt][i] = *reinterpret_cast<const uint4*>(sW + offW[i] + ((c ^ sxW[i]) << 4));
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the next sub-step's reads AHEAD of this sub-step's MFMAs
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if constexpr (sizeof(TI) == 2) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[cur][ni]),
                                                                          __builtin_bit_cast(bf16x8, xf[cur][mi]),
                                                                          acc[mi][ni], 0, 0, 0);
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
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, wf[ni]), __builtin_bit_cast(bf16x8, xf[mi]), acc[mi][ni], 0, 0, 0);
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

template <typename TI, typename TO, int ACT, bool HAS_RES, bool REMAP>
__global__ __launch_bounds__(NTHREADS2) void gemm_kernel_p3(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = ROWB / (int)sizeof(TI);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int tm = lin / p.tiles_n, tn = lin - tm * p.tiles_n;
    const int m0 = tm * BM2, n0 = tn * BN2;

    // staging: X tile = 32 instructions of 8 rows, W tile = 16; wave w issues X instr {w, w+8, w+16, w+24}, W {w, w+8}
    const char* srcX[4];
    const char* srcW[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);
        int gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        srcX[i] = p.A + ((size_t)gm * p.lda) * sizeof(TI) + chunk * 16;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
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
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_asm(srcX[i] + koff, base + i * 8192);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_asm(srcW[i] + koff, base + BM2 * ROWB + i * 8192);
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

    const int nk = p.K / BK;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int st = 0;                      // stage holding slice kt
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // slice kt landed, kt+1 may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(p.dbg & 2)) __syncthreads();   // everyone's part of slice kt is in LDS; everyone is done reading slice kt-1
        int st2 = st + 2;
        st2 = st2 >= NSTAGE2 ? st2 - NSTAGE2 : st2;
        if (kt + 2 < nk && !(p.dbg & 1)) issue(st2, kt + 2);
        const char* sX = smem + st * STAGE2;
        mma_slice_db<TI>(acc, sX, sX + BM2 * ROWB, offX, offW, sxX, sxW, hi);
        st = st + 1 >= NSTAGE2 ? 0 : st + 1;
    }
    __syncthreads();                 // every wave is done with the ring: reuse it as per-wave transpose buffers
    const int mb = m0 + wm * 64, nb = n0 + wn * 64;
    f32x16 (*accp)[2] = acc;
    if (p.dbg & 4) {
        if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.out)[0] = acc[1][1][3];
        return;
    }
    const bool full = mb + 64 <= p.M && nb + 64 <= p.N;
    if constexpr (kBf16PackedEpilogue && sizeof(TO) == 2 && !HAS_RES && !REMAP) {
        if (full) epilogue_lds_bf16<ACT, true>(accp, p, mb, nb, lane, smem + wave * EPI_WAVE_BYTES);
        else epilogue_lds_bf16<ACT, false>(accp, p, mb, nb, lane, smem + wave * EPI_WAVE_BYTES);
    } else {
        if (full) epilogue_lds<TO, ACT, HAS_RES, REMAP, true>(accp, p, mb, nb, lane, smem + wave * EPI_WAVE_BYTES);
        else epilogue_lds<TO, ACT, HAS_RES, REMAP, false>(accp, p, mb, nb, lane, smem + wave * EPI_WAVE_BYTES);
    }
}

template <typename TI, typename TO, int ACT, bool HAS_RES, bool REMAP>
int launch_p3_inst(const GemmArgs& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel_p3<TI, TO, ACT, HAS_RES, REMAP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE2 * STAGE2);
        if (e != hipSuccess) return cfsar_fail("cfsar_gemm: set LDS size: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles_m = (a.M + BM2 - 1) / BM2;
    hipLaunchKernelGGL((gemm_kernel_p3<TI, TO, ACT, HAS_RES, REMAP>), dim3(tiles_m * a.tiles_n), dim3(NTHREADS2),
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

// ============================================================================================================
// v3 ("p4"): 256(M) x 256(N) tile, 512 threads (8 waves as 2(M) x 4(N), each wave 128 x 64 = 4x2 MFMA 32x32 tiles,
// 128 accumulator VGPRs).  K in 64-byte slices (32 bf16) through a FOUR-stage LDS ring (4 x 32 KiB) filled by asm
// LDS-DMA running three slices ahead (counted vmcnt).  Versus p3 this moves 1/3 fewer bytes global->LDS and 1/4 fewer
// bytes LDS->VGPR per MFMA.  64-byte LDS rows: chunk ^= (row>>2)&3 keeps ds_read_b128 conflict-free.
// ============================================================================================================
constexpr int BM4 = 256;
constexpr int BN4 = 256;
constexpr int ROWB4 = 64;
constexpr int STAGE4 = (BM4 + BN4) * ROWB4;   // 32 KiB
constexpr int NSTAGE4 = 4;
constexpr int LDS4 = 8 * EPI_WAVE_BYTES > NSTAGE4 * STAGE4 ? 8 * EPI_WAVE_BYTES : NSTAGE4 * STAGE4;   // 139264 B

__device__ __forceinline__ int swz4(int row) { return (row >> 2) & 3; }

template <typename TI, typename TO, int ACT, bool HAS_RES, bool REMAP>
__global__ __launch_bounds__(NTHREADS2) void gemm_kernel_p4(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = ROWB4 / (int)sizeof(TI);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int tm = lin / p.tiles_n, tn = lin - tm * p.tiles_n;
    const int m0 = tm * BM4, n0 = tn * BN4;

    // staging: each operand tile = 16 instructions of 16 rows x 64 B; wave w issues instructions {w, w+8} of X and of W
    const char* srcX[2];
    const char* srcW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 8 + wave) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ swz4(row);
        int gm = ((p.dbg & 8) ? 0 : m0) + row;
        gm = gm < p.M ? gm : p.M - 1;
        int gn = ((p.dbg & 8) ? 0 : n0) + row;
        gn = gn < p.N ? gn : p.N - 1;
        srcX[i] = p.A + ((size_t)gm * p.lda) * sizeof(TI) + chunk * 16;
        srcW[i] = p.W + ((size_t)gn * p.ldw) * sizeof(TI) + chunk * 16;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    auto issue = [&](int stage, int kt) {
        const unsigned base = __builtin_amdgcn_readfirstlane(ldsw + (unsigned)stage * (unsigned)STAGE4);
        const size_t koff = (size_t)kt * ROWB4;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_asm(srcX[i] + koff, base + i * 8192);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_asm(srcW[i] + koff, base + BM4 * ROWB4 + i * 8192);
    };

    const int wm = wave >> 2, wn = wave & 3;
    const int lr = lane & 31, hi = lane >> 5;
    int offX[4], offW[2], sxX[4], sxW[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rx = wm * 128 + i * 32 + lr;
        offX[i] = rx * ROWB4;
        sxX[i] = swz4(rx);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rw = wn * 64 + i * 32 + lr;
        offW[i] = BM4 * ROWB4 + rw * ROWB4;
        sxW[i] = swz4(rw);
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = p.K / BK;
    // Phase stagger: the workgroups of the first round start together and would stay in lockstep for the whole launch
    // (same work per tile), so every CU would hit its store-only epilogue -- an HBM write burst with idle MFMA pipes --
    // at the same moment.  Half of the first-round workgroups therefore start `stagger` x ~4 us late; later rounds
    // inherit the offset because a workgroup is dispatched when its predecessor on that CU retires.
    if (p.stagger > 0 && b < 256 && ((b >> 3) & 1))
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    // Fragments of slice kt+1 are read from LDS (into the other register set) while the MFMAs of slice kt execute, so
    // no MFMA ever waits for LDS latency behind a barrier.  Invariant at the barrier of step kt: slices <= kt+1 have
    // landed for every wave (each wave leaves only its share of slice kt+2 in flight: vmcnt(4)); slice kt+3 is then
    // issued into the stage of slice kt-1, whose fragments were consumed one step ago.
    auto load_frags = [&](int kt, uint4 (&xf)[2][4], uint4 (&wf)[2][2]) {
        const char* base = smem + (kt & 3) * STAGE4;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xf[s2][i] = *reinterpret_cast<const uint4*>(base + offX[i] + (((2 * s2 + hi) ^ sxX[i]) << 4));
#pragma unroll
            for (int i = 0; i < 2; ++i)
                wf[s2][i] = *reinterpret_cast<const uint4*>(base + offW[i] + (((2 * s2 + hi) ^ sxW[i]) << 4));
        }
    };
    auto mma = [&](uint4 (&xf)[2][4], uint4 (&wf)[2][2]) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if constexpr (sizeof(TI) == 2) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s2][ni]),
                                                                              __builtin_bit_cast(bf16x8, xf[s2][mi]),
                                                                              acc[mi][ni], 0, 0, 0);
                    } else {
                        const f32x4 a = __builtin_bit_cast(f32x4, wf[s2][ni]);
                        const f32x4 bb = __builtin_bit_cast(f32x4, xf[s2][mi]);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bb[j], acc[mi][ni], 0, 0, 0);
                    }
                }
    };
    auto step = [&](int kt, uint4 (&xc)[2][4], uint4 (&wc)[2][2], uint4 (&xn)[2][4], uint4 (&wn_)[2][2]) {
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(p.dbg & 2)) __syncthreads();
        if (kt + 3 < nk && !(p.dbg & 1)) issue((kt + 3) & 3, kt + 3);
        if (kt + 1 < nk && !(p.dbg & 16)) load_frags(kt + 1, xn, wn_);
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.dbg & 32)) mma(xc, wc);
    };
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    if (nk > 2) issue(2, 2);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint4 xfA[2][4], wfA[2][2], xfB[2][4], wfB[2][2];
    load_frags(0, xfA, wfA);
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, xfA, wfA, xfB, wfB);
        if (kt + 1 < nk) step(kt + 1, xfB, wfB, xfA, wfA);
    }
    __syncthreads();
    if (p.dbg & 4) {
        if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.out)[0] = acc[1][1][3] + acc[3][1][2] + acc[2][0][1];
        return;
    }
    char* wbuf = smem + wave * EPI_WAVE_BYTES;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int mb = m0 + wm * 128 + half * 64, nb = n0 + wn * 64;
        const bool full = mb + 64 <= p.M && nb + 64 <= p.N;
        if constexpr (kBf16PackedEpilogue && sizeof(TO) == 2 && !HAS_RES && !REMAP) {
            if (full) epilogue_lds_bf16<ACT, true>(&acc[2 * half], p, mb, nb, lane, wbuf);
            else epilogue_lds_bf16<ACT, false>(&acc[2 * half], p, mb, nb, lane, wbuf);
        } else {
            if (full) epilogue_lds<TO, ACT, HAS_RES, REMAP, true>(&acc[2 * half], p, mb, nb, lane, wbuf);
            else epilogue_lds<TO, ACT, HAS_RES, REMAP, false>(&acc[2 * half], p, mb, nb, lane, wbuf);
        }
    }
}

// PERSIST: launched with 256 workgroups (one per CU); each walks tiles b, b+256, ... -- no workgroup retire/dispatch and
// store-drain latency between tiles (measured ~4-6 us of a ~34 us K=768 tile in the one-tile-per-workgroup form).
template <typename TI, typename TO, int ACT, bool HAS_RES, bool REMAP, bool PERSIST>
__global__ __launch_bounds__(NTHREADS2) void gemm_kernel_p6(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = ROWB4 / (int)sizeof(TI);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    const int wm = wave >> 2, wn = wave & 3;
    const int lr = lane & 31, hi = lane >> 5;
    int offX[4], offW[2], sxX[4], sxW[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rx = wm * 128 + i * 32 + lr;
        offX[i] = rx * ROWB4;
        sxX[i] = swz4(rx);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rw = wn * 64 + i * 32 + lr;
        offW[i] = BM4 * ROWB4 + rw * ROWB4;
        sxW[i] = swz4(rw);
    }
    const int nk = p.K / BK;
    const bool grpB = wave >= 4;

    auto stamp = [&](int lin, int slot) {
        if (p.trace && tid == 0) p.trace[(size_t)lin * 8 + slot] = __builtin_readcyclecounter();
    };
    // staging: each operand tile = 16 instructions of 16 rows x 64 B; wave w issues instructions {w, w+8} of X and of W
    const char* srcX[2];
    const char* srcW[2];
    auto set_src = [&](int lin) {
        const int tm = lin / p.tiles_n, tn = lin - tm * p.tiles_n;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (i * 8 + wave) * 16 + (lane >> 2);
            const int chunk = (lane & 3) ^ swz4(row);
            int gm = tm * BM4 + row;
            gm = gm < p.M ? gm : p.M - 1;
            int gn = tn * BN4 + row;
            gn = gn < p.N ? gn : p.N - 1;
            srcX[i] = p.A + ((size_t)gm * p.lda) * sizeof(TI) + chunk * 16;
            srcW[i] = p.W + ((size_t)gn * p.ldw) * sizeof(TI) + chunk * 16;
        }
    };
    auto issue = [&](int stage, int kt) {
        const unsigned base = __builtin_amdgcn_readfirstlane(ldsw + (unsigned)stage * (unsigned)STAGE4);
        const size_t koff = (size_t)kt * ROWB4;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_asm(srcX[i] + koff, base + i * 8192);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_asm(srcW[i] + koff, base + BM4 * ROWB4 + i * 8192);
    };

    // pre_issued: slices 0..2 of this tile were already issued (persistent form: during / after the previous epilogue)
    auto do_tile = [&](int lin, int next_lin, bool pre_issued) {
    const int tm = lin / p.tiles_n, tn = lin - tm * p.tiles_n;
    const int m0 = tm * BM4, n0 = tn * BN4;
    stamp(lin, 0);
    if (!pre_issued) set_src(lin);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // ---- ping-pong schedule.  Waves 0-3 (group A, top 128 rows) and 4-7 (group B, bottom 128 rows) share the four SIMDs
    // pairwise (wave w and w+4).  Every wave alternates a MEMORY phase (12 ds_read_b128 of slice j's fragments + its 4
    // LDS-DMA issues for slice j+3 + counted waits) and a COMPUTE phase (16 MFMAs), separated by s_barrier; group B runs
    // one barrier behind group A, so on each SIMD one wave feeds the matrix pipe while its partner does LDS/DMA work.
    // Barrier G(2j) .. G(2j+1): A memory j | B compute j-1;   G(2j+1) .. G(2j+2): A compute j | B memory j.
    // Invariants (4 stages, DMA distance 3): a wave ends memory phase j only when its own DMA share of slice j+1 has
    // landed (vmcnt(8) leaves slices j+2, j+3 in flight), so after G(2j) every share of slice j is in LDS for A's reads
    // and after G(2j+1) for B's; slice j+3 overwrites the stage of slice j-1, last read by B before G(2j).
    uint4 xf[2][4], wf[2][2];
    auto mem_phase = [&](int j) {
        const char* base = smem + (j & 3) * STAGE4;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xf[s2][i] = *reinterpret_cast<const uint4*>(base + offX[i] + (((2 * s2 + hi) ^ sxX[i]) << 4));
#pragma unroll
            for (int i = 0; i < 2; ++i)
                wf[s2][i] = *reinterpret_cast<const uint4*>(base + offW[i] + (((2 * s2 + hi) ^ sxW[i]) << 4));
        }
        if (j + 3 < nk) {
            issue((j + 3) & 3, j + 3);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (j + 2 < nk) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute_phase = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s2][ni]),
                                                                          __builtin_bit_cast(bf16x8, xf[s2][mi]),
                                                                          acc[mi][ni], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    if (!pre_issued) {
        issue(0, 0);
        if (nk > 1) issue(1, 1);
        if (nk > 2) issue(2, 2);
        if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // slices 0,1 were issued before the previous epilogue's stores, slice 2 after them: leaving only slice 2 in
        // flight also drains those stores (they have had the whole epilogue to complete)
        if (nk > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                      // G0: slice 0 is in LDS for everyone
    __builtin_amdgcn_sched_barrier(0);
    stamp(lin, 1);
    if (grpB) {
        __builtin_amdgcn_s_barrier();                  // group B starts one barrier late
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int j = 0; j < nk; ++j) {
        mem_phase(j);
        compute_phase();
    }
    if (!grpB) {
        __builtin_amdgcn_s_barrier();                  // group A absorbs the stagger
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    stamp(lin, 2);
    if (p.dbg & 4) {
        if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.out)[0] = acc[1][1][3] + acc[3][1][2] + acc[2][0][1];
        return;
    }
    if constexpr (PERSIST) {
        // Prefetch the next tile's slices 0 and 1 into stages 0/1 (bytes [0, 64 KiB)) BEFORE the epilogue, whose LDS
        // staging is confined to [64 KiB, 64 KiB + 8 x 8704 B): the ~13 % pipeline-fill latency of a K = 768 tile hides
        // behind the epilogue.  The epilogue runs in four 32-row passes per wave.
        const bool has_next = next_lin < p.ntiles;
        if (has_next) {
            set_src(next_lin);
            issue(0, 0);
            if (nk > 1) issue(1, 1);
        }
        char* wbuf = smem + 65536 + wave * (32 * EPI_RS);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int mb = m0 + wm * 128 + mi * 32, nb = n0 + wn * 64;
            if (mb + 32 <= p.M && nb + 64 <= p.N) epilogue_lds<TO, ACT, HAS_RES, REMAP, true, 1>(&acc[mi], p, mb, nb, lane, wbuf);
            else epilogue_lds<TO, ACT, HAS_RES, REMAP, false, 1>(&acc[mi], p, mb, nb, lane, wbuf);
            if (mi == 1) stamp(lin, 3);
        }
        stamp(lin, 4);
        __syncthreads();                                   // staging reads done: stage 2 may be overwritten
        if (has_next && nk > 2) issue(2, 2);
        stamp(lin, 5);
    } else {
    char* wbuf = smem + wave * EPI_WAVE_BYTES;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int mb = m0 + wm * 128 + half * 64, nb = n0 + wn * 64;
        const bool full = mb + 64 <= p.M && nb + 64 <= p.N;
        if constexpr (kBf16PackedEpilogue && sizeof(TO) == 2 && !HAS_RES && !REMAP) {
            if (full) epilogue_lds_bf16<ACT, true>(&acc[2 * half], p, mb, nb, lane, wbuf);
            else epilogue_lds_bf16<ACT, false>(&acc[2 * half], p, mb, nb, lane, wbuf);
        } else {
            if (full) epilogue_lds<TO, ACT, HAS_RES, REMAP, true>(&acc[2 * half], p, mb, nb, lane, wbuf);
            else epilogue_lds<TO, ACT, HAS_RES, REMAP, false>(&acc[2 * half], p, mb, nb, lane, wbuf);
        }
        stamp(lin, 3 + half);
    }
    if (p.trace) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(lin, 5);
    }
    }
    };   // do_tile

    if constexpr (PERSIST) {
        // Block b lives on XCD b % 8; inside each chunk of G tiles the workgroups of one XCD take CONSECUTIVE raster
        // positions, so the tiles an L2 sees at one time share X and W panels.
        const int G = gridDim.x, b = blockIdx.x, per_xcd = G >> 3;
        bool pre = false;
        for (int lin = (b & 7) * per_xcd + (b >> 3); lin < p.ntiles; lin += G) {
            do_tile(lin, lin + G, pre);
            pre = true;
        }
    } else {
        const int nwg = gridDim.x, b = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
        do_tile((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3), p.ntiles, false);
    }
}

template <typename TI, typename TO, int ACT, bool HAS_RES, bool REMAP>
int launch_p4_inst(const GemmArgs& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel_p4<TI, TO, ACT, HAS_RES, REMAP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
        if (e != hipSuccess) return cfsar_fail("cfsar_gemm: set LDS size: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles_m = (a.M + BM4 - 1) / BM4;
    hipLaunchKernelGGL((gemm_kernel_p4<TI, TO, ACT, HAS_RES, REMAP>), dim3(tiles_m * a.tiles_n), dim3(NTHREADS2), LDS4, s, a);
    return cfsar_check_launch("cfsar_gemm(p4)");
}

// bf16 only, the shapes of the ViT blocks
template <typename TO>
int launch_p4(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_n = (a.N + BN4 - 1) / BN4;
    const bool r = a.res != nullptr;
    if (a.row_group > 0 || a.res_mod > 0 || a.act == CFSAR_ACT_GELU_ERF) return -2;
    if (a.act == CFSAR_ACT_QUICKGELU)
        return r ? -2 : launch_p4_inst<__bf16, TO, CFSAR_ACT_QUICKGELU, false, false>(a, s);
    return r ? launch_p4_inst<__bf16, TO, CFSAR_ACT_NONE, true, false>(a, s)
             : launch_p4_inst<__bf16, TO, CFSAR_ACT_NONE, false, false>(a, s);
}

template <typename TI, typename TO, int ACT, bool HAS_RES, bool REMAP, bool PERSIST>
int launch_p6_inst(const GemmArgs& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel_p6<TI, TO, ACT, HAS_RES, REMAP, PERSIST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
        if (e != hipSuccess) return cfsar_fail("cfsar_gemm: set LDS size: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int grid = PERSIST ? 256 : a.ntiles;
    hipLaunchKernelGGL((gemm_kernel_p6<TI, TO, ACT, HAS_RES, REMAP, PERSIST>), dim3(grid), dim3(NTHREADS2), LDS4, s, a);
    return cfsar_check_launch("cfsar_gemm(p6)");
}

template <typename TO, bool PERSIST>
int launch_p6(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_n = (a.N + BN4 - 1) / BN4;
    a.ntiles = ((a.M + BM4 - 1) / BM4) * a.tiles_n;
    const bool r = a.res != nullptr;
    if (a.row_group > 0 || a.res_mod > 0 || a.act == CFSAR_ACT_GELU_ERF) return -2;
    if (a.act == CFSAR_ACT_QUICKGELU)
        return r ? -2 : launch_p6_inst<__bf16, TO, CFSAR_ACT_QUICKGELU, false, false, PERSIST>(a, s);
    return r ? launch_p6_inst<__bf16, TO, CFSAR_ACT_NONE, true, false, PERSIST>(a, s)
             : launch_p6_inst<__bf16, TO, CFSAR_ACT_NONE, false, false, PERSIST>(a, s);
}

// ============================================================================================================
// v4 ("p5"): 256(M) x 128(N) tile, 256 threads (4 waves as 2x2, each wave 128 x 64 = 4x2 MFMA tiles), 64-byte K
// slices, THREE-stage ring of 24 KiB (72 KiB per workgroup) -> TWO independent workgroups per CU.  The two workgroups
// drift out of phase, so one workgroup's pipeline fill / epilogue (no MFMA) overlaps the other's main loop -- the
// serialisation that caps the one-workgroup-per-CU kernels (p3/p4) on the short-K (768) GEMMs of the ViT.
// ============================================================================================================
constexpr int BM5 = 256;
constexpr int BN5 = 128;
constexpr int STAGE5 = (BM5 + BN5) * ROWB4;   // 24 KiB
constexpr int NSTAGE5 = 3;
constexpr int LDS5 = NSTAGE5 * STAGE5;        // 73728 B  (>= 4 * EPI_WAVE_BYTES = 69632)

template <typename TI, typename TO, int ACT, bool HAS_RES, bool REMAP>
__global__ __launch_bounds__(256, 2) void gemm_kernel_p5(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BK = ROWB4 / (int)sizeof(TI);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int tm = lin / p.tiles_n, tn = lin - tm * p.tiles_n;
    const int m0 = tm * BM5, n0 = tn * BN5;

    // staging: X tile = 16 instructions of 16 rows x 64 B, W tile = 8; wave w issues X {w, w+4, w+8, w+12}, W {w, w+4}
    const char* srcX[4];
    const char* srcW[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (i * 4 + wave) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ swz4(row);
        int gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        srcX[i] = p.A + ((size_t)gm * p.lda) * sizeof(TI) + chunk * 16;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 4 + wave) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ swz4(row);
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        srcW[i] = p.W + ((size_t)gn * p.ldw) * sizeof(TI) + chunk * 16;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    auto issue = [&](int stage, int kt) {
        const unsigned base = __builtin_amdgcn_readfirstlane(ldsw + (unsigned)stage * (unsigned)STAGE5);
        const size_t koff = (size_t)kt * ROWB4;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_asm(srcX[i] + koff, base + i * 4096);
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_asm(srcW[i] + koff, base + BM5 * ROWB4 + i * 4096);
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, hi = lane >> 5;
    int offX[4], offW[2], sxX[4], sxW[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rx = wm * 128 + i * 32 + lr;
        offX[i] = rx * ROWB4;
        sxX[i] = swz4(rx);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rw = wn * 64 + i * 32 + lr;
        offW[i] = BM5 * ROWB4 + rw * ROWB4;
        sxW[i] = swz4(rw);
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int nk = p.K / BK;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int st2 = st + 2;
        st2 = st2 >= NSTAGE5 ? st2 - NSTAGE5 : st2;
        if (kt + 2 < nk) issue(st2, kt + 2);
        const char* base = smem + st * STAGE5;
        uint4 xf[2][4], wf[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[0][i] = *reinterpret_cast<const uint4*>(base + offX[i] + ((hi ^ sxX[i]) << 4));
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[0][i] = *reinterpret_cast<const uint4*>(base + offW[i] + ((hi ^ sxW[i]) << 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[1][i] = *reinterpret_cast<const uint4*>(base + offX[i] + (((2 + hi) ^ sxX[i]) << 4));
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[1][i] = *reinterpret_cast<const uint4*>(base + offW[i] + (((2 + hi) ^ sxW[i]) << 4));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if constexpr (sizeof(TI) == 2) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s2][ni]),
                                                                              __builtin_bit_cast(bf16x8, xf[s2][mi]),
                                                                              acc[mi][ni], 0, 0, 0);
                    } else {
                        const f32x4 a = __builtin_bit_cast(f32x4, wf[s2][ni]);
                        const f32x4 bb = __builtin_bit_cast(f32x4, xf[s2][mi]);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bb[j], acc[mi][ni], 0, 0, 0);
                    }
                }
        st = st + 1 >= NSTAGE5 ? 0 : st + 1;
    }
    __syncthreads();
    char* wbuf = smem + wave * EPI_WAVE_BYTES;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int mb = m0 + wm * 128 + half * 64, nb = n0 + wn * 64;
        const bool full = mb + 64 <= p.M && nb + 64 <= p.N;
        if constexpr (kBf16PackedEpilogue && sizeof(TO) == 2 && !HAS_RES && !REMAP) {
            if (full) epilogue_lds_bf16<ACT, true>(&acc[2 * half], p, mb, nb, lane, wbuf);
            else epilogue_lds_bf16<ACT, false>(&acc[2 * half], p, mb, nb, lane, wbuf);
        } else {
            if (full) epilogue_lds<TO, ACT, HAS_RES, REMAP, true>(&acc[2 * half], p, mb, nb, lane, wbuf);
            else epilogue_lds<TO, ACT, HAS_RES, REMAP, false>(&acc[2 * half], p, mb, nb, lane, wbuf);
        }
    }
}

template <typename TI, typename TO, int ACT, bool HAS_RES, bool REMAP>
int launch_p5_inst(const GemmArgs& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel_p5<TI, TO, ACT, HAS_RES, REMAP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS5);
        if (e != hipSuccess) return cfsar_fail("cfsar_gemm: set LDS size: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int tiles_m = (a.M + BM5 - 1) / BM5;
    hipLaunchKernelGGL((gemm_kernel_p5<TI, TO, ACT, HAS_RES, REMAP>), dim3(tiles_m * a.tiles_n), dim3(256), LDS5, s, a);
    return cfsar_check_launch("cfsar_gemm(p5)");
}

template <typename TO>
int launch_p5(const GemmArgs& a0, hipStream_t s) {
    GemmArgs a = a0;
    a.tiles_n = (a.N + BN5 - 1) / BN5;
    const bool r = a.res != nullptr;
    if (a.row_group > 0 || a.res_mod > 0 || a.act == CFSAR_ACT_GELU_ERF) return -2;
    if (a.act == CFSAR_ACT_QUICKGELU)
        return r ? -2 : launch_p5_inst<__bf16, TO, CFSAR_ACT_QUICKGELU, false, false>(a, s);
    return r ? launch_p5_inst<__bf16, TO, CFSAR_ACT_NONE, true, false>(a, s)
             : launch_p5_inst<__bf16, TO, CFSAR_ACT_NONE, false, false>(a, s);
}

}  // namespace

static unsigned long long* g_trace = nullptr;
// dev tool (not in the public header): device buffer of 8 x u64 per 256x256 tile for the p6 phase timestamps; NULL = off
extern "C" void cfsar_debug_set_gemm_trace(void* buf) { g_trace = static_cast<unsigned long long*>(buf); }

extern "C" int cfsar_gemm_ex(const void* A, const void* W, void* out, const float* bias, const void* residual, int M,
                             int N, int K, int lda, int ldw, int ldo, int ldr, int in_dtype, int out_dtype, int act,
                             int row_group, int row_gap, int row_off, int res_mod, int res_off, int res_dtype, int relu,
                             cfsar_stream_t stream) {
    CFSAR_REQUIRE(A && W && out, "cfsar_gemm: null operand");
    CFSAR_REQUIRE(M > 0 && N > 0 && K > 0, "cfsar_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    CFSAR_REQUIRE(in_dtype == CFSAR_F32 || in_dtype == CFSAR_BF16, "cfsar_gemm: bad in_dtype %d", in_dtype);
    CFSAR_REQUIRE(out_dtype == CFSAR_F32 || out_dtype == CFSAR_BF16, "cfsar_gemm: bad out_dtype %d", out_dtype);
    const int esz = in_dtype == CFSAR_BF16 ? 2 : 4;
    const int bk = ROWB / esz;
    CFSAR_REQUIRE(K % bk == 0, "cfsar_gemm: K=%d must be a multiple of %d for this dtype", K, bk);
    CFSAR_REQUIRE(N % 4 == 0, "cfsar_gemm: N=%d must be a multiple of 4", N);
    CFSAR_REQUIRE((lda * esz) % 16 == 0 && (ldw * esz) % 16 == 0, "cfsar_gemm: lda/ldw rows must be 16-byte aligned");
    CFSAR_REQUIRE(lda >= K && ldw >= K && ldo >= N, "cfsar_gemm: leading dimension too small");
    CFSAR_REQUIRE((ldo * (out_dtype == CFSAR_BF16 ? 2 : 4)) % 8 == 0, "cfsar_gemm: ldo alignment");
    CFSAR_REQUIRE(!residual || (ldr >= N && ldr % 4 == 0), "cfsar_gemm: bad ldr");
    CFSAR_REQUIRE(act >= 0 && act <= 2, "cfsar_gemm: bad act %d", act);
    CFSAR_REQUIRE(res_dtype == CFSAR_F32 || res_dtype == CFSAR_BF16, "cfsar_gemm: bad res_dtype %d", res_dtype);
    GemmArgs a;
    a.A = static_cast<const char*>(A);
    a.W = static_cast<const char*>(W);
    a.out = out;
    a.bias = bias;
    a.res = residual;
    a.res_bf16 = res_dtype == CFSAR_BF16;
    a.relu = relu;
    a.M = M; a.N = N; a.K = K;
    a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.ldr = ldr;
    a.act = act;
    a.row_group = row_group; a.row_gap = row_gap; a.row_off = row_off;
    a.res_mod = res_mod; a.res_off = res_off;
    a.tiles_n = (N + BN - 1) / BN;
    static const int dbg = [] { const char* e = getenv("CFSAR_GEMM_DEBUG"); return e ? atoi(e) : 0; }();
    a.dbg = dbg;
    static const int stag = [] { const char* e = getenv("CFSAR_GEMM_STAGGER"); return e ? atoi(e) : -1; }();
    a.stagger = stag;
    a.trace = g_trace;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // variant: 0 = auto, 1 = v1 (128x128, 2-stage, compiler-managed LDS-DMA), 2 = p3 (256x128, 3-stage, asm LDS-DMA)
    static const int forced = [] { const char* e = getenv("CFSAR_GEMM_VARIANT"); return e ? atoi(e) : 0; }();
    // p4 needs enough 256x256 tiles to fill the 256 CUs for >= 2 rounds
    const long tiles4 = (long)((M + BM4 - 1) / BM4) * ((N + BN4 - 1) / BN4);
    // variants: 1 = v1 (128x128), 2 = p3 (256x128, 3-stage), 3 = p4 (256x256, 4-stage), 4 = p5 (256x128, 2 WG/CU),
    // 6 = p6 (256x256 ping-pong).  auto: p6 when >= 2 rounds of 256x256 tiles exist, else p3 / v1.
    if (in_dtype == CFSAR_BF16 && forced == 7) {               // 7 = p6 persistent
        const int rc = out_dtype == CFSAR_BF16 ? launch_p6<__bf16, true>(a, s) : launch_p6<float, true>(a, s);
        if (rc != -2) return rc;
    }
    if (in_dtype == CFSAR_BF16 && (forced == 6 || (forced == 0 && tiles4 >= 512))) {
        const int rc = out_dtype == CFSAR_BF16 ? launch_p6<__bf16, false>(a, s) : launch_p6<float, false>(a, s);
        if (rc != -2) return rc;
    }
    const bool use_p4 = in_dtype == CFSAR_BF16 && (forced == 3 || forced == 6 || forced == 7 || (forced == 0 && tiles4 >= 512));
    if (use_p4) {
        const int rc = out_dtype == CFSAR_BF16 ? launch_p4<__bf16>(a, s) : launch_p4<float>(a, s);
        if (rc != -2) return rc;
    }
    if (in_dtype == CFSAR_BF16 && forced == 4) {
        const int rc = out_dtype == CFSAR_BF16 ? launch_p5<__bf16>(a, s) : launch_p5<float>(a, s);
        if (rc != -2) return rc;
    }
    const bool use_p3 = forced == 2 || forced == 3 || forced == 4 || forced == 6 || (forced == 0 && in_dtype == CFSAR_BF16 && M >= 1024);
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

extern "C" int cfsar_gemm(const void* A, const void* W, void* out, const float* bias, const float* residual, int M,
                          int N, int K, int lda, int ldw, int ldo, int ldr, int in_dtype, int out_dtype, int act,
                          int row_group, int row_gap, int row_off, int res_mod, int res_off, cfsar_stream_t stream) {
    return cfsar_gemm_ex(A, W, out, bias, residual, M, N, K, lda, ldw, ldo, ldr, in_dtype, out_dtype, act, row_group, row_gap,
                         row_off, res_mod, res_off, CFSAR_F32, 0, stream);
}
