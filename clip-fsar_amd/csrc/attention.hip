// A5: scaled-dot-product attention inside nn.MultiheadAttention of the CLIP ViT (few_shot.py:623,633-635):
// per (frame, head): softmax(q k^T / sqrt(64)) v, no mask, no dropout.  Sequence = 197 (B/16) or 257 (L/14) tokens,
// head_dim = 64, so K and V of one (frame, head) fit in LDS and the softmax is single-pass.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"

#ifdef CFSAR_DEV
extern int g_cfsar_walk_enable, g_cfsar_walk_phase;      // experiment: gemm_vit.hip (dbg bit 25 of cfsar_debug_set_vit_dbg)
#endif

namespace {

typedef unsigned att_u32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x32_{bf16,f16} by element type (the fp16 numerics mode runs the same kernel on fp16 q / k / v and fp16 P)
template <typename T>
__device__ __forceinline__ f32x4 att_mfma(typename Vec2B<T>::v8 a, typename Vec2B<T>::v8 b, f32x4 c) {
    if constexpr (std::is_same<T, _Float16>::value) return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// O^T tile -> global, 16 bytes per lane.  A lane (q = lane&15, g = lane>>4) holds d = 16dt + 4g + r of its query row; lanes g
// and g^1 hold the two halves of each 8-wide d chunk.  After one exchange with lane^16 (4 dwords each way) the even-g lane owns
// the chunks of dt 0,1 and the odd-g lane those of dt 2,3: two dwordx4 stores per lane instead of four dwordx2 (the
// attention epilogue is store-ISSUE bound: 59 of 273 us at 640 frames were the 8-byte stores).
// LOW (round 6, the two-word output of the fp16_strict mode): the tile of SECOND words, (T)(x - (float)(T)x) of every x = o * inv.
template <typename T, bool LOW = false>
__device__ __forceinline__ void pack_o_tile(const f32x4 (&o)[4], float inv, int g, uint4 (&val)[2]) {
    unsigned pk[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        typename Vec2B<T>::v4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (LOW) {
                const float x = o[dt][r] * inv;
                v[r] = (T)(x - (float)(T)x);
            } else {
                v[r] = (T)(o[dt][r] * inv);
            }
        }
        const uint2 u = __builtin_bit_cast(uint2, v);
        pk[dt][0] = u.x;
        pk[dt][1] = u.y;
    }
    const bool odd = g & 1;
    // send the two dt chunks the partner will store, keep the two this lane stores.  Written as explicit two-way selects: indexing pk[]
    // with a lane-dependent dt made the compiler build 7-deep compare / select chains over all eight words (71 VALU instructions per
    // tile, round 3 ISA audit).
    unsigned recv[2][2], mine[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            recv[i][w] = __shfl_xor(odd ? pk[i][w] : pk[2 + i][w], 16, 64);
            mine[i][w] = odd ? pk[2 + i][w] : pk[i][w];
        }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // chunk of 8 d values starting at 16 dt + 8 (g >> 1), dt = odd ? 2 + i : i: low half from the even-g lane, high half from the odd-g lane
        val[i] = odd ? make_uint4(recv[i][0], recv[i][1], mine[i][0], mine[i][1]) : make_uint4(mine[i][0], mine[i][1], recv[i][0], recv[i][1]);
    }
}
template <typename T>
__device__ __forceinline__ void store_o_packed(const uint4 (&val)[2], bool valid, T* orow, int g) {
    if (valid) {
        const bool odd = g & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(orow + (odd ? 2 + i : i) * 16 + (g >> 1) * 8) = val[i];
    }
}
template <typename T, bool LOW = false>
__device__ __forceinline__ void store_o_tile(const f32x4 (&o)[4], float inv, bool valid, T* orow, int g) {
    uint4 val[2];
    pack_o_tile<T, LOW>(o, inv, g, val);
    store_o_packed(val, valid, orow, g);
}

// ------------------------------------------------------------------------------------------------------------
// bf16 MFMA kernel (round 2).  One workgroup of NW waves per (head, frame).
//   Staging: the K rows and the V rows of the (frame, head) go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: 8 rows x 128 B per
//         wave-instruction, no VGPR round trip, no transposition pass); both tiles keep the global row-major [key][64] bf16 layout
//         with 128-byte rows and XOR-swizzled 16-byte chunks (the swizzle is applied to the per-lane SOURCE address):
//           K: chunk ^= (row >> 1) & 7         -> the 16 rows x one chunk of a ds_read_b128 K fragment hit 16 different slots
//           V: chunk ^= ((row >> 1) & 3) << 1  -> the 8 rows x 32 B of a ds_read_b64_tr_b16 half hit 8 different 32-byte slots
//         (round 1 loaded K / V into registers, transposed V with 4-byte LDS writes -- 8.2-11.5 K of a workgroup's 28.8 K cycles --
//         and read V^T with 2-way conflicted ds_read2_b64: 37 % of the LDS cycles were conflicts.)
//   Per wave: 16 query rows at a time.  S^T = K . Q^T with v_mfma_f32_16x16x32_bf16 (K rows feed MFMA "A", Q^T feeds "B"), so
//         lane (q = lane & 15, g = lane >> 4) holds, for every 16-key tile j, the scores of keys 16 j + 4 g + {0..3} for ITS query:
//         the softmax reduction over keys is in-lane + two shuffles (xor 16, 32), and the exponentiated scores are already the
//         "B" fragment of the PV MFMA (O^T = V^T . P^T): k-slot (g, i < 4) <-> key 32 kb + 4 g + i, (g, i >= 4) <-> key
//         32 kb + 16 + 4 g + (i - 4).  The matching V^T "A" fragment (d = 16 dt + (lane & 15); the same 8 keys) is TWO
//         ds_read_b64_tr_b16: the hardware transposes a [4 keys][16 d] block of the row-major V tile per 16-lane group
//         (semantics probed in tools/ubench/tr_read_probe.hip).  No P round trip through LDS, no cross-lane data movement.
//   The exponentials of key block kb + 1 are issued before the PV MFMAs of block kb, so the transcendental / VALU work of the
//         softmax runs in the shadow of the matrix pipe instead of in a phase of its own.
// ------------------------------------------------------------------------------------------------------------
// NKB = number of 32-key blocks (7 -> up to 224 keys, 9 -> up to 288 keys); NTV = number of 16-key tiles that hold at least one
// real key when known at compile time (13 for 197 tokens, 17 for 257), 0 = generic (all 2 NKB tiles, masked by ntok).
// NW = waves per workgroup; WPS = waves per SIMD the register budget must allow (workgroups per CU x NW / 4).
__device__ __forceinline__ void att_glds16(const char* src, unsigned lds_addr) {
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
typedef short att_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float att_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
template <typename F, int... Is>
__device__ __forceinline__ void att_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void att_static_for(F&& f) {
    att_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// Per-lane LDS read offsets of the tile computation.  K fragment of tile j: row 16 j + q16, chunk (4 ks + g) ^ ((row >> 1) & 7)
// -- the swizzle term does not depend on j (16 j is a multiple of 16).  V^T fragment of (dt, kb): rows 32 kb (+16) + 4 g +
// (q16 >> 2), bytes ((dt ^ s) << 5) + ((q16 & 3) << 3),  s = (2 g + (q16 >> 3)) & 3.
struct AttLane {
    int q16, g, krow, koff[2], vrow, voff[4];
    __device__ __forceinline__ explicit AttLane(int lane) {
        q16 = lane & 15;
        g = lane >> 4;
        krow = q16 * 128;
        const int kswz = (q16 >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) koff[ks] = ((ks * 4 + g) ^ kswz) << 4;
        vrow = (4 * g + (q16 >> 2)) * 128 + ((q16 & 3) << 3);
        const int vs = (2 * g + (q16 >> 3)) & 3;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) voff[dt] = (dt ^ vs) << 5;
    }
};

// One 16-query tile against the K / V tiles in LDS: o[dt][r] = (unnormalised) O^T[16 dt + 4 g + r][q16], inv = 1 / row sum.
// The LDS fragment reads run TWO steps ahead of the MFMAs that consume them (S^T: two 16-key tiles; PV: one 32-key block = 8
// transpose reads): left to itself hipcc emits read -> wait -> MFMA pairs, and a wave then spends an LDS round trip per MFMA pair
// (13 + 28 round trips per tile; measured 12.7 K cycles per (frame, head) against 2.8 K of matrix-pipe work).
template <typename T, int NKB, int NTV, int KPF_ = 0>
__device__ __forceinline__ void attn_tile(const char* sK, const char* sV, const AttLane& L, const typename Vec2B<T>::v8 (&qf)[2], int ntok,
                                          float scale_log2e, f32x4 (&o)[4], float& inv) {
    constexpr int NT = NKB * 2;
    constexpr int nt_valid = NTV > 0 ? NTV : NT;
    const char* kbase = sK + L.krow;
    const char* vbase = sV + L.vrow;
    const int g = L.g;
    typedef typename Vec2B<T>::v8 T8;
    f32x4 s[NT];
    T8 kf[nt_valid][2];
    auto kload = [&](auto J) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kf[j][ks] = *reinterpret_cast<const T8*>(kbase + j * 2048 + L.koff[ks]);
    };
    // read-ahead in tiles: two, except that the 288-key instance (72 score registers) affords only one inside a 128-register budget
    constexpr int KPF = KPF_ > 0 ? KPF_ : (NKB >= 9 ? 1 : 2);
    att_static_for<KPF>([&](auto J) {
        if constexpr (decltype(J)::value < nt_valid) kload(J);
    });
    att_static_for<NT>([&](auto J) {
        constexpr int j = decltype(J)::value;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (j < nt_valid) {
            if constexpr (j + KPF < nt_valid) kload(std::integral_constant<int, j + KPF>{});
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) acc = att_mfma<T>(kf[j][ks], qf[ks], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        s[j] = acc;
    });
    // mask the padded keys of the last (partial) tile, row max over keys (in-lane, then across the 4 lane groups)
    float mx = -1e30f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (NTV == 0 || j == nt_valid - 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (j * 16 + g * 4 + r >= ntok) s[j][r] = -1e30f;
        }
        // v_max3_f32 from inline asm: fmaxf on MFMA results makes hipcc put a canonicalising v_max in front of every operand
        // (82 + 13 instead of 26 instructions per tile)
        if (j < nt_valid) mx = att_max3(att_max3(mx, s[j][0], s[j][1]), s[j][2], s[j][3]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mxs = mx * scale_log2e;
    // P fragment of key block kb: exp2 of tiles 2 kb, 2 kb + 1 (raw v_exp_f32), rounded to the operand type.  The row sums come from the
    // matrix pipe: one more MFMA per key block with an all-ones "A" fragment adds the block's (rounded) probabilities of every query --
    // 7 MFMAs instead of 52 VALU adds and two cross-lane shuffles per tile (the tile routine is VALU-issue bound), and the normaliser is
    // the sum of exactly the values the PV product uses.
    auto pblock = [&](int kb) __attribute__((always_inline)) {
        T8 pf;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = 2 * kb + t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pv = 0.f;
                if (j < nt_valid) {
                    const float z = fmaf(s[j][r], scale_log2e, -mxs);
                    pv = __builtin_amdgcn_exp2f(z);
                    asm volatile("" : "+v"(pv) : "v"(z));       // operand pin kept from the round-2 fault hunt (csrc/gemm_vit.hip, above quick_gelu4): no instruction, no measured cost
                }
                pf[4 * t + r] = (T)pv;
            }
        }
        return pf;
    };
    // V^T fragments of block kb: second half (keys 32 kb + 16 ...): when that tile is not staged (odd nt_valid) its probabilities
    // are 0 and the fragment re-reads the first half (finite values)
    constexpr int NBLK = (nt_valid + 1) / 2;
    uint4 vf[2][4];
    auto vload = [&](int kb, uint4 (&dst)[4]) __attribute__((always_inline)) {
        const int hoff = (2 * kb + 1 < nt_valid) ? 2048 : 0;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const char* vp = vbase + kb * 4096 + L.voff[dt];
            const att_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)(vp));
            const att_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) att_s16x4*)(vp + hoff));
            const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
            dst[dt] = make_uint4(l2.x, l2.y, h2.x, h2.y);
        }
    };
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    vload(0, vf[0]);
    T8 pcur = pblock(0);
    const T one = (T)1.0f;
    const T8 ones = T8{one, one, one, one, one, one, one, one};
    f32x4 osum = {0.f, 0.f, 0.f, 0.f};
    att_static_for<NBLK>([&](auto KB) {
        constexpr int kb = decltype(KB)::value;
        T8 pnext = pcur;
        if constexpr (kb + 1 < NBLK) {
            vload(kb + 1, vf[(kb + 1) & 1]);                      // next block's transpose reads ...
            pnext = pblock(kb + 1);                                // ... and its exponentials, under the MFMAs below
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            o[dt] = att_mfma<T>(__builtin_bit_cast(T8, vf[kb & 1][dt]), pcur, o[dt]);
        osum = att_mfma<T>(ones, pcur, osum);
        __builtin_amdgcn_sched_barrier(0);
        pcur = pnext;
    });
    inv = __builtin_amdgcn_rcpf(osum[0]);                       // every row of the ones product is the query's sum over all keys
}

// PAIR (round 6, fp16_strict): the output keeps TWO fp16 words per element, out [F ntok, 2 D] = [o_hi | o_lo] (o_lo = the rounding remainder of o_hi):
// out_proj reads it as a K = 2 D operand against [W | W] (cfsar_gemm_residual_wide, wsplit = 2) and its result no longer carries the 11-bit rounding of
// the attention output -- a fifth of the strict mode's remaining error on ViT-L/14 (profiles/r06_strict_budget.md) for one more store pass here.
template <typename T, int NKB, int NTV, int NW, int WPS, int KPFK = 0, bool MEANS = false, bool PAIR = false>
__global__ __launch_bounds__(NW * 64, WPS) void vit_attn_bf16_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                                     int ntok, int D, float scale_log2e, int dbg,
                                                                     __bf16* __restrict__ omean = nullptr) {
    constexpr int NT = NKB * 2;                                    // 16-key tiles
    constexpr int nt_valid = NTV > 0 ? NTV : NT;                   // tiles that are computed
    constexpr int KROWS = nt_valid * 16;                           // rows of K and of V kept in LDS
    constexpr int NPIECE = KROWS / 8;                              // 1 KiB DMA pieces per operand
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = smem + KROWS * 128;

    const int h = blockIdx.x;
#ifdef CFSAR_DEV
    const int f = (dbg & 64) ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;      // experiment: frames from the end (see gemm_vit.hip, dbg bit 24)
#else
    const int f = blockIdx.y;
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t ld = (size_t)3 * D;
    typedef typename Vec2B<T>::v8 T8;
    const T* base = qkv + (size_t)f * ntok * ld + h * 64;
    const char* baseb = reinterpret_cast<const char*>(base);

    // ---- stage K and V: piece p = rows 8p .. 8p+7; lane (r = lane >> 3, c = lane & 7) fills LDS chunk c of its row with the
    // global chunk c ^ swizzle(row).  Rows >= ntok (padding of the last tile) re-read row ntok-1: finite values whose
    // probabilities are exactly 0.
    {
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
        const int r = lane >> 3, c = lane & 7;
#pragma unroll
        for (int it = 0; it < (NPIECE + NW - 1) / NW; ++it) {
            const int pc = wave + it * NW;                          // wave-uniform
            if (pc < NPIECE) {
                int row = pc * 8 + r;
                const int srow = row < ntok ? row : ntok - 1;
                const char* rowp = baseb + (size_t)srow * ld * 2;
                const int ck = c ^ ((row >> 1) & 7);
                att_glds16(rowp + (size_t)D * 2 + ck * 16, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)pc * 1024u));
            }
        }
#pragma unroll
        for (int it = 0; it < (NPIECE + NW - 1) / NW; ++it) {       // V after K: the S phase starts as soon as K has landed
            const int pc = wave + it * NW;
            if (pc < NPIECE) {
                int row = pc * 8 + r;
                const int srow = row < ntok ? row : ntok - 1;
                const char* rowp = baseb + (size_t)srow * ld * 2;
                const int cv = c ^ (((row >> 1) & 3) << 1);
                att_glds16(rowp + (size_t)D * 4 + cv * 16, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(KROWS * 128) + (unsigned)pc * 1024u));
            }
        }
    }
    const int q16 = lane & 15, g = lane >> 4;
    const int nqt = (ntok + 15) >> 4;
    // Q fragment ("B" operand) of this wave's first tile, in flight while the DMA lands: Q[qrow][32 ks + 8 g .. +8].
    // (Tried: Q by LDS-DMA into a wave-private 2 KiB buffer that then stages the O tile for whole-line stores -- 404 vs 410 us at
    // 1 280 frames, inside the noise, for 14-18 KiB of LDS; removed.)
    T8 qf[2];
    {
        int qrow = wave * 16 + q16;
        qrow = qrow < ntok ? qrow : ntok - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const T8*>(base + (size_t)qrow * ld + ks * 32 + g * 8);
    }
    // One wait for everything.  (Tried: K first / V under the S phase with counted vmcnt -- LDS-DMA and VGPR loads share the
    // counter but do not retire in one order, so a counted wait across the two kinds is not a guarantee (wrong results), and
    // with vmcnt(0) everywhere the split bought nothing: 406 vs 411 us at 1 280 frames, other workgroups already fill the wait.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const AttLane L(lane);
    // omean != NULL (round 4, fp16 numerics mode): the per-frame TOKEN MEAN of this head's output, omean[f][64 h + d] -- out_proj's per-frame
    // low-word correction needs it, and this workgroup is the one place that sees all of a (frame, head)'s output rows: no pass over `out`.
    // Fixed summation order per (frame, head): the result does not depend on the batch the frame is served in.
    f32x4 osum[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int qt = wave; qt < nqt; qt += NW) {
        int qrow = qt * 16 + q16;
        const bool qvalid = qrow < ntok;
        if (!qvalid) qrow = ntok - 1;
        T8 qn[2];                                               // the next tile's Q rows: in flight during this tile's computation
        {
            int qr2 = (qt + NW) * 16 + q16;
            qr2 = qr2 < ntok ? qr2 : ntok - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qn[ks] = *reinterpret_cast<const T8*>(base + (size_t)qr2 * ld + ks * 32 + g * 8);
        }
        f32x4 o[4];
        float inv;
#ifdef CFSAR_DEV
        if (dbg & 1) {                                               // ablation: no tile computation (memory pipeline only)
            inv = 1.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{(float)qf[0][0], 0.f, 0.f, (float)qf[1][1]};
        } else
#endif
        attn_tile<T, NKB, NTV, ((KPFK > 0) ? KPFK : ((NKB >= 9 && WPS > 2) ? 1 : 2))>(sK, sV, L, qf, ntok, scale_log2e, o, inv);
        // O^T[d][q]: lane owns query q16, d = 16 dt + 4 g + r
        T* const orow = out + ((size_t)f * ntok + qrow) * (PAIR ? 2 * D : D) + h * 64;
        store_o_tile<T>(o, inv, qvalid, orow, g);
        if constexpr (PAIR) store_o_tile<T, true>(o, inv, qvalid, orow + D, g);
        if constexpr (MEANS) {
            const float w = qvalid ? inv : 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) osum[dt][r] = __builtin_fmaf(o[dt][r], w, osum[dt][r]);
        }
        qf[0] = qn[0];
        qf[1] = qn[1];
    }
    if constexpr (MEANS) {
        // sum over the 16 queries of a lane row (q16 = lane & 15: one DPP row), then over the waves through the (now idle) K tile
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = osum[dt][r];
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));   // row_ror:8
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));   // row_ror:4
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false));   // row_ror:2
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));   // row_ror:1
                osum[dt][r] = v;
            }
        __syncthreads();                                             // every wave is done with sK / sV
        float* red = reinterpret_cast<float*>(smem);
        if (q16 == 0) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * 64 + 16 * dt + 4 * g + r] = osum[dt][r];
        }
        __syncthreads();
        if (tid < 64) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += red[w * 64 + tid];
            omean[(size_t)f * D + h * 64 + tid] = (__bf16)(sum / (float)ntok);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Persistent ring form of the kernel above for ntok <= 16 NTV (ViT-B/16: 197 tokens, NTV = 13): ONE workgroup per CU walks
// (frame, head) items; K / V of item i + 2 are in flight while item i is computed.
//
// Why: the one-item-per-workgroup kernel ran at 3.8-3.9 TB/s whatever the workgroup shape (2 x 7, 2 x 8, 3 x 5 waves per CU: 394-408 us
// at 1 280 frames, 1.55 GB of traffic): with one buffer per workgroup about ONE item (53 KB) per CU is in flight at a time, and
// 256 x 53 KB / ~3.5 us of loaded HBM latency is that rate (Little's law).  The memory floor is ~260 us.
//   * LDS = a ring of THREE K+V buffers (3 x 53 248 B of the CU's 163 840): two items (106 KB) in flight per CU all the time;
//   * wave NTV is a LOADER: it issues every LDS-DMA piece (52 per item) and nothing else, so its vmcnt queue holds one kind of
//     operation, retires in order, and `s_waitcnt vmcnt(52)` means "item i has landed, item i + 1 may still fly".  (Q loads, O
//     stores and DMA share the counter but retire out of order with each other: consumers cannot count across kinds.)
//   * waves 0 .. NTV-1 are CONSUMERS, one 16-query tile each (same S / softmax / PV code as above);
//   * ONE barrier per item: the loader arrives when item i has landed, the consumers when they are done with item i - 1; past it
//     the loader refills the buffer of item i - 1 with item i + 2.
// ------------------------------------------------------------------------------------------------------------
template <int NKB, int NTV>
__global__ __launch_bounds__((NTV + 1) * 64, 4) void vit_attn_ring_kernel(const __bf16* __restrict__ qkv, __bf16* __restrict__ out,
                                                                          int ntok, int D, int heads, int nitems, float scale_log2e) {
    constexpr int nt_valid = NTV;
    constexpr int KROWS = nt_valid * 16;
    constexpr int NPIECE = KROWS / 8;
    constexpr int BUF = 2 * KROWS * 128;
    static_assert(2 * NPIECE <= 63, "the loader's counted wait must fit vmcnt");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t ld = (size_t)3 * D;
    const int first = blockIdx.x, stride = gridDim.x;
    const int nmine = first < nitems ? (nitems - first + stride - 1) / stride : 0;       // same for every wave of the workgroup

    if (wave == NTV) {
        // ------------------------------------------------------------------ loader
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
        const int r = lane >> 3, c = lane & 7;
        auto issue = [&](int k) __attribute__((always_inline)) {       // item first + k * stride -> ring slot k % 3
            const int item = first + k * stride;
            const int f = item / heads, h = item - f * heads;
            const char* baseb = reinterpret_cast<const char*>(qkv + (size_t)f * ntok * ld + h * 64);
            const unsigned slot = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(k % 3) * (unsigned)BUF);
#pragma unroll 2
            for (int pc = 0; pc < NPIECE; ++pc) {                    // (rolled: the loader's addresses must not cost the consumers registers)
                const int row = pc * 8 + r;
                const int srow = row < ntok ? row : ntok - 1;
                att_glds16(baseb + (size_t)srow * ld * 2 + (size_t)D * 2 + ((c ^ ((row >> 1) & 7)) << 4),
                           __builtin_amdgcn_readfirstlane(slot + (unsigned)pc * 1024u));
            }
#pragma unroll 2
            for (int pc = 0; pc < NPIECE; ++pc) {
                const int row = pc * 8 + r;
                const int srow = row < ntok ? row : ntok - 1;
                att_glds16(baseb + (size_t)srow * ld * 2 + (size_t)D * 4 + ((c ^ (((row >> 1) & 3) << 1)) << 4),
                           __builtin_amdgcn_readfirstlane(slot + (unsigned)(KROWS * 128) + (unsigned)pc * 1024u));
            }
        };
        if (nmine > 0) issue(0);
        if (nmine > 1) issue(1);
        for (int k = 0; k < nmine; ++k) {
            if (k + 1 < nmine) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPIECE) : "memory");     // item k landed; k + 1 may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                                                       // item k ready / item k - 1 consumed
            if (k + 2 < nmine) issue(k + 2);
        }
        return;
    }
    // ---------------------------------------------------------------------- consumers: wave w owns query tile w
    const int q16 = lane & 15, g = lane >> 4;
    const int qt = wave;
    int qrow = qt * 16 + q16;
    const bool qvalid = qrow < ntok;
    if (!qvalid) qrow = ntok - 1;
    const bool has_tile = qt * 16 < ntok;                            // wave-uniform
    const AttLane L(lane);
    auto qptr = [&](int k) __attribute__((always_inline)) {
        const int item = first + k * stride;
        const int f = item / heads, h = item - f * heads;
        return qkv + ((size_t)f * ntok + qrow) * ld + h * 64;
    };
    bf16x8 qf[2];
    if (nmine > 0 && has_tile) {
        const __bf16* qp = qptr(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32 + g * 8);
    }
    for (int k = 0; k < nmine; ++k) {
        __syncthreads();                                             // item k has landed in ring slot k % 3
        if (!has_tile) continue;
        const char* sK = smem + (k % 3) * BUF;
        bf16x8 qn[2] = {qf[0], qf[1]};
        if (k + 1 < nmine) {                                         // next item's Q rows: in flight during this item's computation
            const __bf16* qp = qptr(k + 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qn[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 32 + g * 8);
        }
        f32x4 o[4];
        float inv;
        attn_tile<__bf16, NKB, NTV>(sK, sK + KROWS * 128, L, qf, ntok, scale_log2e, o, inv);
        {
            const int item = first + k * stride;
            const int f = item / heads, h = item - f * heads;
            store_o_tile<__bf16>(o, inv, qvalid, out + ((size_t)f * ntok + qrow) * D + h * 64, g);
        }
        qf[0] = qn[0];
        qf[1] = qn[1];
    }
}

// ------------------------------------------------------------------------------------------------------------
// Pipelined persistent form (round 3) for ntok <= 16 NTV <= 256 (ViT-B/16: 197 tokens, NTV = 13 waves): ONE workgroup per CU walks
// (frame, head) items with TWO K + V buffers; while item k is computed, item k + 1 is in flight -- issued by ALL waves (4 LDS-DMA
// pieces each), not by a loader wave.
//
// Why: the one-item-per-workgroup kernel keeps nothing in flight while its two workgroups per CU compute (378 us at 1 280 frames
// against a 294 us memory-only pipeline, 1.55 GB); the round-2 ring kept two items in flight but fed them through ONE loader wave, whose
// 52 serial DMA issues per item (~25 GB/s per CU, MI355X_MICROARCH.md "ldsdma-fill") capped it at 409 us.  Here every wave issues its
// 2 + 2 pieces right behind the barrier that frees the buffer, one query tile per wave (same S / softmax / PV routine), and the
// iteration time is the item's memory time as long as the tile computation (~3.5 K cycles) is shorter than it (~9 K cycles).
//   per item and wave:   compute tile of item k (buffer k & 1)
//                        s_waitcnt vmcnt(0)   -- this wave's pieces and Q rows of item k + 1 have landed (issued one item ago)
//                        store O(k)           -- after the wait: the stores drain under the next item's computation
//                        barrier              -- buffer k & 1 is free, item k + 1 is complete in the other one
//                        issue item k + 2 -> buffer k & 1, load its Q rows
// ------------------------------------------------------------------------------------------------------------
template <typename T, int NKB, int NTV>
__global__ __launch_bounds__(NTV * 64, 4) void vit_attn_pipe_kernel(const T* __restrict__ qkv, T* __restrict__ out, int ntok, int D,
                                                                    int heads, int nitems, float scale_log2e) {
    typedef typename Vec2B<T>::v8 T8;
    constexpr int KROWS = NTV * 16;
    constexpr int NPIECE = KROWS / 8;                               // 1 KiB pieces per operand: 2 per wave
    constexpr int BUF = 2 * KROWS * 128;
    static_assert(NPIECE == 2 * NTV, "two K and two V pieces per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t ld = (size_t)3 * D;
    const int first = blockIdx.x, stride = gridDim.x;
    const int nmine = first < nitems ? (nitems - first + stride - 1) / stride : 0;       // same for every wave of the workgroup
    if (nmine == 0) return;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int r = lane >> 3, c = lane & 7;
    const int q16 = lane & 15, g = lane >> 4;
    int qrow = wave * 16 + q16;
    const bool qvalid = qrow < ntok;
    if (!qvalid) qrow = ntok - 1;
    const AttLane L(lane);
    auto base_of = [&](int k) __attribute__((always_inline)) {
        const int item = first + k * stride;
        const int f = item / heads, h = item - f * heads;
        return qkv + (size_t)f * ntok * ld + h * 64;
    };
    auto issue = [&](int k) __attribute__((always_inline)) {           // this wave's pieces {wave, wave + NTV} of K and of V -> buffer k & 1
        const char* baseb = reinterpret_cast<const char*>(base_of(k));
        const unsigned slot = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(k & 1) * (unsigned)BUF);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = wave + i * NTV;
            const int row = pc * 8 + r;
            const int srow = row < ntok ? row : ntok - 1;
            att_glds16(baseb + (size_t)srow * ld * 2 + (size_t)D * 2 + ((c ^ ((row >> 1) & 7)) << 4),
                       __builtin_amdgcn_readfirstlane(slot + (unsigned)pc * 1024u));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = wave + i * NTV;
            const int row = pc * 8 + r;
            const int srow = row < ntok ? row : ntok - 1;
            att_glds16(baseb + (size_t)srow * ld * 2 + (size_t)D * 4 + ((c ^ (((row >> 1) & 3) << 1)) << 4),
                       __builtin_amdgcn_readfirstlane(slot + (unsigned)(KROWS * 128) + (unsigned)pc * 1024u));
        }
    };
    auto loadq = [&](int k, T8 (&q)[2]) __attribute__((always_inline)) {
        const T* qp = base_of(k) + (size_t)qrow * ld;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) q[ks] = *reinterpret_cast<const T8*>(qp + ks * 32 + g * 8);
    };
    T8 qf[2], qn[2];
    issue(0);
    loadq(0, qf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    qn[0] = qf[0];
    qn[1] = qf[1];
    if (nmine > 1) {
        issue(1);
        loadq(1, qn);
    }
    for (int k = 0; k < nmine; ++k) {
        const char* sK = smem + (k & 1) * BUF;
        f32x4 o[4];
        float inv;
        attn_tile<T, NKB, NTV, 3>(sK, sK + KROWS * 128, L, qf, ntok, scale_log2e, o, inv);
        uint4 val[2];
        pack_o_tile<T>(o, inv, g, val);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // item k + 1: my pieces and Q rows are here (and O(k - 1) has left)
        {
            const int item = first + k * stride;
            const int f = item / heads, h = item - f * heads;
            store_o_packed<T>(val, qvalid, out + ((size_t)f * ntok + qrow) * D + h * 64, g);
        }
        __syncthreads();
        qf[0] = qn[0];
        qf[1] = qn[1];
        if (k + 2 < nmine) {
            issue(k + 2);
            loadq(k + 2, qn);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// fp32 VALU kernel (validation mode).  One workgroup per (head, frame); K and V rows in LDS (fp32); one thread per
// query row with an online softmax (running max / sum, fp32) -- all lanes read the same K/V row (LDS broadcast).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vit_attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           int ntok, int D, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sK = reinterpret_cast<float*>(smem);
    float* sV = sK + (size_t)ntok * 64;
    const int h = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const size_t ld = (size_t)3 * D;
    const float* base = qkv + (size_t)f * ntok * ld + h * 64;
    for (int idx = tid; idx < ntok * 16; idx += 256) {
        const int r = idx >> 4, c = idx & 15;
        *reinterpret_cast<float4*>(sK + r * 64 + c * 4) = *reinterpret_cast<const float4*>(base + (size_t)r * ld + D + c * 4);
        *reinterpret_cast<float4*>(sV + r * 64 + c * 4) =
            *reinterpret_cast<const float4*>(base + (size_t)r * ld + 2 * D + c * 4);
    }
    __syncthreads();
    for (int qrow = tid; qrow < ntok; qrow += 256) {
        float q[64], o[64];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)qrow * ld + c * 4);
            q[4 * c] = v.x * scale; q[4 * c + 1] = v.y * scale; q[4 * c + 2] = v.z * scale; q[4 * c + 3] = v.w * scale;
        }
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] = 0.f;
        float m = -1e30f, l = 0.f;
        for (int key = 0; key < ntok; ++key) {
            const float* kr = sK + key * 64;
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) sc = fmaf(q[d], kr[d], sc);
            const float mn = fmaxf(m, sc);
            const float a = expf(m - mn), pv = expf(sc - mn);
            l = l * a + pv;
            const float* vr = sV + key * 64;
#pragma unroll
            for (int d = 0; d < 64; ++d) o[d] = fmaf(o[d], a, pv * vr[d]);
            m = mn;
        }
        const float inv = 1.0f / l;
        float* orow = out + ((size_t)f * ntok + qrow) * D + h * 64;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            *reinterpret_cast<float4*>(orow + 4 * c) =
                make_float4(o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Class-token attention (round 3).  VisionTransformer.forward keeps ONLY the class token of the last block's output
// (few_shot.py:683: `x = self.ln_post(x[:, 0, :])`), so in that block the attention output -- and everything behind it -- is needed
// for query row 0 of every frame alone: softmax(q_0 K^T / 8) V per (frame, head), 1 query x ntok keys.  One wave per item; an
// instruction covers 64 / LPR key rows of 64 elements (LPR lanes x 16 bytes = one whole row), scores and probabilities live in a
// per-wave LDS row, all arithmetic in fp32.  Reads K and V once (2/3 of the qkv matrix): memory bound.
// ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void vit_attn_cls_kernel(const T* __restrict__ qp, long long ldq, const T* __restrict__ kp,
                                                           const T* __restrict__ vp, int ldkv, T* __restrict__ out, int ntok, int D,
                                                           int heads, int nitems, float scale) {
    constexpr int EPL = 16 / (int)sizeof(T);                      // elements per lane (16 bytes)
    constexpr int LPR = 64 / EPL;                                  // lanes per 64-element row: 8 (2-byte types) or 16 (fp32)
    constexpr int RPI = 64 / LPR;                                  // key rows per wave-instruction
    typedef T tvec __attribute__((ext_vector_type(EPL)));
    __shared__ float sp[4][320];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wave;
    if (item >= nitems) return;
    const int f = item / heads, h = item - f * heads;
    const size_t ld = (size_t)ldkv;
    const T* kbase = kp + (size_t)f * ntok * ld + h * 64;
    const T* vbase = vp + (size_t)f * ntok * ld + h * 64;
    const int r = lane / LPR, c = lane % LPR;
    float q[EPL];
    {
        const tvec qv = *reinterpret_cast<const tvec*>(qp + (size_t)f * ldq + h * 64 + c * EPL);
#pragma unroll
        for (int e = 0; e < EPL; ++e) q[e] = (float)qv[e] * scale;
    }
    float* p = sp[wave];
    const int nit = (ntok + RPI - 1) / RPI;
    for (int it = 0; it < nit; ++it) {
        const int k = it * RPI + r;
        const int kk = k < ntok ? k : ntok - 1;
        const tvec kv = *reinterpret_cast<const tvec*>(kbase + (size_t)kk * ld + c * EPL);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) d = fmaf(q[e], (float)kv[e], d);
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if (c == 0 && k < ntok) p[k] = d;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                            // lgkmcnt(0): this wave's LDS writes (one wave per row: no barrier needed)
    float mx = -1e30f;
    for (int k = lane; k < ntok; k += 64) mx = fmaxf(mx, p[k]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < ntok; k += 64) {
        const float e = __expf(p[k] - mx);
        p[k] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    float acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    for (int it = 0; it < nit; ++it) {
        const int k = it * RPI + r;
        const bool ok = k < ntok;
        const int kk = ok ? k : ntok - 1;
        const tvec vv = *reinterpret_cast<const tvec*>(vbase + (size_t)kk * ld + c * EPL);
        const float pk = ok ? p[kk] : 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = fmaf(pk, (float)vv[e], acc[e]);
    }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
    }
    if (r == 0) {
        tvec ov;
#pragma unroll
        for (int e = 0; e < EPL; ++e) ov[e] = (T)(acc[e] * inv);
        *reinterpret_cast<tvec*>(out + (size_t)f * D + h * 64 + c * EPL) = ov;
    }
}

#ifdef CFSAR_DEV
int g_attn_dbg = 0;
#endif
static inline int attn_dbg() {
#ifdef CFSAR_DEV
    if (g_cfsar_walk_enable) return g_attn_dbg | ((g_cfsar_walk_phase++ & 1) << 6);
    return g_attn_dbg;
#else
    return 0;
#endif
}

template <int NKB, int NTV, int NW, int WPS, int KPFK = 0, typename T = __bf16>
int launch_bf16(const void* qkv, void* out, int F, int ntok, int D, int heads, hipStream_t s, void* omean = nullptr, bool pair = false) {
    constexpr int KROWS = (NTV > 0 ? NTV : NKB * 2) * 16;
    constexpr int LDS = 2 * KROWS * 128;
    const float scale_log2e = 0.125f * 1.4426950408889634f;
    if constexpr (std::is_same<T, _Float16>::value) {
        if (omean != nullptr && pair) {           // fp16_strict: two-word output [o_hi | o_lo] + the per-frame token means
            if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&vit_attn_bf16_kernel<T, NKB, NTV, NW, WPS, KPFK, true, true>), LDS, "cfsar_vit_attention")) return rc;
            hipLaunchKernelGGL((vit_attn_bf16_kernel<T, NKB, NTV, NW, WPS, KPFK, true, true>), dim3(heads, F), dim3(NW * 64), LDS, s,
                               static_cast<const T*>(qkv), static_cast<T*>(out), ntok, D, scale_log2e, attn_dbg(), static_cast<__bf16*>(omean));
            return cfsar_check_launch("cfsar_vit_attention_pair");
        }
        if (omean != nullptr) {                   // the fp16 numerics mode's form that also emits the output's per-frame token means
            if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&vit_attn_bf16_kernel<T, NKB, NTV, NW, WPS, KPFK, true>), LDS, "cfsar_vit_attention")) return rc;
            hipLaunchKernelGGL((vit_attn_bf16_kernel<T, NKB, NTV, NW, WPS, KPFK, true>), dim3(heads, F), dim3(NW * 64), LDS, s,
                               static_cast<const T*>(qkv), static_cast<T*>(out), ntok, D, scale_log2e, attn_dbg(), static_cast<__bf16*>(omean));
            return cfsar_check_launch("cfsar_vit_attention_means");
        }
    }
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&vit_attn_bf16_kernel<T, NKB, NTV, NW, WPS, KPFK>), LDS, "cfsar_vit_attention")) return rc;
    hipLaunchKernelGGL((vit_attn_bf16_kernel<T, NKB, NTV, NW, WPS, KPFK>), dim3(heads, F), dim3(NW * 64), LDS, s,
                       static_cast<const T*>(qkv), static_cast<T*>(out), ntok, D, scale_log2e, attn_dbg(), static_cast<__bf16*>(nullptr));
    return cfsar_check_launch("cfsar_vit_attention(16-bit)");
}

template <typename T, int NKB, int NTV>
int launch_pipe(const void* qkv, void* out, int F, int ntok, int D, int heads, hipStream_t s) {
    constexpr int LDS = 2 * 2 * NTV * 16 * 128;
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&vit_attn_pipe_kernel<T, NKB, NTV>), LDS, "cfsar_vit_attention")) return rc;
    const int nitems = F * heads;
    const int cus = cfsar_num_cus();
    const int grid = nitems < cus ? nitems : cus;
    hipLaunchKernelGGL((vit_attn_pipe_kernel<T, NKB, NTV>), dim3(grid), dim3(NTV * 64), LDS, s, static_cast<const T*>(qkv),
                       static_cast<T*>(out), ntok, D, heads, nitems, 0.125f * 1.4426950408889634f);
    return cfsar_check_launch("cfsar_vit_attention(pipelined)");
}

template <int NKB, int NTV>
int launch_ring(const void* qkv, void* out, int F, int ntok, int D, int heads, hipStream_t s) {
    constexpr int LDS = 3 * 2 * NTV * 16 * 128;
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&vit_attn_ring_kernel<NKB, NTV>), LDS, "cfsar_vit_attention")) return rc;
    const int nitems = F * heads;
    const int cus = cfsar_num_cus();
    const int grid = nitems < cus ? nitems : cus;
    hipLaunchKernelGGL((vit_attn_ring_kernel<NKB, NTV>), dim3(grid), dim3((NTV + 1) * 64), LDS, s, static_cast<const __bf16*>(qkv),
                       static_cast<__bf16*>(out), ntok, D, heads, nitems, 0.125f * 1.4426950408889634f);
    return cfsar_check_launch("cfsar_vit_attention(bf16 ring)");
}

}  // namespace

#ifdef CFSAR_DEV
#include "../../include/clipfsar_hip_dev.h"
static int g_attn_variant = 0;
extern "C" void cfsar_debug_set_attn_variant(int v) { g_attn_variant = v & 255; g_attn_dbg = v >> 8; }
#endif

static int vit_attention_impl(const void* qkv, void* out, int dtype, int F, int ntok, int D, int heads, void* omean, cfsar_stream_t stream, bool pair = false);

extern "C" int cfsar_vit_attention(const void* qkv, void* out, int dtype, int F, int ntok, int D, int heads,
                                   cfsar_stream_t stream) {
    return vit_attention_impl(qkv, out, dtype, F, ntok, D, heads, nullptr, stream);
}

// cfsar_vit_attention (fp16) that also writes the per-frame token means of its output, omean [F, D] bf16 (see the header).
extern "C" int cfsar_vit_attention_means(const void* qkv, void* out, void* omean, int F, int ntok, int D, int heads,
                                         cfsar_stream_t stream) {
    CFSAR_REQUIRE(omean != nullptr, "cfsar_vit_attention_means: null pointer");
    return vit_attention_impl(qkv, out, CFSAR_F16, F, ntok, D, heads, omean, stream);
}

// cfsar_vit_attention_means with the output in TWO fp16 words: out_pair [F ntok, 2 D] = [o_hi | o_lo] (see the header).
extern "C" int cfsar_vit_attention_pair(const void* qkv, void* out_pair, void* omean, int F, int ntok, int D, int heads, cfsar_stream_t stream) {
    CFSAR_REQUIRE(omean != nullptr, "cfsar_vit_attention_pair: null pointer");
    return vit_attention_impl(qkv, out_pair, CFSAR_F16, F, ntok, D, heads, omean, stream, true);
}

static int vit_attention_impl(const void* qkv, void* out, int dtype, int F, int ntok, int D, int heads, void* omean, cfsar_stream_t stream, bool pair) {
    CFSAR_REQUIRE(qkv && out, "cfsar_vit_attention: null pointer");
    CFSAR_REQUIRE(F > 0 && ntok > 0 && heads > 0 && D == heads * 64, "cfsar_vit_attention: need D == heads*64 (D=%d heads=%d)",
                  D, heads);
    CFSAR_REQUIRE(F <= 65535, "cfsar_vit_attention: too many frames per call (%d)", F);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == CFSAR_BF16) {
        CFSAR_REQUIRE(ntok <= 288, "cfsar_vit_attention: ntok=%d > 288", ntok);
        // (A compact-LDS three-workgroups-per-CU form and a pipelined multi-head form were measured slower / equal in round 1 and
        // removed; numbers and per-phase cycle stamps: profiles/r01_attention_ablation.md.)
        // 197 tokens: 13 query tiles over 8 waves, 52 KiB of LDS, 128 registers -> 2 workgroups per CU (measured at 1 280 frames with
        // the pipelined tile routine: 2 x 8 waves 371-378 us, 2 x 7 383, 3 x 5 403, persistent ring 409; memory pipeline alone 294);
        // 257 tokens (ViT-L/14 @224): 17 tiles, 68 KiB
#ifdef CFSAR_DEV
        if (ntok == 197 && g_attn_variant == 7) return launch_ring<7, 13>(qkv, out, F, ntok, D, heads, s);           // persistent ring
        if (ntok == 197 && g_attn_variant == 30) return launch_pipe<__bf16, 7, 13>(qkv, out, F, ntok, D, heads, s);   // pipelined persistent (r03)
        if (ntok == 197 && g_attn_variant == 31) return launch_bf16<7, 13, 8, 4, 3>(qkv, out, F, ntok, D, heads, s);  // r02 product kernel
        if (ntok == 197 && g_attn_variant == 40) return launch_bf16<7, 13, 4, 3, 3>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 197 && g_attn_variant == 41) return launch_bf16<7, 13, 4, 3, 5>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 197 && g_attn_variant == 42) return launch_bf16<7, 13, 3, 3>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 197 && g_attn_variant == 43) return launch_bf16<7, 13, 4, 4>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 197 && g_attn_variant == 44) return launch_bf16<7, 13, 5, 3>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 257 && g_attn_variant == 45) return launch_bf16<9, 17, 4, 3>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 257 && g_attn_variant == 46) return launch_bf16<9, 17, 4, 3, 3>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 257 && g_attn_variant == 47) return launch_bf16<9, 17, 5, 3>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 257 && g_attn_variant == 8) return launch_bf16<9, 17, 9, 4>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 257 && g_attn_variant == 20) return launch_bf16<9, 17, 8, 4>(qkv, out, F, ntok, D, heads, s);   // r02 mid-round default
        if (ntok == 257 && g_attn_variant == 9) return launch_bf16<9, 17, 6, 4>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 257 && g_attn_variant == 13) return launch_bf16<9, 17, 6, 3>(qkv, out, F, ntok, D, heads, s);   // 170-register budget
        if (ntok == 257 && g_attn_variant == 14) return launch_bf16<9, 17, 5, 3>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 257 && g_attn_variant == 15) return launch_bf16<9, 17, 4, 2>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 257 && g_attn_variant == 16) return launch_bf16<9, 17, 3, 2>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 257 && g_attn_variant == 17) return launch_bf16<9, 17, 5, 2>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 197 && g_attn_variant == 18) return launch_bf16<7, 13, 3, 2>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 197 && g_attn_variant == 21) return launch_bf16<7, 13, 4, 2, 3>(qkv, out, F, ntok, D, heads, s);   // K fragments 3 tiles ahead
        if (ntok == 197 && g_attn_variant == 22) return launch_bf16<7, 13, 4, 2, 5>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 197 && g_attn_variant == 23) return launch_bf16<7, 13, 8, 4, 3>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 257 && g_attn_variant == 24) return launch_bf16<9, 17, 4, 2, 3>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 257 && g_attn_variant == 25) return launch_bf16<9, 17, 4, 2, 5>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 197 && g_attn_variant == 19) return launch_bf16<7, 13, 5, 2>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 197 && g_attn_variant == 5) return launch_bf16<7, 13, 8, 4>(qkv, out, F, ntok, D, heads, s);   // one item per workgroup
        if (ntok == 197 && g_attn_variant == 6) return launch_bf16<7, 13, 7, 4>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 197 && g_attn_variant == 2) return launch_bf16<7, 13, 5, 4>(qkv, out, F, ntok, D, heads, s);   // 3 workgroups x 5 waves
        if (ntok == 197 && g_attn_variant == 3) return launch_bf16<7, 13, 8, 4>(qkv, out, F, ntok, D, heads, s);   // 2 x 8 waves
        if (ntok == 197 && g_attn_variant == 4) return launch_bf16<7, 13, 13, 4>(qkv, out, F, ntok, D, heads, s);  // one tile per wave
        if (ntok == 197 && g_attn_variant == 10) return launch_bf16<7, 13, 4, 3>(qkv, out, F, ntok, D, heads, s);  // 3 workgroups x 4 waves
        if (ntok == 197 && g_attn_variant == 11) return launch_bf16<7, 13, 4, 2>(qkv, out, F, ntok, D, heads, s);  // (2-3) x 4 waves, 256 regs
        if (ntok == 197 && g_attn_variant == 12) return launch_bf16<7, 13, 6, 4>(qkv, out, F, ntok, D, heads, s);  // 2 x 6 waves
#endif
        // Round 3: after the VALU diet of the tile routine (canonicalising maxes, select chains, row sums on the matrix pipe: 387 -> 365 us)
        // THREE workgroups of 4 waves per CU (156 KiB of LDS, 168-register budget) beat two of 8: 380 -> 355 us at 1 280 frames on
        // one box (3 x 50 KB of K / V in flight per CU instead of 2 x 50 KB), equal at 80 frames.
        if (ntok == 197) return launch_bf16<7, 13, 4, 3>(qkv, out, F, ntok, D, heads, s);
        // 257 tokens (ViT-L/14): 4 waves per workgroup in a 256-register budget -- no spills (15 at 128 registers) and K fragments
        // two tiles ahead: 488 -> 426 us at 640 frames (8 x 128-register waves: 488, 6 x 170: 475, 3 x 256: 460, 5 x 256: 550)
        if (ntok == 257) return launch_bf16<9, 17, 4, 2, 5>(qkv, out, F, ntok, D, heads, s);
        if (ntok > 224) return launch_bf16<9, 0, 4, 2>(qkv, out, F, ntok, D, heads, s);
        return launch_bf16<7, 0, 7, 4>(qkv, out, F, ntok, D, heads, s);
    }
    if (dtype == CFSAR_F16) {                    // the fp16 numerics mode: the same kernel on fp16 q / k / v, P rounded to fp16
        CFSAR_REQUIRE(ntok <= 288, "cfsar_vit_attention: ntok=%d > 288", ntok);
        if (ntok == 197) return launch_bf16<7, 13, 4, 3, 0, _Float16>(qkv, out, F, ntok, D, heads, s, omean, pair);
        if (ntok == 257) return launch_bf16<9, 17, 4, 2, 5, _Float16>(qkv, out, F, ntok, D, heads, s, omean, pair);
        if (ntok > 224) return launch_bf16<9, 0, 4, 2, 0, _Float16>(qkv, out, F, ntok, D, heads, s, omean, pair);
        return launch_bf16<7, 0, 7, 4, 0, _Float16>(qkv, out, F, ntok, D, heads, s, omean, pair);
    }
    if (dtype == CFSAR_F32) {
        const int lds = ntok * 64 * 4 * 2;
        CFSAR_REQUIRE(lds <= 160 * 1024, "cfsar_vit_attention: ntok=%d too large for the fp32 kernel", ntok);
        if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&vit_attn_f32_kernel), lds, "cfsar_vit_attention")) return rc;
        hipLaunchKernelGGL(vit_attn_f32_kernel, dim3(heads, F), dim3(256), lds, s, static_cast<const float*>(qkv),
                           static_cast<float*>(out), ntok, D, 0.125f);
        return cfsar_check_launch("cfsar_vit_attention(f32)");
    }
    return cfsar_fail("cfsar_vit_attention: bad dtype %d", dtype);
}

// The class-token form: out[f, :] = softmax(q_f K_f^T / 8) V_f per head for ONE query per frame (see vit_attn_cls_kernel).  q: row f at q + f * ldq
// (elements); k / v: token t of frame f at k / v + (f * ntok + t) * ldkv; head h at columns 64 h of each.  With the packed qkv matrix of
// cfsar_vit_attention: q = qkv, ldq = ntok * 3 D, k = qkv + D, v = qkv + 2 D, ldkv = 3 D.  out [F, D]; dtype bf16 | fp16 | f32; ntok <= 320.
extern "C" int cfsar_vit_attention_cls(const void* q, long long ldq, const void* k, const void* v, int ldkv, void* out, int dtype, int F,
                                       int ntok, int D, int heads, cfsar_stream_t stream) {
    CFSAR_REQUIRE(q && k && v && out, "cfsar_vit_attention_cls: null pointer");
    CFSAR_REQUIRE(F > 0 && ntok > 0 && ntok <= 320 && heads > 0 && D == heads * 64 && ldq >= D && ldkv >= D,
                  "cfsar_vit_attention_cls: need D == heads*64, ntok <= 320, ldq / ldkv >= D (D=%d heads=%d ntok=%d)", D, heads, ntok);
    const int esz = dtype == CFSAR_F32 ? 4 : 2;
    CFSAR_REQUIRE((ldq * esz) % 16 == 0 && (ldkv * esz) % 16 == 0 && ((size_t)q % 16 | (size_t)k % 16 | (size_t)v % 16 | (size_t)out % 16) == 0,
                  "cfsar_vit_attention_cls: rows must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nitems = F * heads;
    const dim3 grid((unsigned)((nitems + 3) / 4)), block(256);
    if (dtype == CFSAR_BF16)
        hipLaunchKernelGGL(vit_attn_cls_kernel<__bf16>, grid, block, 0, s, static_cast<const __bf16*>(q), ldq, static_cast<const __bf16*>(k),
                           static_cast<const __bf16*>(v), ldkv, static_cast<__bf16*>(out), ntok, D, heads, nitems, 0.125f);
    else if (dtype == CFSAR_F16)
        hipLaunchKernelGGL(vit_attn_cls_kernel<_Float16>, grid, block, 0, s, static_cast<const _Float16*>(q), ldq, static_cast<const _Float16*>(k),
                           static_cast<const _Float16*>(v), ldkv, static_cast<_Float16*>(out), ntok, D, heads, nitems, 0.125f);
    else if (dtype == CFSAR_F32)
        hipLaunchKernelGGL(vit_attn_cls_kernel<float>, grid, block, 0, s, static_cast<const float*>(q), ldq, static_cast<const float*>(k),
                           static_cast<const float*>(v), ldkv, static_cast<float*>(out), ntok, D, heads, nitems, 0.125f);
    else
        return cfsar_fail("cfsar_vit_attention_cls: bad dtype %d", dtype);
    return cfsar_check_launch("cfsar_vit_attention_cls");
}
