// Shared device helpers and epilogues of the persistent ViT-block GEMM kernels (gemm_vit.hip: one 8-wave workgroup per CU, 256 x 256 tiles;
// gemm_vit4.hip: two 4-wave workgroups per CU, 192 x 128 tiles).  Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_vit.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// residual rows of an output tile's first pass fetched before its last K step on the LDS-DMA paths (A/B: -DCFSAR_PRE_OPATH1_ONLY)
#ifdef CFSAR_PRE_OPATH1_ONLY
constexpr int kPreMinOpath = 1, kPreMaxOpath = 1;
#else
constexpr int kPreMinOpath = 1, kPreMaxOpath = 2;
#endif
#ifndef CFSAR_EPI_PIPE
#define CFSAR_EPI_PIPE 0      // 1 = software-pipelined epilogue stores (measured neutral to negative: profiles/r03_gemm_anatomy.md)
#endif

constexpr int kActRelu = 100;                   // internal: plain instance with ReLU (RN50 1x1 convs: relu(bn(conv)), few_shot.py:222-223)
constexpr int TM = 256, TN = 256;              // output tile
constexpr int ROWB = 128;                      // bytes of K per row per K tile (64 bf16)
constexpr int STAGE = (TM + TN) * ROWB;        // 64 KiB: X rows [0, 32 KiB), W rows [32 KiB, 64 KiB)
constexpr int EPI_OFF = 2 * STAGE;             // wave-private epilogue slabs above the two stages
constexpr int EPI_SLAB = 32 * 128;             // 32 rows x 64 two-byte values
constexpr int LDS_BYTES = EPI_OFF + 8 * EPI_SLAB;   // 163 840 B = all of a CU's LDS

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

// LDS-DMA, 16 B per lane: LDS[m0 + lane*16 .. +16] = *(src).  M0 is compiler-reserved: saved / restored inside the statement.
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

#ifdef CFSAR_DEV
// A/B only: the same with a cache policy on the load (1 = nt, 2 = sc1)
template <int POL>
__device__ __forceinline__ void glds16_asm_pol(const char* src, unsigned lds_addr) {
    unsigned keep;
    if constexpr (POL == 1)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
}
#endif

// x * sigmoid(1.702 x) for four values, few_shot.py:614-616: v_exp_f32 + v_rcp_f32 (1 ulp each).
//
// History of a fault (round 2), kept here because this function was its first victim.  Builds of the LN-folded instances of this
// kernel that contained packed-fp32 VALU instructions -- v_pk_mul_f32 for the 1 / std row scale in front of this function and for the
// accumulator initialisation d x std -- occasionally produced wrong values in lanes 48-63 of the HIGH register of ONE packed pair: here
// as exact zeros of the c_fc output (a stale, huge scaled input -> 2^z = inf -> rcp = 0), in the QKV instance as a missing d x std.
// Rate: up to 13 % of the tiles in one build, once per ~100 launches in another and only with a second kernel on the chip; barriers or
// nops next to the affected code made it MORE frequent, other schedules hid it.  The stand-alone replays in tools/ubench/ (trans_war,
// valu_vmem_hazard, pk_mfma_hazard, pk_under_mfma) do not reproduce it and no spill is involved (it also hit a build with no scratch
// memory), so the root cause is NOT isolated.  What removes it in every build tried: compiling this file without packed-fp32
// instructions (clip-fsar_amd/build.py SOURCE_FLAGS; 0 of 1 500 stress launches against 44 of 150, same speed).  An earlier workaround
// (pinning the transcendental operands with empty asm statements) is no longer needed and was removed.  Guards:
// tests/test_gpu_kernels.py::test_gemm_lnfold_* and ::test_vit_gemms_are_bit_stable_under_a_second_stream, tools/stream_stress.py.
#ifdef CFSAR_GELU_UNFUSED                      // A/B builds: row scale as its own multiply in front of quick_gelu4
constexpr bool kGeluRowFused = false;
#else
constexpr bool kGeluRowFused = true;
#endif
__device__ __forceinline__ void quick_gelu4(float (&v)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float z = -1.702f * 1.4426950408889634f * v[j];
        v[j] = v[j] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
    }
}

// sum over each aligned group of 8 consecutive lanes, result in all 8: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_sum8(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    return v;
}

// 16-byte global store with a cache policy: 0 = default (write-back, line stays in this XCD's L2), 1 = nt, 2 = sc1 nt
// (write-through: the line is not kept, MI355X_MICROARCH.md "stores of each flavour"; nt on top of it measured another 1 %)
template <int POLICY>
__device__ __forceinline__ void store16(void* dst, u32x4 v) {
    if constexpr (POLICY == 1) {
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
    } else if constexpr (POLICY == 2) {              // write-through + non-temporal (round 3: QKV 825 -> 814 us, c_fc 1180 -> 1171 us against sc1 alone)
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
#ifdef CFSAR_DEV
    } else if constexpr (POLICY == 3) {              // A/B only: write-through alone (the round-2 policy)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
    } else if constexpr (POLICY == 4) {              // A/B only: system scope
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
    } else if constexpr (POLICY == 5) {              // A/B only: system scope + nt
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
    } else if constexpr (POLICY == 6) {              // A/B only: sc0
        asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
#endif
    } else {
        *reinterpret_cast<u32x4*>(dst) = v;
    }
}

// linear tile index (inside one XCD's contiguous range) -> (row band, column tile).  `group` row bands are finished before
// the next group starts; inside a group either the band index runs fastest (colfast = 0: concurrent workgroups cover
// group x 32/group tiles) or the column index does (colfast = 1: ~32/tiles_n bands x every column at a time).
__device__ __forceinline__ void tile_of(int lin, int tiles_m, int tiles_n, int group, int colfast, int& tm, int& tn) {
    const int per = group * tiles_n;
    const int gid = lin / per, first = gid * group;
    const int gsz = tiles_m - first < group ? tiles_m - first : group;
    const int rem = lin - gid * per;
    if (colfast) {
        tm = first + rem / tiles_n;
        tn = rem - (rem / tiles_n) * tiles_n;
    } else {
        tm = first + rem % gsz;
        tn = rem / gsz;
    }
}

// ---- epilogue: the wave's 128 x 64 accumulator tile -> global, 32 rows at a time through the wave's 4 KiB slab.
// Accumulator layout (operands swapped): acc[mi][ni][4g + j] = C[row 32 mi + (lane & 31)][col 32 ni + 8 g + 4 (lane >> 5) + j].
// The bias is already IN the accumulators (they are initialised with it, see the kernel), so a pass is: [activation] -> pack to
// 2 bytes -> 8 ds_write_b64 -> 4 ds_read_b128 -> [+ residual] -> 4 global stores of 16 bytes per lane.
// Write: the lane's 4 columns of group (ni, g) go to 8-byte slot ((2 q + hi) ^ (row & 15)) of row `row` (q = 4 ni + g): 16
// consecutive lanes hit 16 different slots -> conflict-free ds_write_b64.  Read: lane (rr = lane >> 3, Q = lane & 7) takes the
// 16-byte chunk Q of rows rr, rr + 8, ...: its two halves are slots (2Q) ^ f and (2Q + 1) ^ f, i.e. the aligned pair (Q ^ (f >> 1))
// with the halves swapped when f is odd -> one conflict-free ds_read_b128 + a per-lane-constant select.  8 lanes then store one
// whole 128-byte line, 8 rows per wave-instruction.  FULL: every row and column of the wave tile is inside the matrix
// (straight-line code, no predicates); otherwise rows are clamped for the loads and the stores are predicated.
// ROWSCALE (LN-folded consumer, see the kernel): the accumulator of row 32 mi + lr is multiplied by rscale[mi] = 1 / std(row)
// before the activation.  STATS (residual producer): per row, the sum and the sum of squares of the 64 STORED (rounded) values of
// this wave are written to stats_out[row][slot = column / 64] -- the LayerNorm statistics of the next LN-folded GEMM come from
// these partials (cfsar_ln_stats_finalize), so the residual stream is never re-read for them.
// (float)pair.half + c in one fp32 VALU instruction (v_fma_mix_f32 reads the fp16 half of the dword directly; HALF = 0 low, 1 high)
template <int HALF>
__device__ __forceinline__ float half_plus(unsigned pair, float c) {
    float r;
    if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(c));
    return r;
}

// CH: byte distance between the slab's 8-row chunks (1 024 = one contiguous 4 KiB slab; the 4-wave kernel interleaves the chunks of its waves)
template <typename TO, int ACT, bool HAS_RES, int STORE, bool FULL, bool ROWSCALE = false, bool PRE = false, bool HB = false, int NMI = 4, int CH = 1024>
__device__ __forceinline__ void epilogue_rows(f32x16 (&acc)[NMI][2], const VitGemmArgs& p, int mb, int nb, int lane, char* slab,
                                              const float (&rscale)[4], const u32x4 (&rv0)[4]) {
    typedef typename Vec2B<TO>::v4 TO4;
    typedef typename Vec2B<TO>::v8 TO8;
    const int lr = lane & 31, hi = lane >> 5;
    const int rr = lane >> 3, Q = lane & 7;
    // per-frame column sums of the stored output (VitGemmArgs::colsum): the fp16 LN-folded QuickGELU instance only
    constexpr bool COLSUM = std::is_same<TO, _Float16>::value && ACT == CFSAR_ACT_QUICKGELU && ROWSCALE && !HAS_RES && !HB;
    // The sums are taken in FIXED POINT (int32, 2^-12 units; |value| clamped to 1 000): integer addition is associative, so a frame's sum does
    // not depend on how its rows fall into tiles, wave tiles or lanes -- an episode's result stays bit-identical whatever batch it is served
    // in (fp16 / fp32 partial sums regroup with the frame's row offset, and ONE flipped bit anywhere re-draws the whole tower's rounding noise).
    typedef _Float16 cs_h8 __attribute__((ext_vector_type(8)));
    // cs1: running sum over ALL row steps so far (raw bit patterns of value + magic: the constant's share is taken out once, at the end; all
    // arithmetic mod 2^32); cs0: its snapshot at the lane's first row of the tile's SECOND frame.  A lane's rows are rr + 8 j, j = 0 .. 4 NMI - 1,
    // so the snapshot step jb is one of two consecutive values over the wave (cs_jb0, cs_jb0 + 1): only those steps pay for it.
    unsigned cs0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cs1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int cs_bnd = 0, cs_jb = 0, cs_jb0 = 0;
    if constexpr (COLSUM) {
        if (p.colsum != nullptr) {
            cs_bnd = (mb / p.corr_tokens + 1) * p.corr_tokens - mb;      // rows of this wave tile in its first frame
            const int jb = (cs_bnd - rr + 7) >> 3;                        // steps of this lane inside the first frame
            cs_jb = jb < 4 * NMI ? jb : 4 * NMI;
            cs_jb0 = __builtin_amdgcn_readfirstlane(cs_bnd >> 3);         // the smallest jb of the wave (rr = 7)
        }
    }
    const bool colok = FULL || nb + 64 <= p.N;              // whole-wave predicate (N % 64 == 0)
    const int ncl = colok ? nb : p.N - 64;                  // clamped column base: loads stay in bounds
    char* wr = CH == 1024 ? slab + lr * 128 : slab + (lr >> 3) * CH + (lr & 7) * 128;
    const int wsw = lr & 15;
    const bool swap_halves = rr & 1;
    // byte offset of (row mb + rr, column ncl + 8 Q); rows advance by 8 per read-back step (M * ldo * 2 < 4 GiB: launcher)
    const unsigned ostep = (unsigned)p.ldo * 16u, rstep = (unsigned)p.ldr * 16u;
    char* outp = reinterpret_cast<char*>(p.out) + ((size_t)(mb + rr) * p.ldo + ncl + 8 * Q) * 2;
    const char* resp = HAS_RES ? reinterpret_cast<const char*>(p.res) + ((size_t)(mb + rr) * p.ldr + ncl + 8 * Q) * 2 : nullptr;
    const int rd0 = rr * 128 + ((Q ^ ((rr >> 1) & 7)) << 4);       // row rr + 8 it: + it * 1024, chunk ^ (4 it & 7) << 4
    // HB: head-blocked output (see VitGemmArgs).  The wave's 64 columns are one (which, head) block; row mb + rr + 8 s is token
    // t0 + 8 s of frame f0, wrapping into the next frame at most once (T >= 128 > 8 * 15).
    char* hb_base = nullptr;                                // address of step 0
    size_t hb_wrap = 0;                                     // added from the first step that falls into the next frame
    int hb_wrap_step = 16;
    if constexpr (HB) {
        const int row0 = mb + rr;
        const int f0 = row0 / p.hb_tokens, t0 = row0 - f0 * p.hb_tokens;
        const int blk = ncl >> 6, which = blk / p.hb_heads, h = blk - which * p.hb_heads;
        const size_t frame_bytes = (size_t)p.hb_heads * p.hb_tokens * 384;
        hb_base = reinterpret_cast<char*>(p.out) + (size_t)f0 * frame_bytes + ((size_t)h * p.hb_tokens + t0) * 384 + which * 128 + Q * 16;
        hb_wrap = frame_bytes - (size_t)p.hb_tokens * 384;
        hb_wrap_step = (p.hb_tokens - t0 + 7) >> 3;         // first step s with t0 + 8 s >= tokens
    }
    auto out_addr = [&](int step) __attribute__((always_inline)) -> char* {
        if constexpr (HB) return hb_base + step * 3072 + (step >= hb_wrap_step ? hb_wrap : (size_t)0);
        else return outp + (size_t)step * ostep;
    };
    // residual rows: the loads of pass mi + 1 are issued before pass mi is processed, so no pass waits for its own loads (the stream
    // is read exactly once: these are HBM / MALL latencies)
    u32x4 rvn[4];
    auto load_res = [&](int mi_) __attribute__((always_inline)) {
        if constexpr (HAS_RES) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int step = mi_ * 4 + it;
                if (FULL || mb + rr + step * 8 < p.M) rvn[it] = *reinterpret_cast<const u32x4*>(resp + (size_t)step * rstep);
                else rvn[it] = u32x4{0, 0, 0, 0};
            }
        }
    };
    // PRE: pass 0's rows come in from the caller, which issued their loads before the last K step of the tile (residual_prefetch)
    if constexpr (PRE) {
#pragma unroll
        for (int it = 0; it < 4; ++it) rvn[it] = rv0[it];
    } else {
        load_res(0);
    }
    // One 32-row pass = convert (8 groups of 4 values: [row scale] [activation] pack, ds_write_b64) -> read back (4 ds_read_b128) ->
    // finish (4 x: [+ residual, statistics] 16-byte store).  CFSAR_EPI_PIPE = 1 (A/B builds) issues the four finish steps of pass
    // mi - 1 BETWEEN the convert groups of pass mi; measured neutral (QKV, c_fc) to negative (residual instances): what the stores
    // cost is not issue time inside the epilogue but memory-system interference with every workgroup's operand loads during the K
    // loops that follow (profiles/r03_gemm_anatomy.md: workgroups that skip their stores slow down exactly like those that store).
    // per 32-row pass: k = -1.702 log2(e) / std and std of the pass's row (two registers live at a time)
    auto gelu_consts = [&](int mi, float& gk, float& gsd) __attribute__((always_inline)) {
        if constexpr (kGeluRowFused && ROWSCALE && ACT == CFSAR_ACT_QUICKGELU) {
            float rs = rscale[mi];
            asm volatile("" : "+v"(rs));            // computed HERE, not hoisted into the K loop's last steps (24 spilled registers)
            gk = -1.702f * 1.4426950408889634f * rs;
            gsd = __builtin_amdgcn_rcpf(rs);
        }
    };
    auto convert_group = [&](int mi, int q, float gk, float gsd) __attribute__((always_inline)) {
        const int ni = q >> 2, g = q & 3;
        TO4 o;
        float v[4];
        if constexpr (kGeluRowFused && ROWSCALE && ACT == CFSAR_ACT_QUICKGELU) {
            // a / std * sigmoid(1.702 a / std) = a / (std + std * 2^(a * k)), k = -1.702 log2(e) / std: the row scale rides inside the
            // sigmoid's denominator (one fma) instead of costing a multiply per element: 3 VALU + 2 transcendental instructions
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = acc[mi][ni][4 * g + j];
                const float e = __builtin_amdgcn_exp2f(a * gk);
                v[j] = a * __builtin_amdgcn_rcpf(__builtin_fmaf(e, gsd, gsd));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[mi][ni][4 * g + j];
                if constexpr (ROWSCALE) v[j] *= rscale[mi];
            }
            if constexpr (ACT == CFSAR_ACT_QUICKGELU) quick_gelu4(v);
            if constexpr (ACT == kActRelu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (TO)v[j];
        const int slot = (2 * (ni * 4 + g) + hi) ^ wsw;
        *reinterpret_cast<TO4*>(wr + slot * 8) = o;
    };
    auto read_back = [&](u32x4 (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 4; ++it)                       // row r = 8 it + rr: (r >> 1) & 7 = ((rr >> 1) + 4 it) & 7
            d[it] = *reinterpret_cast<const u32x4*>(slab + it * CH + (rd0 ^ ((it & 1) << 6)));
    };
    auto finish = [&](int mi, int it, u32x4 x, u32x4 r) __attribute__((always_inline)) {
        const bool rowok = FULL || mb + rr + (mi * 4 + it) * 8 < p.M;
        if (swap_halves) x = u32x4{x[2], x[3], x[0], x[1]};
        if constexpr (HAS_RES && std::is_same<TO, _Float16>::value) {
            // x += residual as four packed fp16 adds; the row statistics of the STORED values as v_dot2_f32_f16 (exact fp16
            // products, fp32 accumulation): 12 VALU instructions per 8 elements instead of ~56 through fp32
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 ones = {(_Float16)1.0f, (_Float16)1.0f};
            float ps = 0.f, pq = 0.f;
            const TO8 s8 = __builtin_bit_cast(TO8, x) + __builtin_bit_cast(TO8, r);        // 4 x v_pk_add_f16
            x = __builtin_bit_cast(u32x4, s8);
            if (p.stats_out) {                           // wave-uniform
                const h2 s01 = __builtin_shufflevector(s8, s8, 0, 1), s23 = __builtin_shufflevector(s8, s8, 2, 3);
                const h2 s45 = __builtin_shufflevector(s8, s8, 4, 5), s67 = __builtin_shufflevector(s8, s8, 6, 7);
                ps = __builtin_amdgcn_fdot2(s01, ones, ps, false);
                pq = __builtin_amdgcn_fdot2(s01, s01, pq, false);
                ps = __builtin_amdgcn_fdot2(s23, ones, ps, false);
                pq = __builtin_amdgcn_fdot2(s23, s23, pq, false);
                ps = __builtin_amdgcn_fdot2(s45, ones, ps, false);
                pq = __builtin_amdgcn_fdot2(s45, s45, pq, false);
                ps = __builtin_amdgcn_fdot2(s67, ones, ps, false);
                pq = __builtin_amdgcn_fdot2(s67, s67, pq, false);
                // the 8 lanes Q = 0..7 of a row are consecutive: quad butterflies + half-row mirror as DPP VALU ops (a
                // __shfl_xor is a ds_bpermute round trip through the LDS pipe: 96 dependent ones per tile before this)
                ps = dpp_sum8(ps);
                pq = dpp_sum8(pq);
                if (Q == 0 && (FULL || (rowok && colok)))
                    *reinterpret_cast<float2*>(p.stats_out + ((size_t)(mb + rr + (mi * 4 + it) * 8) * p.stats_slots + (nb >> 6)) * 2) =
                        make_float2(ps, pq);
            }
        } else if constexpr (HAS_RES) {
            const TO8 a = __builtin_bit_cast(TO8, x), b = __builtin_bit_cast(TO8, r);
            TO8 sres;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = (float)a[j] + (float)b[j];
                if (p.relu) t = fmaxf(t, 0.0f);                     // relu(bn3(conv3) + identity), few_shot.py:224-226
                sres[j] = (TO)t;
            }
            x = __builtin_bit_cast(u32x4, sres);
        }
#ifdef CFSAR_DEV
        if ((p.dbg & 16) && x[0] != 0x7fc12345u) return;               // ablation: everything but the global stores
        if ((p.dbg & 32) && ((blockIdx.x >> 3) & 1) && x[0] != 0x7fc12345u) return;   // ... on every other workgroup of each XCD only
#endif
        if (FULL || (rowok && colok)) store16<STORE>(out_addr(mi * 4 + it), x);
        if constexpr (COLSUM) {
            if (p.colsum != nullptr) {                       // kernel-uniform
                // per element: a packed fp16 clamp, one v_fma_mix_f32 (half + magic) and ONE integer add of the raw bits (2.5 VALU slots; the
                // fp32 clamp + selects form took 9, the masked two-accumulator form 5: the epilogue is instruction bound, r04 session s23 (script archived))
                typedef _Float16 cs_h2 __attribute__((ext_vector_type(2)));
                const int j_ = mi * 4 + it;
                if (j_ == cs_jb0 || j_ == cs_jb0 + 1) {               // wave-uniform: a lane of this wave may enter the second frame at this step
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs0[e] = (j_ == cs_jb) ? cs1[e] : cs0[e];
                }
                constexpr float kMagic = 1.5f * 2048.0f;             // ulp of (v + 1.5 x 2^11) = 2^-12 for |v| < 2^10: its low mantissa bits ARE v in fixed point
                const cs_h2 lim = {(_Float16)1000.0f, (_Float16)1000.0f};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned xw = x[k];                             // (a scalar copy first: __builtin_bit_cast straight from the vector ELEMENT x[k] reads element 0 for every k -- hipcc 7.2)
                    cs_h2 hv = __builtin_bit_cast(cs_h2, xw);
                    hv = __builtin_elementwise_min(__builtin_elementwise_max(hv, -lim), lim);
                    const unsigned cw = __builtin_bit_cast(unsigned, hv);
                    const unsigned b0 = __builtin_bit_cast(unsigned, half_plus<0>(cw, kMagic)), b1 = __builtin_bit_cast(unsigned, half_plus<1>(cw, kMagic));
                    cs1[2 * k] += (FULL || rowok) ? b0 : __builtin_bit_cast(unsigned, kMagic);          // a row past M counts as 0
                    cs1[2 * k + 1] += (FULL || rowok) ? b1 : __builtin_bit_cast(unsigned, kMagic);
                }
            }
        }
    };
#if CFSAR_EPI_PIPE
    u32x4 dprev[4], rvprev[4];
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi) {
        u32x4 rv[4];
        if constexpr (HAS_RES) {
#pragma unroll
            for (int it = 0; it < 4; ++it) rv[it] = rvn[it];
        }
        if (mi < NMI - 1) load_res(mi + 1);
        float gk = 0.f, gsd = 0.f;
        gelu_consts(mi, gk, gsd);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            convert_group(mi, q, gk, gsd);
            if (mi > 0 && (q & 1)) {
                finish(mi - 1, q >> 1, dprev[q >> 1], rvprev[q >> 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        read_back(dprev);
        if constexpr (HAS_RES) {
#pragma unroll
            for (int it = 0; it < 4; ++it) rvprev[it] = rv[it];
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) finish(NMI - 1, it, dprev[it], rvprev[it]);
#else
#pragma unroll
    for (int mi = 0; mi < NMI; ++mi) {
        u32x4 rv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if constexpr (HAS_RES) rv[it] = rvn[it];
        }
        if (mi < NMI - 1) load_res(mi + 1);
        float gk = 0.f, gsd = 0.f;
        gelu_consts(mi, gk, gsd);
#pragma unroll
        for (int q = 0; q < 8; ++q) convert_group(mi, q, gk, gsd);
        u32x4 d[4];
        read_back(d);
#pragma unroll
        for (int it = 0; it < 4; ++it) finish(mi, it, d[it], rv[it]);
    }
#endif
    if constexpr (COLSUM) {
        if (p.colsum != nullptr) {
            // sum over the 8 row groups rr (lanes differing in bits 3, 4, 5): row_ror:8, swizzle xor 16, bpermute xor 32 (integer adds: any order)
            const int partner = (lane ^ 32) << 2;
            constexpr unsigned kbits = 0x45400000u;                  // bits of 1.5 x 2^11 (kMagic above)
            static_assert(__builtin_bit_cast(unsigned, 1.5f * 2048.0f) == kbits, "magic constant");
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int a0 = (int)((cs_jb >= 4 * NMI ? cs1[j] : cs0[j]) - (unsigned)cs_jb * kbits);     // the first frame's rows (all of them if the lane never left it)
                int a1 = (int)(cs1[j] - (unsigned)(4 * NMI) * kbits) - a0;                         // all rows - first frame's rows = second frame's rows
                a0 += __builtin_amdgcn_update_dpp(0, a0, 0x128, 0xF, 0xF, false);
                a1 += __builtin_amdgcn_update_dpp(0, a1, 0x128, 0xF, 0xF, false);
                a0 += __builtin_amdgcn_ds_swizzle(a0, 0x401F);
                a1 += __builtin_amdgcn_ds_swizzle(a1, 0x401F);
                a0 += __builtin_amdgcn_ds_bpermute(partner, a0);
                a1 += __builtin_amdgcn_ds_bpermute(partner, a1);
                cs0[j] = (unsigned)a0;
                cs1[j] = (unsigned)a1;
            }
            if (rr == 0 && colok && mb < p.M) {
                int* dst = reinterpret_cast<int*>(p.colsum) + ((size_t)(mb / (32 * NMI)) * 2) * p.N + ncl + 8 * Q;
                *reinterpret_cast<u32x4*>(dst) = u32x4{cs0[0], cs0[1], cs0[2], cs0[3]};
                *reinterpret_cast<u32x4*>(dst + 4) = u32x4{cs0[4], cs0[5], cs0[6], cs0[7]};
                *reinterpret_cast<u32x4*>(dst + p.N) = u32x4{cs1[0], cs1[1], cs1[2], cs1[3]};
                *reinterpret_cast<u32x4*>(dst + p.N + 4) = u32x4{cs1[4], cs1[5], cs1[6], cs1[7]};
            }
        }
    }
}

// ---- MODE 6 epilogue ("wide" residual; the fp16 numerics mode, round 4).  What differs from epilogue_rows<HAS_RES>: the GEMM result is
// NOT rounded to fp16 before the residual add (round 3 added two fp16 numbers in packed fp16: two roundings per stream update, and the
// first one -- 2^-12 of the update -- is what an fp32 or two-word stream would otherwise be free of).  The accumulators cross the LDS as
// fp32, the residual is added in fp32 and the sum is rounded ONCE: hi = fp16(s); with res_lo != NULL the stream carries a second word
// lo = fp16(s - hi) (hi + lo ~ 22 bits; the next residual add reads both, the LN-folded consumers read hi only and take its
// statistics).  One pass = one 32-row x 32-column half of an MFMA tile (4 KiB of fp32 = the wave's slab): 4 ds_write_b128, 4
// ds_read_b128, 2 x (16 B hi [+ 16 B lo] in, 16 B [+ 16 B] out) per lane; 4 lanes own the 64 contiguous bytes of a row's half.
// 16-byte slot s of row r sits at physical slot s ^ sw(r), sw(r) = ((r >> 1) & 7) ^ ((r & 1) << 2): distinct over 8 consecutive rows (the
// 8-lane groups of ds_write_b128) and over the row sets of ds_read_b128's 16-lane groups (MI355X_MICROARCH.md, LDS table).
// hi + lo and c - h on fp16 HALVES of packed dwords (HALF = 0: the low, 1: the high half) in ONE fp32 VALU instruction each: v_fma_mix_f32 reads
// the halves directly (op_sel picks them), so the packed stream words need no conversion instructions.  Pure register arithmetic: nothing for
// the compiler to mis-schedule (no memory operands; cf. profiles/r04_fault_audit.md on the asm loads).
template <int HALF>
__device__ __forceinline__ float mix_sum(unsigned a, unsigned b) {          // (float)a.half + (float)b.half: exact in fp32 for hi + remainder
    float r;
    if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int HALF>
__device__ __forceinline__ float mix_sub(unsigned pair, float c) {
    float r;
    if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(c));
    return r;
}

template <int STORE, bool FULL, int NMI, int CH = 1024>
__device__ __forceinline__ void epilogue_rows_wide(f32x16 (&acc)[NMI][2], const VitGemmArgs& p, int mb, int nb, int lane, char* slab) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lr = lane & 31, hi = lane >> 5;
    const int rr = lane >> 2, Q = lane & 3;                 // read-back: row rr + 16 it, columns 8 Q .. 8 Q + 7 of the half
    const bool colok = FULL || nb + 64 <= p.N;
    const int ncl = colok ? nb : p.N - 64;
    const int wsw = ((lr >> 1) & 7) ^ ((lr & 1) << 2);
    const int rsw = ((rr >> 1) & 7) ^ ((rr & 1) << 2);      // rows rr and rr + 16 swizzle alike
    char* wr = CH == 1024 ? slab + lr * 128 : slab + (lr >> 3) * CH + (lr & 7) * 128;
    const char* rd = CH == 1024 ? slab + rr * 128 : slab + (rr >> 3) * CH + (rr & 7) * 128;
    const bool has_lo = p.res_lo != nullptr;                // kernel-uniform
    const size_t eoff = ((size_t)(mb + rr) * p.ldo + ncl + 8 * Q) * 2;       // residual in place: ldr == ldo
    char* outp = reinterpret_cast<char*>(p.out) + eoff;
    char* lop = has_lo ? reinterpret_cast<char*>(p.res_lo) + eoff : nullptr;
    const unsigned rstep = (unsigned)p.ldo * 32u;           // 16 rows
    u32x4 rh[2], rl[2], nh[2], nl[2];
    auto load_res = [&](int pass) __attribute__((always_inline)) {          // pass = 2 mi + ni
        const int mi_ = pass >> 1, ni_ = pass & 1;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int step = mi_ * 2 + it;
            const bool ok = FULL || mb + rr + step * 16 < p.M;
            const size_t o = (size_t)step * rstep + ni_ * 64;
            nh[it] = ok ? *reinterpret_cast<const u32x4*>(outp + o) : u32x4{0, 0, 0, 0};
            nl[it] = (ok && has_lo) ? *reinterpret_cast<const u32x4*>(lop + o) : u32x4{0, 0, 0, 0};
        }
    };
    load_res(0);
    float ps[2] = {0.f, 0.f}, pq[2] = {0.f, 0.f};
#pragma unroll
    for (int pass = 0; pass < 2 * NMI; ++pass) {
        const int mi = pass >> 1, ni = pass & 1;
#pragma unroll
        for (int it = 0; it < 2; ++it) { rh[it] = nh[it]; rl[it] = nl[it]; }
        if (pass + 1 < 2 * NMI) load_res(pass + 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f4 v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
            *reinterpret_cast<f4*>(wr + (((2 * g + hi) ^ wsw) << 4)) = v;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const f4 a = *reinterpret_cast<const f4*>(rd + it * 2 * CH + (((2 * Q) ^ rsw) << 4));
            const f4 b = *reinterpret_cast<const f4*>(rd + it * 2 * CH + (((2 * Q + 1) ^ rsw) << 4));
#ifdef CFSAR_WIDE_PLAIN_C                                  // the plain C form (A/B: `python clip-fsar_amd/build.py --variant plainc -DCFSAR_WIDE_PLAIN_C`)
            const h8 xh = __builtin_bit_cast(h8, rh[it]), xl = __builtin_bit_cast(h8, rl[it]);
            h8 oh, ol;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = j < 4 ? a[j] : b[j - 4];
                const float sum = d + ((float)xh[j] + (float)xl[j]);          // hi + lo is exact in fp32
                const _Float16 h = (_Float16)sum;
                oh[j] = h;
                ol[j] = (_Float16)(sum - (float)h);
            }
#else
            // Per pair of elements: the stream's words enter the fp32 sum straight from their packed halves (v_fma_mix_f32: no fp16 -> fp32
            // conversion instructions), hi = one packed conversion, remainder = sum - hi again through v_fma_mix_f32 on the packed hi:
            // 4 VALU instructions per element where the plain C form compiled to 8, bit-identical results (the epilogue is VALU-bound: profiles/r04_gemm_plateau.md)
            u32x4 ohw, olw;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d0 = k < 2 ? a[2 * k] : b[2 * k - 4], d1 = k < 2 ? a[2 * k + 1] : b[2 * k - 3];
                const float s0 = d0 + mix_sum<0>(rh[it][k], rl[it][k]);                  // hi + lo is exact in fp32
                const float s1 = d1 + mix_sum<1>(rh[it][k], rl[it][k]);
                const h2 hp = {(_Float16)s0, (_Float16)s1};
                const unsigned hb = __builtin_bit_cast(unsigned, hp);
                const h2 lp = {(_Float16)mix_sub<0>(hb, s0), (_Float16)mix_sub<1>(hb, s1)};
                ohw[k] = hb;
                olw[k] = __builtin_bit_cast(unsigned, lp);
            }
            const h8 oh = __builtin_bit_cast(h8, ohw), ol = __builtin_bit_cast(h8, olw);
#endif
            const int step = mi * 2 + it;
            const bool rowok = FULL || mb + rr + step * 16 < p.M;
            if (p.stats_out) {                               // wave-uniform: statistics of the STORED hi words (what the consumer reads)
                const h2 ones = {(_Float16)1.0f, (_Float16)1.0f};
                const h2 s01 = __builtin_shufflevector(oh, oh, 0, 1), s23 = __builtin_shufflevector(oh, oh, 2, 3);
                const h2 s45 = __builtin_shufflevector(oh, oh, 4, 5), s67 = __builtin_shufflevector(oh, oh, 6, 7);
                float s_ = 0.f, q_ = 0.f;
                s_ = __builtin_amdgcn_fdot2(s01, ones, s_, false);
                q_ = __builtin_amdgcn_fdot2(s01, s01, q_, false);
                s_ = __builtin_amdgcn_fdot2(s23, ones, s_, false);
                q_ = __builtin_amdgcn_fdot2(s23, s23, q_, false);
                s_ = __builtin_amdgcn_fdot2(s45, ones, s_, false);
                q_ = __builtin_amdgcn_fdot2(s45, s45, q_, false);
                s_ = __builtin_amdgcn_fdot2(s67, ones, s_, false);
                q_ = __builtin_amdgcn_fdot2(s67, s67, q_, false);
                s_ += dpp_move<0xB1>(s_);                    // the 4 lanes Q = 0..3 of a row are one quad
                q_ += dpp_move<0xB1>(q_);
                s_ += dpp_move<0x4E>(s_);
                q_ += dpp_move<0x4E>(q_);
                if (ni == 0) { ps[it] = s_; pq[it] = q_; }
                else {
                    ps[it] += s_; pq[it] += q_;
                    if (Q == 0 && (FULL || (rowok && colok)))
                        *reinterpret_cast<float2*>(p.stats_out + ((size_t)(mb + rr + step * 16) * p.stats_slots + (nb >> 6)) * 2) =
                            make_float2(ps[it], pq[it]);
                }
            }
            if (FULL || (rowok && colok)) {
                const size_t o = (size_t)step * rstep + ni * 64;
                store16<STORE>(outp + o, __builtin_bit_cast(u32x4, oh));
                if (has_lo) store16<STORE>(lop + o, __builtin_bit_cast(u32x4, ol));
            }
        }
    }
}

}  // namespace
