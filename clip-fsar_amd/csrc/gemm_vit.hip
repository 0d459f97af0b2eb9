// The bf16 GEMMs of the CLIP ViT blocks at batch scale: QKV / c_fc (bias [+ QuickGELU], bf16 out) and out_proj / c_proj
// (bias + fp16 residual stream, fp16 out) -- few_shot.py:623,626-628,633-640 -- as ONE persistent kernel whose operand
// pipeline never drains between output tiles.
//
// What it keeps from gemm_kernel_p12 (gemm.hip): 256 x 256 output tile, 512 threads = two waves per SIMD (2(M) x 4(N) waves,
// 128 x 64 wave tiles, 128 accumulators), 128-byte K tiles fetched as whole cache lines, two 64 KiB LDS stages with
// XOR-swizzled 128-byte rows, one barrier per K tile, operands swapped so a lane owns 4 consecutive output columns.
//
// What is new (round 2; the numbers that motivated it are in profiles/r02_gemm_pmc_baseline.md):
//   * The epilogue no longer aliases the operand stages.  Each wave transposes ONE 32-row x 64-column slab at a time through a
//     private 4 KiB region above the stages (values packed to 2 bytes BEFORE the LDS round trip, 8-byte conflict-free writes,
//     16-byte reads, whole 128-byte output lines per 8 lanes).  p12 staged 17 KiB of fp32 per wave over the stages, so every
//     tile paid a pipeline drain + a 7 K-cycle refill (measured) around a 6.8 K-cycle epilogue.
//   * Because the stages stay intact, the last K tile of an output tile already stages K tile 0 of the NEXT output tile (and, on
//     the register path, holds K tile 1 in flight through the epilogue): the refill disappears and the L2 -> LDS stream keeps
//     running while the epilogue's VALU / store work executes.
//   * Two operand paths behind one schedule (OPATH): 0 = global_load -> VGPR -> ds_write_b128 (p12's; loads run two K tiles ahead),
//     1 = LDS-DMA (global_load_lds_dwordx4 issued from inline asm, one K tile ahead, no ds_write traffic, 32 fewer VGPRs).
//   * Tile walk and store cache policy are parameters (the XCD's L2 holds 4 MiB: what matters is which tiles its 32 CUs work
//     on at the same time and whether 128 KiB of output per tile is allowed to evict the operands).
//   * Register pressure at the tile boundary decides more than any of the above (profiles/r02_gemm_ab.md section 5): the bias and
//     LayerNorm-statistics vectors of a tile are loaded AFTER the previous tile's epilogue (no spills in the LDS-DMA instances),
//     the LN-folded instances start their accumulators from one rank-2 MFMA instead of 128 multiplies, the residual instances
//     add in packed fp16, take their row statistics with v_dot2_f32_f16 + DPP sums and fetch the residual rows one pass ahead.
//   * Built without packed-fp32 VALU instructions (build.py SOURCE_FLAGS; the story is above quick_gelu4).
#include <type_traits>
#include <utility>

#include "common.h"
#include "gemm_vit.h"

#include "gemm_vit_epi.h"

#ifdef CFSAR_DEV
int g_cfsar_walk_enable = 0, g_cfsar_walk_phase = 0;      // experiment (dbg bit 25 of cfsar_debug_set_vit_dbg): see attention.hip too
#endif

namespace {

// OPATH 0: register-staged operands (global_load_dwordx4 -> VGPR -> ds_write_b128), loads two K tiles ahead.
// OPATH 1: LDS-DMA operands (global_load_lds_dwordx4), one K tile ahead.
// MODE 0: out = act(A W^T + bias)                      (TI = bf16 operands; dev builds also instantiate TI = f16 for timing A/B)
// MODE 1: x   = x + A W^T + bias, fp16 in place        (TI = bf16; optional row-statistics partials, see epilogue_rows)
// MODE 2: out = act(LayerNorm(x) W^T + bias) computed WITHOUT materialising LayerNorm(x) (few_shot.py:605-611, 626-640):
//         with W' = W diag(gamma) (fp16, folded at init), c_n = sum_k W'_nk, d_n = sum_k beta_k W_nk + bias_n and the row's
//         (mean, std):   LN(x) W^T + bias = (x W'^T - mean c) / std + d.
//         The MFMAs run on the raw fp16 residual stream (TI = f16: same rate as bf16, 3 more mantissa bits); the accumulators
//         start at d_n * std_m - mean_m c_n, produced by ONE extra MFMA per 32x32 tile (a rank-2 outer product, operands split
//         into fp16 hi + lo parts: ~22 bits), and the epilogue multiplies the row by 1 / std_m.
// SHORTK: K = 128 (two K tiles; the LDS-DMA path only): the first step is also the second-to-last one.
// MIW: 32-row MFMA tiles per wave = output tile of 64 MIW rows: 4 (256 x 256, the batch-scale form) or 3 (192 x 256: at one or two episodes
// per call a GEMM is 2.2 rounds of 256-row tiles on 256 CUs that cost 3; 192-row tiles make it 2.9 rounds of 0.75: -25 %)
template <typename TI, typename TO, int ACT, int MODE, int OPATH, int STORE, bool SHORTK = false, int MIW = 4>
__global__ __launch_bounds__(512, 2) void vit_gemm_kernel(VitGemmArgs p) {
    static_assert(!SHORTK || OPATH == 1, "K = 128 runs on the LDS-DMA path");
    static_assert(MIW == 4 || (MIW == 3 && !SHORTK), "tile height 256 or 192");
    constexpr int TMv = 64 * MIW;                 // output tile rows
    constexpr int WR = 32 * MIW;                  // rows of one wave
    constexpr int NM = 2 * MIW;                   // MFMAs per sub-step
    constexpr int NL = MIW + 2;                   // fragment loads per sub-step
    constexpr bool WIDE = MODE == 6;              // residual with the fp32 add + optional second stream word (epilogue_rows_wide)
    constexpr bool HAS_RES = MODE == 1 || WIDE;
    constexpr bool LNFOLD = MODE == 2 || MODE == 4;       // 4 = LN-folded with head-blocked output (the QKV GEMM)
    constexpr bool HB = MODE == 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int lr = lane & 31, hi = lane >> 5;

    // ---- tile walk: virtual block ids b, b + grid, ...; id -> XCD-contiguous linear index -> (band, column)
    const int nt = p.ntiles, grid = (int)gridDim.x;
    const int xq = nt >> 3, xr = nt & 7;
    const int tiles_m = nt / p.tiles_n;
    auto origin = [&](int b, int& m0, int& n0) __attribute__((always_inline)) {
        const int xcd = b & 7;
        const int lin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
        int tm, tn;
        tile_of(lin, tiles_m, p.tiles_n, p.group, p.colfast, tm, tn);
#ifdef CFSAR_DEV
        if (p.dbg & (1 << 26)) {                           // experiment: row bands INTERLEAVED over the XCDs (band = 8 * (i / tiles_n) + xcd: the chip works on one
            const int i = b >> 3;                          // contiguous front of ~85 bands instead of 8 distant ranges); needs tiles_m % 8 == 0
            tm = (i / p.tiles_n) * 8 + (b & 7);
            tn = i - (i / p.tiles_n) * p.tiles_n;
        }
        if (p.dbg & (1 << 24)) tm = tiles_m - 1 - tm;      // experiment: row bands walked from the end (the producer's most recent rows first: Infinity Cache)
#endif
        m0 = tm * TMv;
        n0 = tn * TN;
#ifdef CFSAR_DEV
        if (p.dbg & 8) { m0 = 0; n0 = 0; }     // ablation: every workgroup reads tile (0, 0) (cache-hot operands)
#endif
    };
    // per-lane source offsets of the 4 + 4 staging pieces (8 rows x 128 B each; wave w owns pieces {w, w+8, w+16, w+24})
    auto offsets = [&](int m0, int n0, unsigned (&ox)[4], unsigned (&ow)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {                       // X: pieces 0 .. MIW - 1 are used
            const int row = (i * 8 + wave) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ swz(row);
            int gm = m0 + row;
            gm = gm < p.M ? gm : p.M - 1;
            int gn = n0 + row;
            gn = gn < p.N ? gn : p.N - 1;
            if (p.ha_tokens > 0) {                          // head-blocked A (kernel-uniform): row gm = f T + t, K tile kt = head kt
                const int f = gm / p.ha_tokens, t = gm - f * p.ha_tokens;
                ox[i] = ((unsigned)f * (unsigned)(p.K >> 6) * (unsigned)p.ha_tokens + (unsigned)t) * 128u + chunk * 16;
            } else {
                ox[i] = (unsigned)gm * (unsigned)p.lda * 2u + chunk * 16;
            }
            ow[i] = (unsigned)gn * (unsigned)p.ldw * 2u + chunk * 16;
        }
    };

    const int wr_off = wave * 1024 + lane * 16;          // register path: + piece i * 8192 (+ TM*ROWB for W) inside a stage
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);   // DMA path: same layout
    int rdX[4], rdW[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rx = wm * WR + i * 32 + lr;                                  // i >= MIW: unused
        rdX[i] = rx * ROWB + ((hi ^ swz(rx)) << 4);                        // sub-step ss: ^ (ss << 5)
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rw = wn * 64 + i * 32 + lr;
        rdW[i] = TM * ROWB + rw * ROWB + ((hi ^ swz(rw)) << 4);
    }

    const int nk = p.K / 64;
    int b = blockIdx.x;
    if (b >= nt) return;
    int m0 = 0, n0 = 0;                    // origin of the current output tile
    const unsigned akstride = p.ha_tokens > 0 ? (unsigned)p.ha_tokens * 128u : (unsigned)ROWB;     // bytes between K tiles of A
    f32x16 acc[MIW][2];
    // Bias / LayerNorm terms are added by ONE extra MFMA per 32x32 tile AFTER the last K step (a rank-1 / rank-2 outer product, see
    // tail_fold below); the accumulators start at zero (the first sub-step's MFMAs take the inline constant 0 as C).  What a lane
    // needs for it is tiny -- the bias (or c / d) of ITS two columns 32 ni + lr and, LN-folded, (mean | std) and 1 / std of ITS four
    // rows 32 mi + lr -- and is fetched by compiler-invisible loads during the second-to-last K step: nothing is loaded after the
    // epilogue's stores any more (in-order memory returns: a load behind 16 stores waits for all of them; round 2 loaded 8 x 16 B of
    // bias or 4 x 16 B of statistics per lane there and initialised 128 accumulators from them).
    constexpr bool kFuseStatsDecl = LNFOLD && OPATH == 2 && MIW == 3;     // see kFuseStats below
    constexpr int NTL = LNFOLD ? 10 : 2;
    float tl[NTL] = {};                  // [0..1] bias | c (hi == 0) / d (hi == 1) of column 32 ni + lr; LNFOLD: [2..5] mean | std, [6..9] 1 / std
    // per-frame low-word correction (VitGemmArgs::corr; fp16-output LN-folded and wide-residual instances only)
    constexpr bool CORR = std::is_same<TO, _Float16>::value && std::is_same<TI, _Float16>::value && (MODE == 2 || MODE == 6);
    float tcq[2] = {0.f, 0.f};           // corr[f0 + hi][column 32 ni + lr]
    int corr_bnd = 0;                    // rows of the wave's tile that belong to frame f0 (the others: f0 + 1)
    int corr_par = 0;                    // parity of f0
    float rscale[4] = {1.f, 1.f, 1.f, 1.f};
    // Round 4: these loads are plain, compiler-VISIBLE loads.  Rounds 2-3 issued them from inline asm (hidden from hipcc's s_waitcnt bookkeeping, so
    // that no compiler wait would also cover the LDS-DMA pieces in flight); hipcc treats an asm load's destination as written when the statement
    // ends, and a build with packed-fp32 instructions spilled / reused such a destination before its data had landed -> wrong lanes, a memory
    // fault (profiles/r04_fault_audit.md).  Visible loads cannot do that: the compiler waits before it touches the value.  They cost nothing
    // because the tail operands are consumed (tail_pin) right behind the last K step's own vmcnt(0), before that step issues the next tile's
    // DMA pieces -- the compiler's wait finds the counter at zero; the "memory"-clobbering asm statements of the step keep the loads at this
    // program point.  NOT volatile: hipcc waits for every volatile load on the spot (-2 % / -3.5 %).  -DCFSAR_HIDDEN_TAIL_LOADS: the asm form (A/B).
    auto asm_load = [&](const float* ptr) __attribute__((always_inline)) -> float {
#ifndef CFSAR_HIDDEN_TAIL_LOADS
        return *ptr;
#else
        float v;
        asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
        return v;
#endif
    };
    auto tail_loads = [&](int m0_, int n0_) __attribute__((always_inline)) {
        int nb_ = n0_ + wn * 64;
        nb_ = nb_ + 64 <= p.N ? nb_ : p.N - 64;
        const float* cd = LNFOLD ? (hi ? p.bias : p.cvec) : p.bias;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) tl[ni] = asm_load(cd + nb_ + ni * 32 + lr);
        if constexpr (CORR) {
            if (p.corr != nullptr) {                                   // kernel-uniform
                const int r0 = m0_ + wm * WR;                          // first row of the wave's tile (wave-uniform)
                const int f0 = r0 / p.corr_tokens;
                corr_bnd = (f0 + 1) * p.corr_tokens - r0;
                corr_par = f0 & 1;
                // the k slot of the lanes with hi == h serves the frame of PARITY h among {f0, f0 + 1}: which slot a row's correction sits in
                // depends on the row alone, not on the tile it falls into (192- and 256-row tile forms give the same bits)
                int f = ((f0 & 1) == hi) ? f0 : f0 + 1;
                const int fl = (p.M - 1) / p.corr_tokens;
                f = f < fl ? f : fl;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) tcq[ni] = asm_load(p.corr + (size_t)f * p.N + nb_ + ni * 32 + lr);
            }
        }
        if constexpr (LNFOLD) {
            if (!(kFuseStatsDecl && p.part != nullptr))
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi) {
                int m = m0_ + wm * WR + mi * 32 + lr;
                m = m < p.M ? m : p.M - 1;
                tl[2 + mi] = asm_load(p.rowstats + (size_t)m * 4 + hi);
                tl[6 + mi] = asm_load(p.rowstats + (size_t)m * 4 + 2);
            }
        }
    };
    // after the wait that covers tail_loads: the values are defined from here on (the compiler must not have copied them earlier)
    auto tail_pin = [&]() __attribute__((always_inline)) {
        if constexpr (CORR) asm volatile("" : "+v"(tcq[0]), "+v"(tcq[1]));
        if constexpr (LNFOLD)
            asm volatile("" : "+v"(tl[0]), "+v"(tl[1]), "+v"(tl[2]), "+v"(tl[3]), "+v"(tl[4]), "+v"(tl[5]), "+v"(tl[6]), "+v"(tl[7]), "+v"(tl[8]), "+v"(tl[NTL - 1]));
        else
            asm volatile("" : "+v"(tl[0]), "+v"(tl[1]));
    };
    // acc += bias (x) 1   |   acc += d (x) std - c (x) mean : the k slots 0..2 of the lanes with hi == 0 (k = 0..2) and, LN-folded, of the
    // lanes with hi == 1 (k = 8..10) carry the factors split into three bf16 (24 bits) or two fp16 (22 bits) parts.
    auto tail_fold = [&]() __attribute__((always_inline)) {
        if constexpr (!LNFOLD) {
            typedef typename Vec2B<TI>::v8 TI8;
            TI8 bw[2];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const float bv = hi ? 0.f : tl[ni];
                const TI h = (TI)bv;
                const float r1 = bv - (float)h;
                const TI m = (TI)r1;
                const TI l = (TI)(r1 - (float)m);
                TI q = (TI)0.f;
                if constexpr (CORR) q = (TI)tcq[ni];                    // k slot 3: corr[f0 + hi][n] x [row in frame f0 + hi]
                bw[ni] = TI8{h, m, l, q, 0, 0, 0, 0};
            }
            const TI one = (TI)(hi ? 0.f : 1.f);
            TI8 ones = TI8{one, one, one, 0, 0, 0, 0, 0};
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi) {
                if constexpr (CORR) {
                    const bool in0 = mi * 32 + lr < corr_bnd;
                    const TI ind = (TI)((p.corr != nullptr && ((in0 ? corr_par : corr_par ^ 1) == hi)) ? 1.f : 0.f);
                    ones = TI8{one, one, one, ind, 0, 0, 0, 0};
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    if constexpr (std::is_same<TI, _Float16>::value) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[ni], ones, acc[mi][ni], 0, 0, 0);
                    else acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[ni], ones, acc[mi][ni], 0, 0, 0);
                }
            }
        } else {
            // k slots 0..2 of the lanes with hi == 0 carry (c_hi, c_hi, c_lo) x (-mean_hi, -mean_lo, -mean_hi), the same slots of the
            // lanes with hi == 1 (k = 8..10) carry (d_hi, d_hi, d_lo) x (std_hi, std_lo, std_hi): every other k slot is zero.
            f16x8 cw[2], mx[4];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const _Float16 h = (_Float16)tl[ni], l = (_Float16)(tl[ni] - (float)h);
                _Float16 q = (_Float16)0.f;
                if constexpr (CORR) q = (_Float16)tcq[ni];             // k slot 3: corr[f0 + hi][n] x std(row) [row in frame f0 + hi]
                cw[ni] = f16x8{h, h, l, q, 0, 0, 0, 0};
            }
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi) {
                const float nm = hi ? tl[2 + mi] : -tl[2 + mi];
                const _Float16 h = (_Float16)nm, l = (_Float16)(nm - (float)h);
                _Float16 sdh = (_Float16)0.f;
                if constexpr (CORR) {
                    // the accumulator is divided by std in the epilogue: the correction (normalised units) enters multiplied by it
                    const bool in0 = mi * 32 + lr < corr_bnd;
                    // (corr_raw: the correction is already in raw-stream units -- xbar W_lo^T, the row's 1 / std applies to it like to everything else)
                    sdh = (_Float16)((p.corr != nullptr && ((in0 ? corr_par : corr_par ^ 1) == hi)) ? (p.corr_raw ? 1.0f : __builtin_amdgcn_rcpf(tl[6 + mi])) : 0.f);
                }
                mx[mi] = f16x8{h, l, h, sdh, 0, 0, 0, 0};
                rscale[mi] = tl[6 + mi];
            }
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cw[ni], mx[mi], acc[mi][ni], 0, 0, 0);
        }
    };
    // ---- statistics finalized in here (p.part != NULL; LN-folded early-DMA instances, K >= 512): during K step mi the lane fetches its half
    // (hi) of row 32 mi + lr's partials -- part_slots / 4 loads of 16 bytes, compiler-invisible like tail_loads, covered by the next
    // step's vmcnt(0) -- and during step mi + 1 sums them, adds the partner lane's half and leaves (mean | std, 1 / std) in tl[] exactly as
    // tail_loads would have loaded them from cfsar_ln_stats_finalize's output.
    // 192-row instances only: in the 256-row ones the 8 + 16 extra live registers cost 75-98 spills (and the finalize launch is 0.4 % of a
    // 16-episode step; it is 8 % of a one-episode step, which is where the 192-row tiles run)
    constexpr bool kFuseStats = LNFOLD && OPATH == 2 && MIW == 3;
    const bool fuse_stats = kFuseStats && p.part != nullptr;                  // kernel-uniform
    // Arithmetic = cfsar_ln_stats_finalize's, bit for bit (ln_stats_finalize8_kernel: pair sums s_q = slot 2q + slot 2q + 1, then
    // ((s0 + s1) + (s2 + s3)) + ((s6 + s7) + (s4 + s5))): the lanes with hi == 0 take slots 0-7, those with hi == 1 slots 8-15 (8-11 and
    // zeros for 12 slots) -- an episode's logits do not depend on which instance served it (tests: batch-size invariance).
    u32x4 sp[4] = {};
    auto stat_loads = [&](auto MI_) __attribute__((always_inline)) {
        constexpr int mi = decltype(MI_)::value;
        int m = m0 + wm * WR + mi * 32 + lr;
        m = m < p.M ? m : p.M - 1;
        const char* src = reinterpret_cast<const char*>(p.part) + ((size_t)m * p.part_slots + (hi ? 8 : 0)) * 8;
        sp[2] = u32x4{0u, 0u, 0u, 0u};
        sp[3] = u32x4{0u, 0u, 0u, 0u};
#ifndef CFSAR_HIDDEN_TAIL_LOADS
        sp[0] = *reinterpret_cast<const u32x4*>(src);
        sp[1] = *reinterpret_cast<const u32x4*>(src + 16);
        if (p.part_slots == 16 || hi == 0) {
            sp[2] = *reinterpret_cast<const u32x4*>(src + 32);
            sp[3] = *reinterpret_cast<const u32x4*>(src + 48);
        }
#else
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sp[0]) : "v"(src) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(sp[1]) : "v"(src) : "memory");
        if (p.part_slots == 16 || hi == 0) {
            asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "+v"(sp[2]) : "v"(src) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "+v"(sp[3]) : "v"(src) : "memory");
        }
#endif
    };
    auto stat_consume = [&](auto MI_) __attribute__((always_inline)) {
        constexpr int mi = decltype(MI_)::value;
        asm volatile("" : "+v"(sp[0]), "+v"(sp[1]), "+v"(sp[2]), "+v"(sp[3]));      // defined from here on (the wait of this step covered them)
        auto f = [](unsigned u) { return __builtin_bit_cast(float, u); };
        float s = ((f(sp[0][0]) + f(sp[0][2])) + (f(sp[1][0]) + f(sp[1][2]))) + ((f(sp[2][0]) + f(sp[2][2])) + (f(sp[3][0]) + f(sp[3][2])));
        float sq = ((f(sp[0][1]) + f(sp[0][3])) + (f(sp[1][1]) + f(sp[1][3]))) + ((f(sp[2][1]) + f(sp[2][3])) + (f(sp[3][1]) + f(sp[3][3])));
        const int partner = (lane ^ 32) << 2;
        s += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner, __builtin_bit_cast(int, s)));
        sq += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner, __builtin_bit_cast(int, sq)));
        const float invD = p.part_invD, eps = p.part_eps;
        const float mean = s * invD;
        const float var = fmaxf(sq * invD - mean * mean, 0.f);
        const float sd = sqrtf(var + eps);
        tl[2 + mi] = hi ? sd : mean;
        tl[6 + mi] = 1.0f / sd;
    };
    // K step `kstat` of an output tile: consume row tile kstat - 1, fetch row tile kstat (uniform branches, no MFMA inside)
    auto stats_at = [&](int kstat) __attribute__((always_inline)) {
        if constexpr (kFuseStats) {
            if (fuse_stats && kstat >= 0 && kstat <= MIW) {
                static_for<MIW>([&](auto MI_) {
                    if (kstat == decltype(MI_)::value + 1) stat_consume(MI_);
                });
                static_for<MIW>([&](auto MI_) {
                    if (kstat == decltype(MI_)::value) stat_loads(MI_);
                });
            }
        }
    };
    u32x4 GX[4], GW[4];
    uint4 xfA[4], wfA[2], xfB[4], wfB[2];
    auto gloadX = [&](const unsigned (&ox)[4], int kt, auto J) __attribute__((always_inline)) {
        GX[decltype(J)::value] = *reinterpret_cast<const u32x4*>(p.A + (size_t)(kt >= p.nka ? kt - p.nka : kt) * akstride + ox[decltype(J)::value]);
    };
    auto gloadW = [&](const unsigned (&ow)[4], int kt, auto J) __attribute__((always_inline)) {
        GW[decltype(J)::value] = *reinterpret_cast<const u32x4*>(p.W + (size_t)kt * ROWB + ow[decltype(J)::value]);
    };
    auto swriteX = [&](int stage, auto J) __attribute__((always_inline)) {
        *reinterpret_cast<u32x4*>(smem + stage * STAGE + wr_off + decltype(J)::value * 8192) = GX[decltype(J)::value];
    };
    auto swriteW = [&](int stage, auto J) __attribute__((always_inline)) {
        *reinterpret_cast<u32x4*>(smem + stage * STAGE + TM * ROWB + wr_off + decltype(J)::value * 8192) = GW[decltype(J)::value];
    };
    auto dmaX = [&](const unsigned (&ox)[4], int kt, int stage, auto J) __attribute__((always_inline)) {
        const char* src = p.A + (size_t)(kt >= p.nka ? kt - p.nka : kt) * akstride + ox[decltype(J)::value];      // split weights: A's K tiles repeat
        const unsigned dst = ldsw + (unsigned)stage * (unsigned)STAGE + (unsigned)decltype(J)::value * 8192u;
#ifdef CFSAR_DEV
        if (p.dbg & 4096) { glds16_asm_pol<1>(src, dst); return; }       // A/B: streaming operand loaded nt / sc1
        if (p.dbg & 16384) { glds16_asm_pol<2>(src, dst); return; }
#endif
        glds16_asm(src, dst);
    };
    auto dmaW = [&](const unsigned (&ow)[4], int kt, int stage, auto J) __attribute__((always_inline)) {
        const char* src = p.W + (size_t)kt * ROWB + ow[decltype(J)::value];
        const unsigned dst = ldsw + (unsigned)stage * (unsigned)STAGE + (unsigned)(TM * ROWB) + (unsigned)decltype(J)::value * 8192u;
#ifdef CFSAR_DEV
        if (p.dbg & 8192) { glds16_asm_pol<1>(src, dst); return; }
#endif
        glds16_asm(src, dst);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    // read order = order of first use by the MFMA sequence (ni-major): x0 w0 x1 x2 x3 w1
    auto load_one = [&](int stage, int ss, auto J, uint4 (&xf)[4], uint4 (&wf)[2]) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        const char* base = smem + stage * STAGE;
        const int x2 = ss << 5;
        constexpr int isx[6] = {1, 0, 1, 1, MIW == 4 ? 1 : 0, 0};       // MIW = 3: x0 w0 x1 x2 w1
        constexpr int idx[6] = {0, 0, 1, 2, MIW == 4 ? 3 : 1, 1};
        if constexpr (isx[j] != 0) xf[idx[j]] = *reinterpret_cast<const uint4*>(base + (rdX[idx[j]] ^ x2));
        else wf[idx[j]] = *reinterpret_cast<const uint4*>(base + (rdW[idx[j]] ^ x2));
    };
    auto mfma_one = [&](auto J, uint4 (&xf)[4], uint4 (&wf)[2], auto ZERO) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        constexpr int ni = j / MIW, mi = j % MIW;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const f32x16 c = decltype(ZERO)::value ? zero : acc[mi][ni];          // first sub-step of an output tile: C = 0 (inline constant)
        if constexpr (std::is_same<TI, _Float16>::value)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[ni]), __builtin_bit_cast(f16x8, xf[mi]), c, 0, 0, 0);
        else
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[ni]), __builtin_bit_cast(bf16x8, xf[mi]), c, 0, 0, 0);
    };
    // One 128-byte K tile = 4 sub-steps of 8 MFMAs; `cur` / `nxt` = LDS stages of this K tile / the following one.
    //   LOAD : operands of a later K tile are fetched: (ox, ow, ksrc) name them (register path: two K tiles ahead -> VGPRs;
    //          DMA path: one K tile ahead -> stage nxt)
    //   WRITE: (register path) the VGPRs hold the following K tile: write it to stage nxt
    //   SYNC : barrier after MFMA 1 of sub-step 3 (the following K tile is complete in LDS for every wave)
    //   FRAGS: prefetch the first fragments of the following K tile after the barrier (off at an output-tile boundary: the
    //          epilogue runs in between)
    auto step = [&](int cur, int nxt, const unsigned (&ox)[4], const unsigned (&ow)[4], int ksrc, auto LOAD, auto WRITE, auto SYNC, auto FRAGS, auto ZERO, auto TAIL, int kstat = -1) __attribute__((always_inline)) {
        constexpr bool load = decltype(LOAD)::value, write = decltype(WRITE)::value, sync = decltype(SYNC)::value,
                       frags = decltype(FRAGS)::value, tail = decltype(TAIL)::value;
        static_for<NM>([&](auto J) {                                    // sub-step 0
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfA, wfA, ZERO);
            if constexpr (j < NL) load_one(cur, 1, J, xfB, wfB);
            if constexpr (j >= NM - MIW) {                              // the MIW pieces of X
                if constexpr (OPATH == 0) { if constexpr (write) swriteX(nxt, std::integral_constant<int, j - (NM - MIW)>{}); }
                else if constexpr (OPATH == 1) { if constexpr (load) dmaX(ox, ksrc, nxt, std::integral_constant<int, j - (NM - MIW)>{}); }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<NM>([&](auto J) {                                    // sub-step 1
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfB, wfB, F_{});
            if constexpr (j < NL) load_one(cur, 2, J, xfA, wfA);
            if constexpr (OPATH == 0) {
                if constexpr (j >= NM - MIW && load) gloadX(ox, ksrc, std::integral_constant<int, j - (NM - MIW)>{});
            } else if constexpr (OPATH == 1) {
                if constexpr (j >= NM - 4 && load) dmaW(ow, ksrc, nxt, std::integral_constant<int, j - (NM - 4)>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<NM>([&](auto J) {                                    // sub-step 2
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfA, wfA, F_{});
            if constexpr (j < NL) load_one(cur, 3, J, xfB, wfB);
            if constexpr (j >= NM - 4 && OPATH == 0 && write) swriteW(nxt, std::integral_constant<int, j - (NM - 4)>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<NM>([&](auto J) {                                    // sub-step 3
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfB, wfB, F_{});
            if constexpr (j < 2) {
                if constexpr (OPATH == 0 && load) {
                    gloadW(ow, ksrc, std::integral_constant<int, 2 * j>{});
                    gloadW(ow, ksrc, std::integral_constant<int, 2 * j + 1>{});
                }
            } else if constexpr (frags) {
                load_one(nxt, 0, std::integral_constant<int, j - 2>{}, xfA, wfA);
                if constexpr (j == NM - 1 && NL > NM - 2) load_one(nxt, 0, std::integral_constant<int, NL - 1>{}, xfA, wfA);   // MIW = 3: 5 loads, 4 slots
            }
            // OPATH 2: every wave has issued its last fragment reads of stage `cur` before the barrier of this sub-step -> the
            // stage is free: K tile `ksrc` (two ahead) starts its flight NOW and has a whole K step to land (OPATH 1 issues the
            // same pieces 1.5 - 2.5 sub-steps later, into the other stage)
            if constexpr (tail && j == 2) tail_loads(m0, n0);          // this tile's bias / statistics: covered by the NEXT step's wait
            if constexpr (kFuseStats && j == 2) stats_at(kstat);       // behind this step's wait and barrier
            if constexpr (OPATH == 2 && load && j >= 2 && MIW == 4) {
                if constexpr (j < 6) dmaX(ox, ksrc, cur, std::integral_constant<int, j - 2>{});
                else {
                    dmaW(ow, ksrc, cur, std::integral_constant<int, 2 * (j - 6)>{});
                    dmaW(ow, ksrc, cur, std::integral_constant<int, 2 * (j - 6) + 1>{});
                }
            }
            if constexpr (OPATH == 2 && load && j >= 2 && MIW == 3) {   // 3 + 4 pieces over the slots j = 2 .. 5: X0 X1 | X2 W0 | W1 W2 | W3
                if constexpr (j == 2) { dmaX(ox, ksrc, cur, std::integral_constant<int, 0>{}); dmaX(ox, ksrc, cur, std::integral_constant<int, 1>{}); }
                if constexpr (j == 3) { dmaX(ox, ksrc, cur, std::integral_constant<int, 2>{}); dmaW(ow, ksrc, cur, std::integral_constant<int, 0>{}); }
                if constexpr (j == 4) { dmaW(ow, ksrc, cur, std::integral_constant<int, 1>{}); dmaW(ow, ksrc, cur, std::integral_constant<int, 2>{}); }
                if constexpr (j == 5) { dmaW(ow, ksrc, cur, std::integral_constant<int, 3>{}); }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j == 1 && sync) {
                if constexpr (OPATH >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed
#ifndef CFSAR_HIDDEN_TAIL_LOADS
                // last K step of the tile: the tail operands (loaded one step earlier) are consumed HERE, right behind the wait that covers
                // them and before this step issues the next tile's DMA pieces -- the compiler's own wait for them finds the counter at zero
                if constexpr (!frags && OPATH >= 1) tail_pin();
#endif
                __syncthreads();                                        // (register path: hipcc adds lgkmcnt(0) for the ds_writes)
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    origin(b, m0, n0);
    unsigned offX[4], offW[4];
    offsets(m0, n0, offX, offW);
    // ---- pipeline fill for the first output tile of this workgroup
    if constexpr (OPATH == 0) {
        static_for<MIW>([&](auto J) { gloadX(offX, 0, J); });
        static_for<4>([&](auto J) { gloadW(offW, 0, J); });
        static_for<MIW>([&](auto J) { swriteX(0, J); });
        static_for<4>([&](auto J) { swriteW(0, J); });
        static_for<MIW>([&](auto J) { gloadX(offX, 1, J); });
        static_for<4>([&](auto J) { gloadW(offW, 1, J); });
    } else {
        static_for<MIW>([&](auto J) { dmaX(offX, 0, 0, J); });
        static_for<4>([&](auto J) { dmaW(offW, 0, 0, J); });
        if constexpr (OPATH == 2) {                       // K tile 1 is in flight before the first step as well
            static_for<MIW>([&](auto J) { dmaX(offX, 1, 1, J); });
            static_for<4>([&](auto J) { dmaW(offW, 1, 1, J); });
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(MIW + 4) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
#ifdef CFSAR_DEV
    if (p.dbg & 32768) {                          // A/B: static priority for the younger half of the workgroup (cdna guide T5, static form)
        if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    }
    if (p.dbg & 65536) {                          // A/B: ... for the older half
        if (wave < 4) __builtin_amdgcn_s_setprio(1);
    }
#endif
    int sb = 0;                                   // stage that holds K tile 0 of the current output tile
    char* slab = smem + EPI_OFF + wave * EPI_SLAB;
#ifdef CFSAR_DEV
    int trace_i = 0;
    if (p.dbg & 128) {
        const int n = ((blockIdx.x >> 3) & 31) * p.stagger_unit;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
    }
#define CFSAR_TRACE(slot) do { if (p.trace && tid == 0 && trace_i < 64) p.trace[((size_t)blockIdx.x * 64 + trace_i) * 4 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CFSAR_TRACE(slot) do { } while (0)
#endif
    for (;;) {
        const int bn = b + grid;
        const bool has_next = bn < nt;
        int m0n = m0, n0n = n0;
        if (has_next) origin(bn, m0n, n0n);
        static_for<NL>([&](auto J) { load_one(sb, 0, J, xfA, wfA); });
        CFSAR_TRACE(0);
        int kt = 0;
        // The tail steps ALWAYS prefetch (one straight-line MFMA stream: a fork on has_next would merge two 128-register
        // accumulator sets through phi copies).  After the last output tile of this workgroup the "next" origin is the current
        // one, so the surplus loads re-read valid memory and the surplus LDS writes land in the free stage.
        u32x4 rv0[4] = {};
        // (LDS-DMA instance only: in the register-staged one 16 more live registers across the last K step cost 22-30 spills and 10 %)
        auto residual_prefetch = [&]() __attribute__((always_inline)) {     // rows rr + 8 it of the wave's first 32-row pass
            if constexpr (HAS_RES && !WIDE && (OPATH >= kPreMinOpath && OPATH <= kPreMaxOpath)) {
                const int mb_ = m0 + wm * WR, nb_ = n0 + wn * 64;
                const int ncl_ = nb_ + 64 <= p.N ? nb_ : p.N - 64;
                const int rr_ = lane >> 3, Q_ = lane & 7;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    int row = mb_ + rr_ + it * 8;
                    row = row < p.M ? row : p.M - 1;                        // clamped: out-of-range rows are never stored
                    rv0[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.res) + ((size_t)row * p.ldr + ncl_ + 8 * Q_) * 2);
                }
            }
        };
        // nk >= 3 (launcher; nk = 2 is the SHORTK instance).  Step 0 starts the accumulators (C = 0), step nk - 2 fetches the tail operands, then the bias /
        // LayerNorm terms go in by one more MFMA per 32x32 tile.
        if constexpr (OPATH == 0) {
            step(sb & 1, (sb + 1) & 1, offX, offW, 2, T_{}, T_{}, T_{}, T_{}, T_{}, F_{});
            for (kt = 1; kt < nk - 2; ++kt) step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, kt + 2, T_{}, T_{}, T_{}, T_{}, F_{}, F_{});
            offsets(m0n, n0n, offX, offW);              // this tile's remaining K tiles are already in registers / LDS
            step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, 0, T_{}, T_{}, T_{}, T_{}, F_{}, T_{});      // loads K tile 0 of the next tile
            ++kt;
            step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, 1, T_{}, T_{}, T_{}, F_{}, F_{}, F_{});      // writes it, loads K tile 1
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(MIW + 4) : "memory");     // the tail operands (older than the loads of K tile 1) have landed
        } else if constexpr (OPATH == 1 && SHORTK) {
            step(sb & 1, (sb + 1) & 1, offX, offW, 1, T_{}, F_{}, T_{}, T_{}, T_{}, T_{});
            kt = 1;
            offsets(m0n, n0n, offX, offW);
            residual_prefetch();
            step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, 0, T_{}, F_{}, T_{}, F_{}, F_{}, F_{});      // K tile 0 of the next tile
        } else if constexpr (OPATH == 1) {
            step(sb & 1, (sb + 1) & 1, offX, offW, 1, T_{}, F_{}, T_{}, T_{}, T_{}, F_{});
            for (kt = 1; kt < nk - 2; ++kt) step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, kt + 1, T_{}, F_{}, T_{}, T_{}, F_{}, F_{});
            step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, kt + 1, T_{}, F_{}, T_{}, T_{}, F_{}, T_{});
            ++kt;
            offsets(m0n, n0n, offX, offW);
            residual_prefetch();
            step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, 0, T_{}, F_{}, T_{}, F_{}, F_{}, F_{});      // K tile 0 of the next tile
        } else {
            // step kt waits for K tile kt + 1 (issued by step kt - 1) and issues K tile kt + 2 into its own stage after its barrier
            step(sb & 1, (sb + 1) & 1, offX, offW, 2, T_{}, F_{}, T_{}, T_{}, T_{}, F_{}, 0);
            for (kt = 1; kt < nk - 2; ++kt) step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, kt + 2, T_{}, F_{}, T_{}, T_{}, F_{}, F_{}, kt);
            offsets(m0n, n0n, offX, offW);
            step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, 0, T_{}, F_{}, T_{}, T_{}, F_{}, T_{});      // K tile 0 of the next tile
            ++kt;
            residual_prefetch();
            step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, 1, T_{}, F_{}, T_{}, F_{}, F_{}, F_{});      // K tile 1 of the next tile: in flight
        }                                                                                                  // through the epilogue, BEFORE its stores
        tail_pin();
        tail_fold();
        CFSAR_TRACE(1);
#ifdef CFSAR_DEV
        if (p.dbg & 4) {                                                // ablation: no epilogue (keep the accumulators live)
            if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.out)[0] = acc[1][1][3] + acc[MIW - 1][1][2] + acc[2][0][1];
        } else
#endif
        {
            const int mb = m0 + wm * WR, nb = n0 + wn * 64;
            if constexpr (WIDE) {
                if (mb + WR <= p.M && nb + 64 <= p.N) epilogue_rows_wide<STORE, true, MIW>(acc, p, mb, nb, lane, slab);
                else epilogue_rows_wide<STORE, false, MIW>(acc, p, mb, nb, lane, slab);
            } else
            if (mb + WR <= p.M && nb + 64 <= p.N) epilogue_rows<TO, ACT, HAS_RES, STORE, true, LNFOLD, HAS_RES && (OPATH >= kPreMinOpath && OPATH <= kPreMaxOpath), HB, MIW>(acc, p, mb, nb, lane, slab, rscale, rv0);
            else epilogue_rows<TO, ACT, HAS_RES, STORE, false, LNFOLD, HAS_RES && (OPATH >= kPreMinOpath && OPATH <= kPreMaxOpath), HB, MIW>(acc, p, mb, nb, lane, slab, rscale, rv0);
        }
        CFSAR_TRACE(2);
#ifdef CFSAR_DEV
        ++trace_i;
#endif
        if (!has_next) {
            if constexpr (OPATH == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus DMA of the last step targets this workgroup's LDS
            break;
        }
        sb = (sb + nk) & 1;
        b = bn;
        m0 = m0n;
        n0 = n0n;
    }
}

int persistent_grid() {
    const int n = cfsar_num_cus() & ~7;         // one workgroup per CU; the block -> XCD walk assumes a multiple of 8
    return n >= 8 ? n : 8;
}

// the 192-row form exists for the product policy's instances only (compile time): LN-folded on the early-DMA path with write-through
// stores, residual on the early-DMA and the register-staged path with plain stores
constexpr bool vit_has_192(int mode, int opath, int store) {
    return ((mode == 1 || mode == 6) && store == 0 && (opath == 0 || opath == 2)) || ((mode == 2 || mode == 4) && opath == 2 && store == 2);
}

int persistent_grid();
int vit_pick_miw(int M, int tiles_n, int mode, int opath, int store, int K, int dbg) {
    if (K == 128 || !vit_has_192(mode, opath, store)) return 4;
    const long long G = persistent_grid();
    const long long t256 = (long long)((M + 255) / 256) * tiles_n, t192 = (long long)((M + 191) / 192) * tiles_n;
    const long long r256 = ((t256 + G - 1) / G) * 256, r192 = ((t192 + G - 1) / G) * 192;
    int miw = r192 * 10 <= r256 * 9 ? 3 : 4;
#ifdef CFSAR_DEV
    if (dbg & (1 << 17)) miw = 3;          // forced (tests / A/B)
    if (dbg & (1 << 18)) miw = 4;
#endif
    (void)dbg;
    return miw;
}

template <typename TI, typename TO, int ACT, int MODE, int OPATH, int STORE, bool SHORTK, int MIW = 4>
int launch_inst2(const VitGemmArgs& a, hipStream_t s) {
    auto* fn = &vit_gemm_kernel<TI, TO, ACT, MODE, OPATH, STORE, SHORTK, MIW>;
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(fn), LDS_BYTES, "cfsar_gemm(vit)")) return rc;
    const int grid = a.ntiles < persistent_grid() ? ((a.ntiles + 7) & ~7) : persistent_grid();
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), LDS_BYTES, s, a);
    return cfsar_check_launch("cfsar_gemm(vit)");
}

template <typename TI, typename TO, int ACT, int MODE, int OPATH, int STORE>
int launch_inst(const VitGemmArgs& a, hipStream_t s) {
    if (a.K == 128) return launch_inst2<TI, TO, ACT, MODE, 1, STORE, true>(a, s);       // two K tiles: its own instance
    if constexpr (vit_has_192(MODE, OPATH, STORE)) {
        if (a.miw == 3) return launch_inst2<TI, TO, ACT, MODE, OPATH, STORE, false, 3>(a, s);
    }
    return launch_inst2<TI, TO, ACT, MODE, OPATH, STORE, false>(a, s);
}

// mode: 0 bias -> bf16, 1 residual -> fp16 in place, 2 LN-folded (fp16 operands) -> bf16; f16io (the fp16 numerics mode): fp16
// operands in mode 1, fp16 output in mode 2
template <int OPATH, int STORE>
int launch_path(const VitGemmArgs& a, int mode, bool f16io, hipStream_t s) {
    if (mode == 5) return launch_inst<__bf16, __bf16, CFSAR_ACT_NONE, 1, OPATH, STORE>(a, s);     // bf16 residual (+ ReLU): RN50 conv3
    if (mode == 6) return launch_inst<_Float16, _Float16, CFSAR_ACT_NONE, 6, OPATH, STORE>(a, s);  // wide residual (fp16 numerics mode)
    if (mode == 1) {
        if (f16io) return launch_inst<_Float16, _Float16, CFSAR_ACT_NONE, 1, OPATH, STORE>(a, s);
        return launch_inst<__bf16, _Float16, CFSAR_ACT_NONE, 1, OPATH, STORE>(a, s);
    }
    if (mode == 2) {
        if (f16io) {
            if (a.hb_tokens > 0) return -2;
            if (a.act == CFSAR_ACT_QUICKGELU) return launch_inst<_Float16, _Float16, CFSAR_ACT_QUICKGELU, 2, OPATH, STORE>(a, s);
            return launch_inst<_Float16, _Float16, CFSAR_ACT_NONE, 2, OPATH, STORE>(a, s);
        }
        if (a.act == CFSAR_ACT_QUICKGELU) return launch_inst<_Float16, __bf16, CFSAR_ACT_QUICKGELU, 2, OPATH, STORE>(a, s);
        if (a.hb_tokens > 0) return launch_inst<_Float16, __bf16, CFSAR_ACT_NONE, 4, OPATH, STORE>(a, s);
        return launch_inst<_Float16, __bf16, CFSAR_ACT_NONE, 2, OPATH, STORE>(a, s);
    }
#ifdef CFSAR_DEV
    if (mode == 3) return launch_inst<_Float16, __bf16, CFSAR_ACT_NONE, 0, OPATH, STORE>(a, s);   // timing A/B: fp16 MFMA on a plain GEMM
#endif
    if (a.act == CFSAR_ACT_QUICKGELU) return launch_inst<__bf16, __bf16, CFSAR_ACT_QUICKGELU, 0, OPATH, STORE>(a, s);
    if (a.relu) return launch_inst<__bf16, __bf16, kActRelu, 0, OPATH, STORE>(a, s);
    return launch_inst<__bf16, __bf16, CFSAR_ACT_NONE, 0, OPATH, STORE>(a, s);
}

}  // namespace

#ifdef CFSAR_DEV
static int g_stagger_unit = 0;
static long long* g_trace = nullptr;
#endif
// Returns -2 when the call is outside this kernel's contract (the caller falls back to the generic kernels of gemm.hip).
int cfsar_gemm_vit_try(const VitGemmCall& c, hipStream_t s) {
    const bool lnfold = c.rowstats != nullptr || c.part != nullptr;
    const bool f16io = lnfold ? c.out_dtype == CFSAR_F16 : c.in_dtype == CFSAR_F16;     // the fp16 numerics mode (see launch_path)
    const bool f16res = !lnfold && c.out_dtype == CFSAR_F16 && c.res && c.res_dtype == CFSAR_F16 && c.act == CFSAR_ACT_NONE;
    const bool bf16plain = (c.out_dtype == CFSAR_BF16 || (lnfold && c.out_dtype == CFSAR_F16)) && !c.res &&
                           (c.act == CFSAR_ACT_NONE || c.act == CFSAR_ACT_QUICKGELU);
    // bf16 activations + bf16 residual (a separate buffer) [+ ReLU] -> bf16: the conv3 + identity of the RN50 bottlenecks (short K: what
    // this kernel's tile-to-tile operand pipeline is for)
    const bool bf16res = !lnfold && c.in_dtype == CFSAR_BF16 && c.out_dtype == CFSAR_BF16 && c.res && c.res_dtype == CFSAR_BF16 &&
                         c.act == CFSAR_ACT_NONE && c.res != c.out;
    if (!lnfold && c.in_dtype == CFSAR_F16 && !f16res) return -2;
    // relu(A W^T + bias), N >= 256 (a 64- or 128-wide conv1 would waste most of the 256-wide tile: those stay on the 256 x 128 / x 64 kernels)
    const bool plainrelu = bf16plain && !lnfold && c.relu && c.act == CFSAR_ACT_NONE && c.out_dtype == CFSAR_BF16 && c.N >= 256;
    if (!(f16res || bf16plain || bf16res) || !c.bias || (c.relu && !(bf16res || plainrelu))) return -2;
    if (lnfold && (!bf16plain || !c.cvec)) return -2;
    if (c.stats_out && !f16res) return -2;
    if (c.K % 64 != 0 || c.K < 128 || c.N % 64 != 0 || c.ldo % 8 != 0 || (c.res && c.ldr % 8 != 0)) return -2;
    if ((size_t)c.M * c.lda * 2 >= (1ull << 32) || (size_t)c.N * c.ldw * 2 >= (1ull << 32) ||
        (size_t)c.M * c.ldo * 2 >= (1ull << 32)) return -2;                                                 // 32-bit byte offsets
    VitGemmArgs a;
    a.A = static_cast<const char*>(c.A);
    a.W = static_cast<const char*>(c.W);
    a.out = c.out;
    a.bias = c.bias;
    a.res = c.res;
    a.rowstats = c.rowstats;
    const int ka = c.ka > 0 ? c.ka : c.K;           // K of A (split weights: K = 2 ka)
    // K = 2 ka: split weights [w_hi | w_lo] against A walked twice; 2 K = 3 ka (round 6): A = [a_hi | a_lo] against [w_hi | w_hi | w_lo] -- the K tiles
    // behind ka wrap to A's first tiles again (kt -> kt - nka): the third segment reads a_hi
    if (!(ka == c.K || (((2 * ka == c.K && ka % 64 == 0) || (3 * ka == 2 * c.K && c.wide && ka % 128 == 0)) && ka >= 128 && c.ha_tokens == 0 &&
                        c.hb_tokens == 0 && c.in_dtype == CFSAR_F16)))
        return cfsar_fail("cfsar_gemm (vit): split operands need K = 2 ka (ka %% 64 == 0) or 2 K = 3 ka (ka %% 128 == 0), fp16 operands, row-major layouts (K=%d, ka=%d)", c.K, ka);
    if (c.wide && !(f16res && c.in_dtype == CFSAR_F16 && c.res == c.out && c.ldr == c.ldo && c.ha_tokens == 0))
        return cfsar_fail("cfsar_gemm_residual_wide: fp16 operands, the residual stream updated in place");
    a.nka = ka / 64;
    a.res_lo = c.wide ? c.res_lo : nullptr;
    a.corr = c.corr;
    a.corr_tokens = c.corr_tokens;
    a.corr_raw = c.corr_raw;
    a.colsum = c.colsum;
    if (c.colsum && !(lnfold && c.act == CFSAR_ACT_QUICKGELU && c.out_dtype == CFSAR_F16 && c.corr_tokens >= 128 && c.hb_tokens == 0))
        return cfsar_fail("cfsar_gemm_lnfold_hp: per-frame output means exist for the fp16 QuickGELU form with >= 128 tokens per frame");
    if (c.corr && !(c.corr_tokens >= 128 && c.in_dtype == CFSAR_F16 && c.out_dtype == CFSAR_F16 && (lnfold || c.wide) && c.hb_tokens == 0))
        return cfsar_fail("cfsar_gemm (vit): the per-frame correction needs >= 128 tokens per frame and an fp16-mode instance (tokens=%d)", c.corr_tokens);
    a.part = c.part; a.part_slots = c.part_slots; a.part_invD = 1.0f / (float)ka; a.part_eps = c.part_eps;
    if (c.part && !(c.opath == 2 && ka >= 512 && (c.part_slots == 12 || c.part_slots == 16) && c.part_slots * 64 == ka))
        return cfsar_fail("cfsar_gemm_lnfold_partials: needs K = 64 slots in {768, 1024} (K=%d, slots=%d)", ka, c.part_slots);
    a.cvec = c.cvec;
    a.stats_out = c.stats_out;
    a.stats_slots = c.N / 64;
    a.M = c.M; a.N = c.N; a.K = c.K;
    a.lda = c.lda; a.ldw = c.ldw; a.ldo = c.ldo; a.ldr = c.ldr;
    a.act = c.act;
    a.relu = c.relu;
    a.tiles_n = (c.N + TN - 1) / TN;
    // Tile height: 192 rows when that saves a tenth of the rounds-x-rows the persistent grid walks (one or two episodes per call:
    // 62 bands x 9 columns of 256-row tiles are 2.2 rounds on 256 CUs and cost 3; 83 x 9 of 192 rows cost 3 x 0.75)
    a.miw = vit_pick_miw(c.M, a.tiles_n, lnfold ? 2 : ((f16res || bf16res) ? (c.wide ? 6 : 1) : 0), c.opath, c.store, c.K, c.dbg);
    if (c.part && a.miw != 3) return cfsar_fail("cfsar_gemm_lnfold_partials: internal: fused statistics need the 192-row instance");
    if (c.out_miw) *c.out_miw = a.miw;
    a.ntiles = ((c.M + 64 * a.miw - 1) / (64 * a.miw)) * a.tiles_n;
    a.group = c.group > 0 ? c.group : 8;
    a.colfast = c.colfast;
    a.hb_tokens = c.hb_tokens; a.hb_heads = c.hb_heads; a.ha_tokens = c.ha_tokens;
    if (a.hb_tokens > 0 && !(lnfold && c.act == CFSAR_ACT_NONE && a.hb_tokens >= 128 && a.hb_heads > 0 && c.N == 3 * 64 * a.hb_heads &&
                             c.M % a.hb_tokens == 0))
        return cfsar_fail("cfsar_gemm_lnfold: head-blocked output needs act NONE, N = 192 heads, tokens >= 128, M a multiple of tokens");
    if (a.ha_tokens > 0 && !(f16res && a.ha_tokens >= 8 && c.M % a.ha_tokens == 0))
        return cfsar_fail("cfsar_gemm_residual_stats: head-blocked A needs M a multiple of tokens");
#ifdef CFSAR_DEV
    a.dbg = c.dbg;
    if (g_cfsar_walk_enable) a.dbg |= ((g_cfsar_walk_phase++ & 1) << 24);      // experiment: alternate the walk direction launch by launch
    a.stagger_unit = g_stagger_unit;
    a.trace = g_trace;
#endif
    int mode = lnfold ? 2 : (f16res ? (c.wide ? 6 : 1) : (bf16res ? 5 : 0));
#ifdef CFSAR_DEV
    if (mode == 0 && c.act == CFSAR_ACT_NONE && (c.dbg & 64)) mode = 3;
#endif
    int opath = c.opath;
#ifdef CFSAR_DEV
    // Developer library only: the two alternative forms measured in round 5 (profiles/r05_gemm_forms.md; both lose against this kernel).
    // Calls they do not cover fall back to the 8-wave kernel.
    if (opath == 4 || opath == 5) {                 // 4 = two 4-wave workgroups per CU (gemm_vit4.hip), 5 = one wave per SIMD (gemm_vit1w.hip)
        const int rc = opath == 4 ? cfsar_gemm_vit4_launch(a, mode, f16io, c.store, s) : cfsar_gemm_vit1w_launch(a, mode, f16io, c.store, s);
        if (rc != -2) {
            if (c.out_miw) *c.out_miw = opath == 4 ? 3 : 4;
            return rc;
        }
        opath = c.K <= 1024 ? 2 : 0;
    }
#endif
    switch (opath * 4 + c.store) {
        case 0: return launch_path<0, 0>(a, mode, f16io, s);
        case 2: return launch_path<0, 2>(a, mode, f16io, s);
        case 4: return launch_path<1, 0>(a, mode, f16io, s);
        case 6: return launch_path<1, 2>(a, mode, f16io, s);
        case 8: return launch_path<2, 0>(a, mode, f16io, s);
        case 10: return launch_path<2, 2>(a, mode, f16io, s);
#ifdef CFSAR_DEV
        case 1: return launch_path<0, 1>(a, mode, f16io, s);
        case 5: return launch_path<1, 1>(a, mode, f16io, s);
        case 11: return launch_path<2, 3>(a, mode, f16io, s);     // store-policy A/B on the early-DMA path: variants 31 (sc1 nt),
        case 12: return launch_path<2, 4>(a, mode, f16io, s);     // 32 (sc0 sc1), 33 (sc0 sc1 nt), 34 (sc0)
        case 13: return launch_path<2, 5>(a, mode, f16io, s);
        case 14: return launch_path<2, 6>(a, mode, f16io, s);
#endif
        default: return -2;
    }
}

namespace {
#ifdef CFSAR_DEV
int g_force_opath = -1, g_force_store = -1, g_force_dbg = 0;
#endif
// Operand path (same-box A/B, profiles/r03_gemm_anatomy.md and round 5 below): the LN-folded short-K launches -- QKV, c_fc -- take the LDS-DMA
// path with the pieces issued right behind the previous step's barrier (2); the residual launches of the one-word stream (out_proj, c_proj) and the
// long-K c_proj of the two-word stream the register-staged path (0).
// kind: 0 = LN-folded launch, 1 = residual launch, 2 = wide residual launch of the fp16 mode (developer builds: ablation bits 21 / 22 keep the product policy for the LN-folded / the residual
// launches, so a forced form can be A/B'd on one kind of launch alone)
int vit_policy_opath(int K, int kind = -1) {
#ifdef CFSAR_DEV
    const bool keep = (kind == 0 && (g_force_dbg & (1 << 21))) || ((kind == 1 || kind == 2) && (g_force_dbg & (1 << 22)));
    if (keep) { }
    else if (g_force_opath >= 10) { if (K <= 1024) return g_force_opath - 10; }     // 10 + path: short-K launches only
    else if (g_force_opath >= 0) return g_force_opath;
#endif
    // (Round 5 measured the register-staged path on the short-K launches again: out_proj +0.3 %, the bf16 mode's QKV / c_fc +0.35 ... +0.66 % in the
    // bench legs of the DEVELOPER library -- and +0.1 % / -1.3 % with the PRODUCT library, previous build against new build in separate processes:
    // the two builds allocate registers differently (product: 21 spilled registers in the register-staged LN-folded instance, 0 in the LDS-DMA one;
    // developer: 8 and 0; the developer DMA residual instance spills 4 where the product's spills none).  The rule by K stays; an operand-path A/B is
    // only valid between product builds.  profiles/r05_lnfold_path_ab.log, r05_outproj_path_ab.log, r05_bench_outproj_path_ab.txt, r05_bench_lnfold_path_ab.txt)
    (void)kind;
    return K <= 1024 ? 2 : 0;
}
// Tile walk of the residual launches inside an XCD's range (tile_of): the long-K one (c_proj: three or four column tiles per row band, a weight matrix
// larger than the L2) takes its tiles column-fastest in groups of 16 bands -- 2.4-3 % faster than the band-fastest groups of 8 that the short-K
// launches keep (same-box A/B at 16 and 36 episodes, profiles/r05_forms_s31_colfast.log; out_proj, QKV and c_fc lose 1-3 % with it).  The short-K
// launches (QKV, c_fc, out_proj) walk band-fastest in groups of SIX bands: against groups of 8, c_fc -1.6 ... -2.2 %, out_proj 0 ... -2.4 %, QKV
// 0 ... -0.9 % at 16 / 36 episodes, ViT-L/14 shapes 0 ... -1 % (groups of 2 / 3 / 4 / 16 / 32 are worse somewhere; profiles/r05_forms_s33_groups.log).
// The walk changes which workgroup computes a tile, never a value.
int vit_policy_colfast(int K) { return K > 1024 ? 1 : 0; }
int vit_policy_group(int K) { return K > 1024 ? 16 : 6; }
int vit_policy_store(int dflt) {
#ifdef CFSAR_DEV
    // (dflt names the kind of launch: 2 = LN-folded, 0 = residual; ablation bits 21 / 22 keep the product policy for that kind)
    if (g_force_store >= 0 && !((dflt == 2 && (g_force_dbg & (1 << 21))) || (dflt == 0 && (g_force_dbg & (1 << 22))))) return g_force_store;
#endif
    return dflt;
}
}
int cfsar_vit_policy_opath(int K) { return vit_policy_opath(K); }      // the cfsar_gemm dispatcher (gemm.hip) uses the same policy
#ifdef CFSAR_DEV
#include "../../include/clipfsar_hip_dev.h"
// dev builds only: operand path / store policy of cfsar_gemm_lnfold and cfsar_gemm_residual_stats; -1 = product policy
extern "C" void cfsar_debug_set_vit_paths(int opath, int store) { g_force_opath = opath; g_force_store = store; }
extern "C" void cfsar_debug_set_vit_dbg(int dbg) { g_force_dbg = dbg & ~(1 << 25); g_cfsar_walk_enable = (dbg >> 25) & 1; g_cfsar_walk_phase = 0; }
// trace buffer ([grid][64][4] long long, device memory; NULL = off) and stagger unit (x 64 cycles) for dbg bit 128
extern "C" void cfsar_debug_set_vit_trace(void* trace, int stagger_unit) { g_trace = static_cast<long long*>(trace); g_stagger_unit = stagger_unit; }
#endif

namespace {
// per-frame token means of the GEMM's OUTPUT from the wave tiles' per-frame column sums (VitGemmArgs::colsum, int32 fixed point): frame f
// covers rows [f T, (f + 1) T) = the wave tiles b0 .. b1 of `wr` rows; tile b contributes its slot f - (b wr) / T (0 or 1).
__global__ __launch_bounds__(256) void frame_means_from_colsums_kernel(const int* __restrict__ cs, __bf16* __restrict__ out, int wr,
                                                                       int tokens, int N) {
    const int f = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int r0 = f * tokens, r1 = r0 + tokens - 1;
    long long s = 0;                                        // fixed point, 2^-12 units: exact, order-free
    for (int b = r0 / wr; b <= r1 / wr; ++b) {
        const int j = f - (b * wr) / tokens;
        if (j == 0 || j == 1) s += cs[((size_t)b * 2 + j) * N + n];
    }
    out[(size_t)f * N + n] = (__bf16)((float)s * (1.0f / 4096.0f) / (float)tokens);
}
}  // namespace

// out = act(LayerNorm(x; gamma, beta) W^T + bias) with the LayerNorm folded into the GEMM (MODE 2 above).  See the header.
static int gemm_lnfold_impl(const void* x, const void* Wg, void* out, const float* cvec, const float* dvec,
                            const float* rowstats, int M, int N, int K, int lda, int ldw, int ldo, int act, int out_dtype, int hb_tokens,
                            int hb_heads, cfsar_stream_t stream, const float* partial = nullptr, int slots = 0, float eps = 0.f, int wsplit = 0,
                            const float* corr = nullptr, int corr_tokens = 0, void* colmean_out = nullptr, void* colsum_ws = nullptr) {
    CFSAR_REQUIRE(out_dtype == CFSAR_BF16 || out_dtype == CFSAR_F16, "cfsar_gemm_lnfold: out_dtype must be bf16 or fp16, got %d", out_dtype);
    CFSAR_REQUIRE(x && Wg && out && cvec && dvec && (rowstats || partial), "cfsar_gemm_lnfold: null pointer");
    CFSAR_REQUIRE(M > 0 && N > 0 && K >= 128 && K % 64 == 0 && N % 64 == 0, "cfsar_gemm_lnfold: bad shape M=%d N=%d K=%d (K %% 64, N %% 64, K >= 128)", M, N, K);
    CFSAR_REQUIRE(lda >= K && ldw >= (wsplit ? 2 * K : K) && ldo >= N && lda % 8 == 0 && ldw % 8 == 0 && ldo % 8 == 0, "cfsar_gemm_lnfold: bad leading dimension");
    CFSAR_REQUIRE(act == CFSAR_ACT_NONE || act == CFSAR_ACT_QUICKGELU, "cfsar_gemm_lnfold: bad act %d", act);
    const int ka_ = K;
    if (wsplit) K = 2 * K;                            // split weights [N, 2 ka] = [hi | lo]: see VitGemmArgs::nka
    VitGemmCall c;
    c.ka = ka_;
    c.corr_raw = corr_tokens < 0;                     // cfsar_gemm_lnfold_hp: corr_tokens < 0 selects the raw-stream form
    if (corr_tokens < 0) corr_tokens = -corr_tokens;
    c.corr = corr; c.corr_tokens = corr_tokens;
    int miw_used = 4;
    c.colsum = colmean_out ? colsum_ws : nullptr; c.out_miw = &miw_used;
    c.A = x; c.W = Wg; c.out = out; c.bias = dvec; c.res = nullptr; c.rowstats = partial ? nullptr : rowstats; c.cvec = cvec; c.stats_out = nullptr;
    c.part = partial; c.part_slots = slots; c.part_eps = eps;
    c.M = M; c.N = N; c.K = K; c.lda = lda; c.ldw = ldw; c.ldo = ldo; c.ldr = 0;
    c.out_dtype = out_dtype; c.in_dtype = CFSAR_F16; c.res_dtype = CFSAR_F32; c.act = act; c.relu = 0;
    // The policies go by the OPERAND's K: a split-weight launch of K = 768 walks 1 536 but keeps the short-K launches' LDS-DMA path (no spilled registers;
    // the register-staged LN-folded instances spill 29-35) and band-fastest walk.  Product-build A/B in separate processes, round 6 (profiles/r06_split_policy.txt):
    // strict mode with split QKV 234.8 -> 239.1 episodes/s, split QKV + c_fc 202.6 -> 210.5, split all 191.5 -> 198.1.  -DCFSAR_SPLIT_POLICY_BY_WALK: by the walked K.
#ifdef CFSAR_SPLIT_POLICY_BY_WALK
    const int Kp = K;
#else
    const int Kp = ka_;
#endif
    c.opath = vit_policy_opath(Kp, 0); c.store = vit_policy_store(2); c.group = vit_policy_group(Kp); c.colfast = vit_policy_colfast(Kp); c.dbg = 0;
    c.hb_tokens = hb_tokens; c.hb_heads = hb_heads; c.ha_tokens = 0;
#ifdef CFSAR_DEV
    c.dbg = g_force_dbg;
#endif
    const int rc = cfsar_gemm_vit_try(c, static_cast<hipStream_t>(stream));
    if (rc == 0 && colmean_out) {
        const int frames = (M + corr_tokens - 1) / corr_tokens;
        hipLaunchKernelGGL(frame_means_from_colsums_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)frames), dim3(256), 0,
                           static_cast<hipStream_t>(stream), static_cast<const int*>(colsum_ws), static_cast<__bf16*>(colmean_out),
                           32 * miw_used, corr_tokens, N);
        return cfsar_check_launch("cfsar_gemm_lnfold_hp(frame means)");
    }
    return rc == -2 ? cfsar_fail("cfsar_gemm_lnfold: operands too large for 32-bit offsets") : rc;
}

extern "C" int cfsar_gemm_lnfold(const void* x, const void* Wg, void* out, const float* cvec, const float* dvec,
                                 const float* rowstats, int M, int N, int K, int lda, int ldw, int ldo, int act, int out_dtype,
                                 cfsar_stream_t stream) {
    return gemm_lnfold_impl(x, Wg, out, cvec, dvec, rowstats, M, N, K, lda, ldw, ldo, act, out_dtype, 0, 0, stream);
}

// The QKV form with head-blocked output: out[((f heads + h) tokens + t) * 192 + 64 which + c] (see the header).
extern "C" int cfsar_gemm_lnfold_heads(const void* x, const void* Wg, void* out, const float* cvec, const float* dvec,
                                       const float* rowstats, int M, int N, int K, int lda, int ldw, int tokens, int heads,
                                       cfsar_stream_t stream) {
    CFSAR_REQUIRE(tokens >= 128 && heads > 0 && N == 192 * heads && M % tokens == 0,
                  "cfsar_gemm_lnfold_heads: needs tokens >= 128, N = 192 heads, M a multiple of tokens (M=%d N=%d tokens=%d heads=%d)", M, N,
                  tokens, heads);
    return gemm_lnfold_impl(x, Wg, out, cvec, dvec, rowstats, M, N, K, lda, ldw, N, CFSAR_ACT_NONE, CFSAR_BF16, tokens, heads, stream);
}

// cfsar_gemm_lnfold / cfsar_gemm_lnfold_heads (tokens > 0) with the row statistics taken straight from the producer's partials
// (cfsar_gemm_residual_stats: [M, slots, 2]) and finalized inside the kernel: no cfsar_ln_stats_finalize launch in between.  K = 64 slots
// in {768, 1024} (the ViT-B / ViT-L widths); other widths keep the two-launch form.  See the header.
static int lnfold_partials_impl(const void* x, const void* Wg, void* out, const float* cvec, const float* dvec,
                                const float* partial, int slots, float eps, float* rowstats_ws, int M, int N, int K, int lda,
                                int ldw, int ldo, int act, int out_dtype, int tokens, int heads, int wsplit, cfsar_stream_t stream,
                                const float* corr = nullptr, int corr_tokens = 0, void* colmean_out = nullptr, void* colsum_ws = nullptr) {
    CFSAR_REQUIRE(partial != nullptr && rowstats_ws != nullptr, "cfsar_gemm_lnfold_partials: null partials / workspace");
    CFSAR_REQUIRE(slots > 0 && slots * 64 == K, "cfsar_gemm_lnfold_partials: K=%d is not 64 x slots=%d", K, slots);
    int dbg = 0;
#ifdef CFSAR_DEV
    dbg = g_force_dbg;
#endif
    // The 192-row instances (small M: one or two episodes per call) finalize the statistics themselves; at batch scale the 256-row
    // instances have no registers to spare for it and the finalize launch is 0.4 % of the step: two launches from here.
    const int Kt = wsplit ? 2 * K : K;
    const bool fused = vit_policy_opath(Kt, 0) == 2 && (slots == 12 || slots == 16) && K >= 512 &&
                       vit_pick_miw(M, (N + TN - 1) / TN, 2, 2, vit_policy_store(2), Kt, dbg) == 3;
    if (!fused) {
        if (int rc = cfsar_ln_stats_finalize(partial, rowstats_ws, M, slots, K, eps, stream)) return rc;
        if (tokens > 0) return cfsar_gemm_lnfold_heads(x, Wg, out, cvec, dvec, rowstats_ws, M, N, K, lda, ldw, tokens, heads, stream);
        return gemm_lnfold_impl(x, Wg, out, cvec, dvec, rowstats_ws, M, N, K, lda, ldw, ldo, act, out_dtype, 0, 0, stream, nullptr, 0, 0.f, wsplit, corr, corr_tokens,
                                colmean_out, colsum_ws);
    }
    if (tokens > 0) {
        CFSAR_REQUIRE(tokens >= 128 && heads > 0 && N == 192 * heads && M % tokens == 0 && act == CFSAR_ACT_NONE && out_dtype == CFSAR_BF16,
                      "cfsar_gemm_lnfold_partials: head-blocked output needs tokens >= 128, N = 192 heads, M a multiple of tokens, act NONE, bf16");
        return gemm_lnfold_impl(x, Wg, out, cvec, dvec, nullptr, M, N, K, lda, ldw, N, act, out_dtype, tokens, heads, stream, partial, slots, eps);
    }
    return gemm_lnfold_impl(x, Wg, out, cvec, dvec, nullptr, M, N, K, lda, ldw, ldo, act, out_dtype, 0, 0, stream, partial, slots, eps, wsplit, corr, corr_tokens,
                            colmean_out, colsum_ws);
}

extern "C" int cfsar_gemm_lnfold_partials(const void* x, const void* Wg, void* out, const float* cvec, const float* dvec,
                                          const float* partial, int slots, float eps, float* rowstats_ws, int M, int N, int K, int lda,
                                          int ldw, int ldo, int act, int out_dtype, int tokens, int heads, cfsar_stream_t stream) {
    return lnfold_partials_impl(x, Wg, out, cvec, dvec, partial, slots, eps, rowstats_ws, M, N, K, lda, ldw, ldo, act, out_dtype, tokens, heads, 0, stream);
}

// The fp16 numerics mode's LN-folded GEMM: wsplit = 1: SPLIT weights Wg [N, 2 K] = [fp16(W gamma) | fp16(W gamma - fp16(W gamma))], the kernel
// walks x's K tiles twice in one fp32 accumulation chain; corr != NULL: per-frame low-word correction [ceil(M / corr_tokens), N] (see the header).  Statistics: rowstats [M, 4] (partial == NULL) or the producer's
// partials [M, slots, 2] finalized here (rowstats_ws [M, 4] is then the workspace of the two-launch form).
extern "C" int cfsar_gemm_lnfold_hp(const void* x, const void* Wg, void* out, const float* cvec, const float* dvec, const float* rowstats,
                                    const float* partial, int slots, float eps, float* rowstats_ws, int M, int N, int K, int lda, int ldw,
                                    int ldo, int act, int out_dtype, int wsplit, const float* corr, int corr_tokens, void* colmean_out,
                                    void* colsum_ws, cfsar_stream_t stream) {
    CFSAR_REQUIRE(out_dtype == CFSAR_F16, "cfsar_gemm_lnfold_hp: fp16 output only (the fp16 numerics mode)");
    CFSAR_REQUIRE((corr == nullptr && colmean_out == nullptr) || corr_tokens >= 128 || corr_tokens <= -128, "cfsar_gemm_lnfold_hp: the per-frame forms need >= 128 tokens per frame");
    CFSAR_REQUIRE(corr_tokens >= 0 || corr != nullptr, "cfsar_gemm_lnfold_hp: corr_tokens < 0 (raw-stream form of the correction) needs corr");
    CFSAR_REQUIRE(colmean_out == nullptr || (colsum_ws != nullptr && act == CFSAR_ACT_QUICKGELU), "cfsar_gemm_lnfold_hp: output means need the workspace and act = QUICKGELU");
    if (partial != nullptr) {
        CFSAR_REQUIRE(slots > 0 && slots * 64 == K, "cfsar_gemm_lnfold_hp: K=%d is not 64 x slots=%d", K, slots);
        return lnfold_partials_impl(x, Wg, out, cvec, dvec, partial, slots, eps, rowstats_ws, M, N, K, lda, ldw, ldo, act, out_dtype, 0, 0, wsplit, stream,
                                    corr, corr_tokens, colmean_out, colsum_ws);
    }
    return gemm_lnfold_impl(x, Wg, out, cvec, dvec, rowstats, M, N, K, lda, ldw, ldo, act, out_dtype, 0, 0, stream, nullptr, 0, 0.f, wsplit, corr, corr_tokens,
                            colmean_out, colsum_ws);
}

// x = x + A W^T + bias (fp16 residual stream, in place) and, if stats_partial != NULL, the per-row partial LayerNorm
// statistics of the NEW x: stats_partial[m][n / 64] = (sum, sum of squares) over columns [64 (n/64), +64).  See the header.
static int gemm_residual_stats_impl(const void* A, const void* W, void* x, const float* bias, float* stats_partial, int M,
                                    int N, int K, int lda, int ldw, int ldx, int in_dtype, int ha_tokens, cfsar_stream_t stream) {
    CFSAR_REQUIRE(in_dtype == CFSAR_BF16 || in_dtype == CFSAR_F16, "cfsar_gemm_residual_stats: in_dtype must be bf16 or fp16, got %d", in_dtype);
    CFSAR_REQUIRE(A && W && x && bias, "cfsar_gemm_residual_stats: null pointer");
    CFSAR_REQUIRE(M > 0 && N > 0 && K >= 128 && K % 64 == 0 && N % 64 == 0, "cfsar_gemm_residual_stats: bad shape M=%d N=%d K=%d", M, N, K);
    CFSAR_REQUIRE(lda >= K && ldw >= K && ldx >= N && lda % 8 == 0 && ldw % 8 == 0 && ldx % 8 == 0, "cfsar_gemm_residual_stats: bad leading dimension");
    VitGemmCall c;
    c.A = A; c.W = W; c.out = x; c.bias = bias; c.res = x; c.rowstats = nullptr; c.cvec = nullptr; c.stats_out = stats_partial;
    c.part = nullptr; c.part_slots = 0; c.part_eps = 0.f;
    c.M = M; c.N = N; c.K = K; c.lda = lda; c.ldw = ldw; c.ldo = ldx; c.ldr = ldx;
    c.out_dtype = CFSAR_F16; c.in_dtype = in_dtype; c.res_dtype = CFSAR_F16; c.act = CFSAR_ACT_NONE; c.relu = 0;
    c.opath = vit_policy_opath(K, 1); c.store = vit_policy_store(0); c.group = vit_policy_group(K); c.colfast = vit_policy_colfast(K); c.dbg = 0;
    c.hb_tokens = 0; c.hb_heads = 0; c.ha_tokens = ha_tokens;
#ifdef CFSAR_DEV
    c.dbg = g_force_dbg & ((1 << 17) | (1 << 18));       // tile-height overrides only
#endif
    const int rc = cfsar_gemm_vit_try(c, static_cast<hipStream_t>(stream));
    return rc == -2 ? cfsar_fail("cfsar_gemm_residual_stats: operands too large for 32-bit offsets") : rc;
}

extern "C" int cfsar_gemm_residual_stats(const void* A, const void* W, void* x, const float* bias, float* stats_partial, int M,
                                         int N, int K, int lda, int ldw, int ldx, int in_dtype, cfsar_stream_t stream) {
    return gemm_residual_stats_impl(A, W, x, bias, stats_partial, M, N, K, lda, ldw, ldx, in_dtype, 0, stream);
}

// The fp16 numerics mode's residual GEMM: x = x + A W^T + bias with the add in fp32 and ONE rounding (MODE 6); x_lo != NULL: two-word
// stream (x_hi + x_lo); wsplit: W is [N, 2 K] = [hi | lo].  See the header.
extern "C" int cfsar_gemm_residual_wide(const void* A, const void* W, void* x_hi, void* x_lo, const float* bias, float* stats_partial, int M,
                                        int N, int K, int wsplit, int lda, int ldw, int ldx, const float* corr, int corr_tokens,
                                        cfsar_stream_t stream) {
    CFSAR_REQUIRE(corr == nullptr || corr_tokens >= 128, "cfsar_gemm_residual_wide: the per-frame correction needs >= 128 tokens per frame");
    CFSAR_REQUIRE(A && W && x_hi && bias, "cfsar_gemm_residual_wide: null pointer");
    CFSAR_REQUIRE(M > 0 && N > 0 && K >= 128 && K % 64 == 0 && N % 64 == 0, "cfsar_gemm_residual_wide: bad shape M=%d N=%d K=%d", M, N, K);
    CFSAR_REQUIRE(wsplit >= 0 && wsplit <= 2, "cfsar_gemm_residual_wide: wsplit must be 0, 1 or 2, got %d", wsplit);
    CFSAR_REQUIRE(wsplit != 2 || corr == nullptr, "cfsar_gemm_residual_wide: wsplit = 2 carries the weights' second word itself (no per-frame correction)");
    const int Kt = wsplit == 2 ? 3 * K : (wsplit ? 2 * K : K);
    const int Ka = wsplit == 2 ? 2 * K : K;                  // columns of A: [a_hi | a_lo] in the two-word form
    CFSAR_REQUIRE(lda >= Ka && ldw >= Kt && ldx >= N && lda % 8 == 0 && ldw % 8 == 0 && ldx % 8 == 0, "cfsar_gemm_residual_wide: bad leading dimension");
    VitGemmCall c;
    c.A = A; c.W = W; c.out = x_hi; c.bias = bias; c.res = x_hi; c.rowstats = nullptr; c.cvec = nullptr; c.stats_out = stats_partial;
    c.part = nullptr; c.part_slots = 0; c.part_eps = 0.f;
    c.M = M; c.N = N; c.K = Kt; c.lda = lda; c.ldw = ldw; c.ldo = ldx; c.ldr = ldx;
    c.out_dtype = CFSAR_F16; c.in_dtype = CFSAR_F16; c.res_dtype = CFSAR_F16; c.act = CFSAR_ACT_NONE; c.relu = 0;
#ifdef CFSAR_SPLIT_POLICY_BY_WALK
    const int Kp = Kt;
#else
    const int Kp = K;                // (see gemm_lnfold_impl: the policies go by the operand's K)
#endif
    c.opath = vit_policy_opath(Kp, 2); c.store = vit_policy_store(0); c.group = vit_policy_group(Kp); c.colfast = vit_policy_colfast(Kp); c.dbg = 0;
    c.hb_tokens = 0; c.hb_heads = 0; c.ha_tokens = 0;
    c.ka = Ka; c.wide = 1; c.res_lo = x_lo; c.corr = corr; c.corr_tokens = corr_tokens;
#ifdef CFSAR_DEV
    c.dbg = g_force_dbg & ((1 << 17) | (1 << 18));
#endif
    const int rc = cfsar_gemm_vit_try(c, static_cast<hipStream_t>(stream));
    return rc == -2 ? cfsar_fail("cfsar_gemm_residual_wide: operands too large for 32-bit offsets") : rc;
}

// The out_proj form: A is the attention output in head-blocked layout A[((f heads + h) tokens + t) * 64 + c], heads = K / 64.
extern "C" int cfsar_gemm_residual_stats_heads(const void* A, const void* W, void* x, const float* bias, float* stats_partial, int M,
                                               int N, int K, int ldw, int ldx, int tokens, cfsar_stream_t stream) {
    CFSAR_REQUIRE(tokens >= 8 && M % tokens == 0, "cfsar_gemm_residual_stats_heads: M=%d is not a multiple of tokens=%d", M, tokens);
    return gemm_residual_stats_impl(A, W, x, bias, stats_partial, M, N, K, K, ldw, ldx, CFSAR_BF16, tokens, stream);
}
