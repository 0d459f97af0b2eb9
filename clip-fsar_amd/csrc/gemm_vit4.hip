// The ViT-block GEMMs of gemm_vit.hip as TWO 4-wave workgroups per CU (round 5; VERDICT r4 item 1, docs/history/r04.md "Open (1)").
//
// Why: in the 8-wave kernel both waves of a SIMD enter the epilogue of the same output tile together -- the matrix pipe idles for the
// 2 us (QKV) ... 6.5 us (c_fc, QuickGELU) the conversion / LDS transpose / store issue takes, and all eight waves stand at ONE barrier per
// K tile.  Here a CU holds two independent 256-thread workgroups (one wave per SIMD each): while one is in its epilogue or waits at its
// barrier, the other one's MFMAs own the matrix pipes.
//
// Geometry: 192 x 128 output tile (2 x 2 waves, 96 x 64 wave tiles = 96 accumulators), 128-byte K tiles, two 40 KiB LDS stages per
// workgroup (X rows [0, 24 KiB), W rows [24 KiB, 40 KiB)) = 80 KiB = exactly half a CU's LDS.  No room is left for epilogue slabs, so the
// epilogue's 32-row transposes run INSIDE the stage the tile's last K step has just retired: a wave's four 1 KiB slab chunks are the LDS-DMA
// destinations of its OWN first four X pieces, i.e. nothing another wave reads or writes -- the next tile's K tile 0 is already in the other
// stage (issued one step earlier, as in gemm_vit.hip), K tile 1 is issued by each wave right behind its epilogue into the chunks it has
// just finished with, without a barrier.  What the aliasing costs is the depth of that one K tile's flight (one K step instead of two at
// each tile start); the other workgroup of the CU covers it.
//
// Operand path: LDS-DMA only (gemm_vit.hip's OPATH 2: the pieces of K tile kt + 2 are issued right behind the barrier of step kt into
// the stage that barrier retired).  Modes, tail MFMA (bias / LayerNorm terms / per-frame correction) and epilogues are gemm_vit.hip's
// (gemm_vit_epi.h); head-blocked layouts, the K = 128 form and the in-kernel statistics finalization are not carried over.
#include "gemm_vit_epi.h"

namespace {

constexpr int NW4 = 4;                         // waves per workgroup
constexpr int TN4 = 128;                       // output tile columns

template <typename TI, typename TO, int ACT, int MODE, int STORE, int MIW>
__global__ __launch_bounds__(256, 2) void vit_gemm4_kernel(VitGemmArgs p) {
    constexpr int TMv = 64 * MIW;                 // output tile rows (192)
    constexpr int WR = 32 * MIW;                  // rows of one wave
    constexpr int NM = 2 * MIW;                   // MFMAs per sub-step
    constexpr int NL = MIW + 2;                   // fragment loads per sub-step
    constexpr int XB = TMv * ROWB, WB = TN4 * ROWB, STG = XB + WB;      // 24 KiB + 16 KiB
    constexpr int PX = TMv / (8 * NW4), PW = TN4 / (8 * NW4);          // LDS-DMA pieces (8 rows x 128 B) per wave per K tile: 6 + 4
    constexpr int CH = NW4 * 1024;                // distance of a wave's consecutive pieces = of its slab chunks
    static_assert(MIW == 3 && PX == 6 && PW == 4 && NM == 6, "the DMA slot schedule below is written for 192 x 128 tiles");
    constexpr bool WIDE = MODE == 6;
    constexpr bool HAS_RES = MODE == 1 || WIDE;
    constexpr bool LNFOLD = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, hi = lane >> 5;

    // ---- tile walk (gemm_vit.hip's): virtual block ids b, b + grid, ...; id -> XCD-contiguous linear index -> (band, column)
    const int nt = p.ntiles, grid = (int)gridDim.x;
    const int xq = nt >> 3, xr = nt & 7;
    const int tiles_m = nt / p.tiles_n;
    auto origin = [&](int b, int& m0, int& n0) __attribute__((always_inline)) {
        const int xcd = b & 7;
        const int lin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
        int tm, tn;
        tile_of(lin, tiles_m, p.tiles_n, p.group, p.colfast, tm, tn);
        m0 = tm * TMv;
        n0 = tn * TN4;
#ifdef CFSAR_DEV
        if (p.dbg & 8) { m0 = 0; n0 = 0; }     // ablation: every workgroup reads tile (0, 0) (cache-hot operands)
#endif
    };
    // per-lane source offsets of the wave's staging pieces: piece i covers tile rows (4 i + wave) * 8 .. + 7, 128 B each, 16-byte chunks
    // XOR-swizzled by the row (the LDS image is lane-linear: the swizzle sits on the SOURCE address and on the fragment reads)
    auto offsets = [&](int m0, int n0, unsigned (&ox)[PX], unsigned (&ow)[PW]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            const int row = (i * NW4 + wave) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ swz(row);
            int gm = m0 + row;
            gm = gm < p.M ? gm : p.M - 1;
            ox[i] = (unsigned)gm * (unsigned)p.lda * 2u + chunk * 16;
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int row = (i * NW4 + wave) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ swz(row);
            int gn = n0 + row;
            gn = gn < p.N ? gn : p.N - 1;
            ow[i] = (unsigned)gn * (unsigned)p.ldw * 2u + chunk * 16;
        }
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    int rdX[MIW], rdW[2];
#pragma unroll
    for (int i = 0; i < MIW; ++i) {
        const int rx = wm * WR + i * 32 + lr;
        rdX[i] = rx * ROWB + ((hi ^ swz(rx)) << 4);                        // sub-step ss: ^ (ss << 5)
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rw = wn * 64 + i * 32 + lr;
        rdW[i] = XB + rw * ROWB + ((hi ^ swz(rw)) << 4);
    }

    const int nk = p.K / 64;
    int b = blockIdx.x;
    if (b >= nt) return;
    int m0 = 0, n0 = 0;
    f32x16 acc[MIW][2];
    // ---- tail operands (gemm_vit.hip: bias / LayerNorm terms / per-frame correction enter by ONE extra MFMA per 32 x 32 tile after the last K step)
    constexpr int NTL = LNFOLD ? 10 : 2;
    float tl[NTL] = {};
    constexpr bool CORR = std::is_same<TO, _Float16>::value && std::is_same<TI, _Float16>::value && (MODE == 2 || MODE == 6);
    float tcq[2] = {0.f, 0.f};
    int corr_bnd = 0, corr_par = 0;
    float rscale[4] = {1.f, 1.f, 1.f, 1.f};
    auto tail_loads = [&](int m0_, int n0_) __attribute__((always_inline)) {
        int nb_ = n0_ + wn * 64;
        nb_ = nb_ + 64 <= p.N ? nb_ : p.N - 64;
        const float* cd = LNFOLD ? (hi ? p.bias : p.cvec) : p.bias;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) tl[ni] = cd[nb_ + ni * 32 + lr];
        if constexpr (CORR) {
            if (p.corr != nullptr) {                                   // kernel-uniform
                const int r0 = m0_ + wm * WR;
                const int f0 = r0 / p.corr_tokens;
                corr_bnd = (f0 + 1) * p.corr_tokens - r0;
                corr_par = f0 & 1;
                int f = ((f0 & 1) == hi) ? f0 : f0 + 1;
                const int fl = (p.M - 1) / p.corr_tokens;
                f = f < fl ? f : fl;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) tcq[ni] = p.corr[(size_t)f * p.N + nb_ + ni * 32 + lr];
            }
        }
        if constexpr (LNFOLD) {
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi) {
                int m = m0_ + wm * WR + mi * 32 + lr;
                m = m < p.M ? m : p.M - 1;
                tl[2 + mi] = p.rowstats[(size_t)m * 4 + hi];
                tl[6 + mi] = p.rowstats[(size_t)m * 4 + 2];
            }
        }
    };
    auto tail_pin = [&]() __attribute__((always_inline)) {
        if constexpr (CORR) asm volatile("" : "+v"(tcq[0]), "+v"(tcq[1]));
        if constexpr (LNFOLD)
            asm volatile("" : "+v"(tl[0]), "+v"(tl[1]), "+v"(tl[2]), "+v"(tl[3]), "+v"(tl[4]), "+v"(tl[6]), "+v"(tl[7]), "+v"(tl[8]));
        else
            asm volatile("" : "+v"(tl[0]), "+v"(tl[1]));
    };
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
                if constexpr (CORR) q = (TI)tcq[ni];
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
            f16x8 cw[2], mx[MIW];
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const _Float16 h = (_Float16)tl[ni], l = (_Float16)(tl[ni] - (float)h);
                _Float16 q = (_Float16)0.f;
                if constexpr (CORR) q = (_Float16)tcq[ni];
                cw[ni] = f16x8{h, h, l, q, 0, 0, 0, 0};
            }
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi) {
                const float nm = hi ? tl[2 + mi] : -tl[2 + mi];
                const _Float16 h = (_Float16)nm, l = (_Float16)(nm - (float)h);
                _Float16 sdh = (_Float16)0.f;
                if constexpr (CORR) {
                    const bool in0 = mi * 32 + lr < corr_bnd;
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

    uint4 xfA[MIW], wfA[2], xfB[MIW], wfB[2];
    auto dmaX = [&](const unsigned (&ox)[PX], int kt, int stage, auto J) __attribute__((always_inline)) {
        const char* src = p.A + (size_t)(kt >= p.nka ? kt - p.nka : kt) * ROWB + ox[decltype(J)::value];      // split weights: A's K tiles repeat
        glds16_asm(src, ldsw + (unsigned)stage * (unsigned)STG + (unsigned)decltype(J)::value * (unsigned)CH);
    };
    auto dmaW = [&](const unsigned (&ow)[PW], int kt, int stage, auto J) __attribute__((always_inline)) {
        const char* src = p.W + (size_t)kt * ROWB + ow[decltype(J)::value];
        glds16_asm(src, ldsw + (unsigned)stage * (unsigned)STG + (unsigned)XB + (unsigned)decltype(J)::value * (unsigned)CH);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    // read order = order of first use by the MFMA sequence (ni-major): x0 w0 x1 x2 w1
    auto load_one = [&](int stage, int ss, auto J, uint4 (&xf)[MIW], uint4 (&wf)[2]) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        const char* base = smem + stage * STG;
        const int x2 = ss << 5;
        constexpr int isx[5] = {1, 0, 1, 1, 0};
        constexpr int idx[5] = {0, 0, 1, 2, 1};
        if constexpr (isx[j] != 0) xf[idx[j]] = *reinterpret_cast<const uint4*>(base + (rdX[idx[j]] ^ x2));
        else wf[idx[j]] = *reinterpret_cast<const uint4*>(base + (rdW[idx[j]] ^ x2));
    };
    auto mfma_one = [&](auto J, uint4 (&xf)[MIW], uint4 (&wf)[2], auto ZERO) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        constexpr int ni = j / MIW, mi = j % MIW;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const f32x16 c = decltype(ZERO)::value ? zero : acc[mi][ni];
        acc[mi][ni] = cfsar_mfma_32x32x16<TI>(wf[ni], xf[mi], c);
    };
    // One 128-byte K tile = 4 sub-steps of 6 MFMAs; `cur` / `nxt` = LDS stages of this K tile / the following one.
    //   LOAD : K tile `ksrc` of (ox, ow) is issued into stage `cur` behind this step's barrier (it is two K tiles ahead, or K tile 0 of the next
    //          output tile)
    //   FRAGS: prefetch the first fragments of the following K tile behind the barrier (off in an output tile's last step)
    //   TAIL : fetch this tile's tail operands behind the barrier (second-to-last step: the last step's wait covers them)
    //   LAST : last step of an output tile: the tail operands are pinned right behind its wait
    auto step = [&](int cur, int nxt, const unsigned (&ox)[PX], const unsigned (&ow)[PW], int ksrc, auto LOAD, auto FRAGS, auto ZERO, auto TAIL, auto LAST) __attribute__((always_inline)) {
        constexpr bool load = decltype(LOAD)::value, frags = decltype(FRAGS)::value, tail = decltype(TAIL)::value, last = decltype(LAST)::value;
        static_for<NM>([&](auto J) {                                    // sub-step 0
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfA, wfA, ZERO);
            if constexpr (j < NL) load_one(cur, 1, J, xfB, wfB);
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<NM>([&](auto J) {                                    // sub-step 1
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfB, wfB, F_{});
            if constexpr (j < NL) load_one(cur, 2, J, xfA, wfA);
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<NM>([&](auto J) {                                    // sub-step 2
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfA, wfA, F_{});
            if constexpr (j < NL) load_one(cur, 3, J, xfB, wfB);
            __builtin_amdgcn_sched_barrier(0);
        });
        static_for<NM>([&](auto J) {                                    // sub-step 3
            constexpr int j = decltype(J)::value;
            mfma_one(J, xfB, wfB, F_{});
            if constexpr (j >= 2 && frags) {
                load_one(nxt, 0, std::integral_constant<int, j - 2>{}, xfA, wfA);
                if constexpr (j == NM - 1) load_one(nxt, 0, std::integral_constant<int, NL - 1>{}, xfA, wfA);   // 5 loads, 4 slots
            }
            if constexpr (tail && j == 2) tail_loads(m0, n0);
            // every wave has issued (and, at the barrier, completed) its last fragment reads of stage `cur`: 6 + 4 pieces over the slots j = 2 .. 5
            if constexpr (load && j == 2) { dmaX(ox, ksrc, cur, std::integral_constant<int, 0>{}); dmaX(ox, ksrc, cur, std::integral_constant<int, 1>{}); dmaX(ox, ksrc, cur, std::integral_constant<int, 2>{}); }
            if constexpr (load && j == 3) { dmaX(ox, ksrc, cur, std::integral_constant<int, 3>{}); dmaX(ox, ksrc, cur, std::integral_constant<int, 4>{}); dmaX(ox, ksrc, cur, std::integral_constant<int, 5>{}); }
            if constexpr (load && j == 4) { dmaW(ow, ksrc, cur, std::integral_constant<int, 0>{}); dmaW(ow, ksrc, cur, std::integral_constant<int, 1>{}); }
            if constexpr (load && j == 5) { dmaW(ow, ksrc, cur, std::integral_constant<int, 2>{}); dmaW(ow, ksrc, cur, std::integral_constant<int, 3>{}); }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's DMA pieces of the following K tile have landed (and its epilogue stores retired)
                if constexpr (last) tail_pin();
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    origin(b, m0, n0);
    unsigned offX[PX], offW[PW];
    offsets(m0, n0, offX, offW);
    // ---- pipeline fill for the first output tile of this workgroup: K tiles 0 and 1
    static_for<PX>([&](auto J) { dmaX(offX, 0, 0, J); });
    static_for<PW>([&](auto J) { dmaW(offW, 0, 0, J); });
    static_for<PX>([&](auto J) { dmaX(offX, 1, 1, J); });
    static_for<PW>([&](auto J) { dmaW(offW, 1, 1, J); });
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PX + PW) : "memory");
    __syncthreads();
    int sb = 0;                                   // stage that holds K tile 0 of the current output tile
#ifdef CFSAR_DEV
    int trace_i = 0;
#define CFSAR_TRACE4(slot) do { if (p.trace && tid == 0 && trace_i < 64) p.trace[((size_t)blockIdx.x * 64 + trace_i) * 4 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CFSAR_TRACE4(slot) do { } while (0)
#endif
    for (;;) {
        const int bn = b + grid;
        const bool has_next = bn < nt;
        int m0n = m0, n0n = n0;
        if (has_next) origin(bn, m0n, n0n);
        static_for<NL>([&](auto J) { load_one(sb, 0, J, xfA, wfA); });
        CFSAR_TRACE4(0);
#ifdef CFSAR_DEV
        if (p.dbg & (1 << 19)) __builtin_amdgcn_s_setprio(1);          // A/B: the K loop outranks the partner workgroup's epilogue
#endif
        int kt = 0;
        // nk >= 3 (launcher).  The second-to-last step ALWAYS prefetches K tile 0 of the "next" tile (after this workgroup's last tile that is the
        // current one again: the surplus loads re-read valid memory into the free stage and are waited for by the last step).
        step(sb & 1, (sb + 1) & 1, offX, offW, 2, T_{}, T_{}, T_{}, F_{}, F_{});
        for (kt = 1; kt < nk - 2; ++kt) step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, kt + 2, T_{}, T_{}, F_{}, F_{}, F_{});
        offsets(m0n, n0n, offX, offW);
        step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, 0, T_{}, T_{}, F_{}, T_{}, F_{});      // K tile 0 of the next tile -> this step's stage
        ++kt;
        step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, 0, F_{}, F_{}, F_{}, F_{}, T_{});      // last step: its stage becomes the slab
        tail_fold();
        CFSAR_TRACE4(1);
#ifdef CFSAR_DEV
        if (p.dbg & (1 << 19)) __builtin_amdgcn_s_setprio(0);
#endif
        char* slab = smem + ((sb + kt) & 1) * STG + wave * 1024;
#ifdef CFSAR_DEV
        if (p.dbg & 4) {                                                // ablation: no epilogue (keep the accumulators live)
            if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.out)[0] = acc[1][1][3] + acc[MIW - 1][1][2] + acc[2][0][1];
        } else
#endif
        {
            const int mb = m0 + wm * WR, nb = n0 + wn * 64;
            const u32x4 rv0[4] = {};
            if constexpr (WIDE) {
                if (mb + WR <= p.M && nb + 64 <= p.N) epilogue_rows_wide<STORE, true, MIW, CH>(acc, p, mb, nb, lane, slab);
                else epilogue_rows_wide<STORE, false, MIW, CH>(acc, p, mb, nb, lane, slab);
            } else if (mb + WR <= p.M && nb + 64 <= p.N) epilogue_rows<TO, ACT, HAS_RES, STORE, true, LNFOLD, false, false, MIW, CH>(acc, p, mb, nb, lane, slab, rscale, rv0);
            else epilogue_rows<TO, ACT, HAS_RES, STORE, false, LNFOLD, false, false, MIW, CH>(acc, p, mb, nb, lane, slab, rscale, rv0);
        }
        CFSAR_TRACE4(2);
#ifdef CFSAR_DEV
        ++trace_i;
#endif
        if (!has_next) break;
        // K tile 1 of the next tile into the stage the epilogue has just used: this wave's pieces cover its own slab chunks (its LDS reads
        // have returned: their values went into the stores above) and chunks no epilogue touches
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_for<PX>([&](auto J) { dmaX(offX, 1, (sb + kt) & 1, J); });
        static_for<PW>([&](auto J) { dmaW(offW, 1, (sb + kt) & 1, J); });
        sb = (sb + nk) & 1;
        b = bn;
        m0 = m0n;
        n0 = n0n;
    }
}

constexpr int LDS4 = 2 * (64 * 3 * ROWB + TN4 * ROWB);     // 81 920 B = half a CU's LDS

template <typename TI, typename TO, int ACT, int MODE, int STORE>
int launch4_inst(const VitGemmArgs& a, hipStream_t s) {
    auto* fn = &vit_gemm4_kernel<TI, TO, ACT, MODE, STORE, 3>;
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(fn), LDS4, "cfsar_gemm(vit4)")) return rc;
    const int full = 2 * (cfsar_num_cus() & ~7);            // two workgroups per CU; the block -> XCD walk assumes a multiple of 8
    const int grid = a.ntiles < full ? ((a.ntiles + 7) & ~7) : full;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), LDS4, s, a);
    return cfsar_check_launch("cfsar_gemm(vit4)");
}

template <int STORE>
int launch4_path(const VitGemmArgs& a, int mode, bool f16io, hipStream_t s) {
    if (mode == 6) return launch4_inst<_Float16, _Float16, CFSAR_ACT_NONE, 6, STORE>(a, s);
    if (mode == 1) {
        if (f16io) return launch4_inst<_Float16, _Float16, CFSAR_ACT_NONE, 1, STORE>(a, s);
        return launch4_inst<__bf16, _Float16, CFSAR_ACT_NONE, 1, STORE>(a, s);
    }
    if (mode == 2) {
        if (f16io) {
            if (a.act == CFSAR_ACT_QUICKGELU) return launch4_inst<_Float16, _Float16, CFSAR_ACT_QUICKGELU, 2, STORE>(a, s);
            return launch4_inst<_Float16, _Float16, CFSAR_ACT_NONE, 2, STORE>(a, s);
        }
        if (a.act == CFSAR_ACT_QUICKGELU) return launch4_inst<_Float16, __bf16, CFSAR_ACT_QUICKGELU, 2, STORE>(a, s);
        return launch4_inst<_Float16, __bf16, CFSAR_ACT_NONE, 2, STORE>(a, s);
    }
    if (mode != 0 || a.relu) return -2;
    if (a.act == CFSAR_ACT_QUICKGELU) return launch4_inst<__bf16, __bf16, CFSAR_ACT_QUICKGELU, 0, STORE>(a, s);
    return launch4_inst<__bf16, __bf16, CFSAR_ACT_NONE, 0, STORE>(a, s);
}

}  // namespace

// The two-workgroups-per-CU form of cfsar_gemm_vit_try's launch: `a0` as the 8-wave launcher filled it; tiles are re-derived for 192 x 128.
// -2 = this form does not cover the call (head-blocked layouts, K < 192, RN50's bf16 residual / ReLU instances, fused statistics).
int cfsar_gemm_vit4_launch(const VitGemmArgs& a0, int mode, bool f16io, int store, hipStream_t s) {
    if (a0.hb_tokens > 0 || a0.ha_tokens > 0 || a0.K < 192 || a0.part != nullptr || a0.relu || mode == 5 || mode == 3) return -2;
    VitGemmArgs a = a0;
    a.miw = 3;
    a.tiles_n = (a.N + TN4 - 1) / TN4;
    a.ntiles = ((a.M + 191) / 192) * a.tiles_n;
    if (store == 2) return launch4_path<2>(a, mode, f16io, s);
    if (store == 0) return launch4_path<0>(a, mode, f16io, s);
    return -2;
}
