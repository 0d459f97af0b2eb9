// The ViT-block GEMMs of gemm_vit.hip with ONE wave per SIMD (round 5): a 256-thread workgroup per CU, 256 x 256 output tiles, 2 x 2 waves
// with 128 x 128 wave tiles -- 256 accumulators per lane, the whole accumulator half of the unified 512-entry register file.
//
// Why (profiles/r05_gemm_forms.md): the 8-wave kernel's waves own 128 x 64 tiles, i.e. 6 fragment reads (24 KiB of LDS traffic per wave
// and K tile) for 8 MFMAs; a 128 x 128 wave tile reads 8 fragments for 16 MFMAs -- a third less LDS -> register traffic per FLOP on a chip
// that runs these kernels power-limited -- one barrier per 64 MFMAs of a wave instead of per 32, and no second wave on the SIMD whose
// MFMAs come out of this wave's stream.  The price: nothing covers this wave's own stalls, so its instruction stream has to keep the
// matrix pipe fed by itself (fragments of sub-step s + 1 are read during sub-step s, LDS-DMA pieces two K tiles ahead).
//
// Everything else is gemm_vit.hip's early-DMA form (OPATH 2): two 64 KiB LDS stages with XOR-swizzled 128-byte rows filled by LDS-DMA,
// the operand pipeline runs through tile boundaries, wave-private 4 KiB epilogue slabs above the stages, tail MFMA for bias / LayerNorm
// terms / per-frame correction, epilogues of gemm_vit_epi.h (called once per 64-column half of the wave tile).
#include "gemm_vit_epi.h"

namespace {

template <typename TI, typename TO, int ACT, int MODE, int STORE>
__global__ __launch_bounds__(256, 1) void vit_gemm1w_kernel(VitGemmArgs p) {
    constexpr int MIW = 4, NIW = 4;               // 32 x 32 MFMA tiles per wave: 4 x 4
    constexpr int NWV = 4;                        // waves
    constexpr int WR = 128, WC = 128;             // wave tile
    constexpr int NM = MIW * NIW;                 // MFMAs per sub-step (16)
    constexpr int NL = MIW + NIW;                 // fragment loads per sub-step (8)
    constexpr int XB = TM * ROWB, STG = XB + TN * ROWB;               // 32 KiB + 32 KiB
    constexpr int PX = TM / (8 * NWV), PW = TN / (8 * NWV);           // LDS-DMA pieces per wave per K tile: 8 + 8
    constexpr int PCH = NWV * 1024;               // distance of a wave's consecutive pieces
    constexpr int SLAB0 = 2 * STG;                // wave-private 4 KiB epilogue slabs above the two stages
    constexpr bool WIDE = MODE == 6;
    constexpr bool HAS_RES = MODE == 1 || WIDE;
    constexpr bool LNFOLD = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, hi = lane >> 5;

    const int nt = p.ntiles, grid = (int)gridDim.x;
    const int xq = nt >> 3, xr = nt & 7;
    const int tiles_m = nt / p.tiles_n;
    auto origin = [&](int b, int& m0, int& n0) __attribute__((always_inline)) {
        const int xcd = b & 7;
        const int lin = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (b >> 3);
        int tm, tn;
        tile_of(lin, tiles_m, p.tiles_n, p.group, p.colfast, tm, tn);
        m0 = tm * TM;
        n0 = tn * TN;
#ifdef CFSAR_DEV
        if (p.dbg & 8) { m0 = 0; n0 = 0; }
#endif
    };
    auto offsets = [&](int m0, int n0, unsigned (&ox)[PX], unsigned (&ow)[PW]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            const int row = (i * NWV + wave) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ swz(row);
            int gm = m0 + row;
            gm = gm < p.M ? gm : p.M - 1;
            ox[i] = (unsigned)gm * (unsigned)p.lda * 2u + chunk * 16;
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int row = (i * NWV + wave) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ swz(row);
            int gn = n0 + row;
            gn = gn < p.N ? gn : p.N - 1;
            ow[i] = (unsigned)gn * (unsigned)p.ldw * 2u + chunk * 16;
        }
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    int rdX[MIW], rdW[NIW];
#pragma unroll
    for (int i = 0; i < MIW; ++i) {
        const int rx = wm * WR + i * 32 + lr;
        rdX[i] = rx * ROWB + ((hi ^ swz(rx)) << 4);
    }
#pragma unroll
    for (int i = 0; i < NIW; ++i) {
        const int rw = wn * WC + i * 32 + lr;
        rdW[i] = XB + rw * ROWB + ((hi ^ swz(rw)) << 4);
    }

    const int nk = p.K / 64;
    int b = blockIdx.x;
    if (b >= nt) return;
    int m0 = 0, n0 = 0;
    f32x16 acc[2][MIW][2];                        // [64-column half][mi][ni within the half]
    // ---- tail operands: [0..3] bias | c (hi == 0) / d (hi == 1) of column 32 ni + lr; LN-folded: [4..7] mean | std, [8..11] 1 / std of row 32 mi + lr
    constexpr int NTL = LNFOLD ? 12 : 4;
    float tl[NTL] = {};
    constexpr bool CORR = std::is_same<TO, _Float16>::value && std::is_same<TI, _Float16>::value && (MODE == 2 || MODE == 6);
    float tcq[NIW] = {0.f, 0.f, 0.f, 0.f};
    int corr_bnd = 0, corr_par = 0;
    float rscale[4] = {1.f, 1.f, 1.f, 1.f};
    auto tail_loads = [&](int m0_, int n0_) __attribute__((always_inline)) {
        const float* cd = LNFOLD ? (hi ? p.bias : p.cvec) : p.bias;
#pragma unroll
        for (int ni = 0; ni < NIW; ++ni) {
            int nb_ = n0_ + wn * WC + (ni >> 1) * 64;
            nb_ = nb_ + 64 <= p.N ? nb_ : p.N - 64;
            tl[ni] = cd[nb_ + (ni & 1) * 32 + lr];
        }
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
                for (int ni = 0; ni < NIW; ++ni) {
                    int nb_ = n0_ + wn * WC + (ni >> 1) * 64;
                    nb_ = nb_ + 64 <= p.N ? nb_ : p.N - 64;
                    tcq[ni] = p.corr[(size_t)f * p.N + nb_ + (ni & 1) * 32 + lr];
                }
            }
        }
        if constexpr (LNFOLD) {
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi) {
                int m = m0_ + wm * WR + mi * 32 + lr;
                m = m < p.M ? m : p.M - 1;
                tl[4 + mi] = p.rowstats[(size_t)m * 4 + hi];
                tl[8 + mi] = p.rowstats[(size_t)m * 4 + 2];
            }
        }
    };
    auto tail_pin = [&]() __attribute__((always_inline)) {
        if constexpr (CORR) asm volatile("" : "+v"(tcq[0]), "+v"(tcq[1]), "+v"(tcq[2]), "+v"(tcq[3]));
        if constexpr (LNFOLD)
            asm volatile("" : "+v"(tl[0]), "+v"(tl[1]), "+v"(tl[2]), "+v"(tl[3]), "+v"(tl[4]), "+v"(tl[5]), "+v"(tl[6]), "+v"(tl[7]), "+v"(tl[8]), "+v"(tl[9]), "+v"(tl[10]), "+v"(tl[11]));
        else
            asm volatile("" : "+v"(tl[0]), "+v"(tl[1]), "+v"(tl[2]), "+v"(tl[3]));
    };
    auto tail_fold = [&]() __attribute__((always_inline)) {
        if constexpr (!LNFOLD) {
            typedef typename Vec2B<TI>::v8 TI8;
            TI8 bw[NIW];
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) {
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
                for (int ni = 0; ni < NIW; ++ni) {
                    f32x16& a = acc[ni >> 1][mi][ni & 1];
                    if constexpr (std::is_same<TI, _Float16>::value) a = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[ni], ones, a, 0, 0, 0);
                    else a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[ni], ones, a, 0, 0, 0);
                }
            }
        } else {
            f16x8 cw[NIW], mx[MIW];
#pragma unroll
            for (int ni = 0; ni < NIW; ++ni) {
                const _Float16 h = (_Float16)tl[ni], l = (_Float16)(tl[ni] - (float)h);
                _Float16 q = (_Float16)0.f;
                if constexpr (CORR) q = (_Float16)tcq[ni];
                cw[ni] = f16x8{h, h, l, q, 0, 0, 0, 0};
            }
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi) {
                const float nm = hi ? tl[4 + mi] : -tl[4 + mi];
                const _Float16 h = (_Float16)nm, l = (_Float16)(nm - (float)h);
                _Float16 sdh = (_Float16)0.f;
                if constexpr (CORR) {
                    const bool in0 = mi * 32 + lr < corr_bnd;
                    sdh = (_Float16)((p.corr != nullptr && ((in0 ? corr_par : corr_par ^ 1) == hi)) ? (p.corr_raw ? 1.0f : __builtin_amdgcn_rcpf(tl[8 + mi])) : 0.f);
                }
                mx[mi] = f16x8{h, l, h, sdh, 0, 0, 0, 0};
                rscale[mi] = tl[8 + mi];
            }
#pragma unroll
            for (int mi = 0; mi < MIW; ++mi)
#pragma unroll
                for (int ni = 0; ni < NIW; ++ni)
                    acc[ni >> 1][mi][ni & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cw[ni], mx[mi], acc[ni >> 1][mi][ni & 1], 0, 0, 0);
        }
    };

    uint4 xfA[MIW], wfA[NIW], xfB[MIW], wfB[NIW];
    auto dmaX = [&](const unsigned (&ox)[PX], int kt, int stage, auto J) __attribute__((always_inline)) {
        const char* src = p.A + (size_t)(kt >= p.nka ? kt - p.nka : kt) * ROWB + ox[decltype(J)::value];
        glds16_asm(src, ldsw + (unsigned)stage * (unsigned)STG + (unsigned)decltype(J)::value * (unsigned)PCH);
    };
    auto dmaW = [&](const unsigned (&ow)[PW], int kt, int stage, auto J) __attribute__((always_inline)) {
        const char* src = p.W + (size_t)kt * ROWB + ow[decltype(J)::value];
        glds16_asm(src, ldsw + (unsigned)stage * (unsigned)STG + (unsigned)XB + (unsigned)decltype(J)::value * (unsigned)PCH);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    // read order = order of first use by the MFMA sequence (ni-major): x0 w0 x1 x2 x3 w1 w2 w3
    auto load_one = [&](int stage, int ss, auto J, uint4 (&xf)[MIW], uint4 (&wf)[NIW]) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        const char* base = smem + stage * STG;
        const int x2 = ss << 5;
        constexpr int isx[8] = {1, 0, 1, 1, 1, 0, 0, 0};
        constexpr int idx[8] = {0, 0, 1, 2, 3, 1, 2, 3};
        if constexpr (isx[j] != 0) xf[idx[j]] = *reinterpret_cast<const uint4*>(base + (rdX[idx[j]] ^ x2));
        else wf[idx[j]] = *reinterpret_cast<const uint4*>(base + (rdW[idx[j]] ^ x2));
    };
    auto mfma_one = [&](auto J, uint4 (&xf)[MIW], uint4 (&wf)[NIW], auto ZERO) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        constexpr int ni = j / MIW, mi = j % MIW;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const f32x16 c = decltype(ZERO)::value ? zero : acc[ni >> 1][mi][ni & 1];
        acc[ni >> 1][mi][ni & 1] = cfsar_mfma_32x32x16<TI>(wf[ni], xf[mi], c);
    };
    // One 128-byte K tile = 4 sub-steps of 16 MFMAs; `cur` / `nxt` = LDS stages of this K tile / the following one (gemm_vit.hip's OPATH 2 step).
    auto step = [&](int cur, int nxt, const unsigned (&ox)[PX], const unsigned (&ow)[PW], int ksrc, auto FRAGS, auto ZERO, auto TAIL, auto LAST) __attribute__((always_inline)) {
        constexpr bool frags = decltype(FRAGS)::value, tail = decltype(TAIL)::value, last = decltype(LAST)::value;
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
            if constexpr (j >= 2 && j < 2 + NL && frags) load_one(nxt, 0, std::integral_constant<int, j - 2>{}, xfA, wfA);
            if constexpr (tail && j == 2) tail_loads(m0, n0);
            // behind the barrier every wave has completed its fragment reads of stage `cur`: K tile `ksrc` (two ahead) starts its flight, 8 + 8
            // pieces over the slots j = 2 .. 13
            if constexpr (j >= 2 && j < 10) dmaX(ox, ksrc, cur, std::integral_constant<int, j - 2>{});
            if constexpr (j >= 10 && j < 14) {
                dmaW(ow, ksrc, cur, std::integral_constant<int, 2 * (j - 10)>{});
                dmaW(ow, ksrc, cur, std::integral_constant<int, 2 * (j - 10) + 1>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's DMA pieces of the following K tile have landed
                if constexpr (last) tail_pin();                         // ... and so have the tail operands (fetched one step earlier)
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    origin(b, m0, n0);
    unsigned offX[PX], offW[PW];
    offsets(m0, n0, offX, offW);
    static_for<PX>([&](auto J) { dmaX(offX, 0, 0, J); });
    static_for<PW>([&](auto J) { dmaW(offW, 0, 0, J); });
    static_for<PX>([&](auto J) { dmaX(offX, 1, 1, J); });
    static_for<PW>([&](auto J) { dmaW(offW, 1, 1, J); });
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PX + PW) : "memory");
    __syncthreads();
    int sb = 0;
    char* slab = smem + SLAB0 + wave * EPI_SLAB;
#ifdef CFSAR_DEV
    int trace_i = 0;
#define CFSAR_TRACE1(slot) do { if (p.trace && tid == 0 && trace_i < 64) p.trace[((size_t)blockIdx.x * 64 + trace_i) * 4 + (slot)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CFSAR_TRACE1(slot) do { } while (0)
#endif
    for (;;) {
        const int bn = b + grid;
        const bool has_next = bn < nt;
        int m0n = m0, n0n = n0;
        if (has_next) origin(bn, m0n, n0n);
        static_for<NL>([&](auto J) { load_one(sb, 0, J, xfA, wfA); });
        CFSAR_TRACE1(0);
        int kt = 0;
        // nk >= 3.  The tail steps ALWAYS prefetch (after this workgroup's last tile the "next" origin is the current one: the surplus loads
        // re-read valid memory into free stages and are waited for before the kernel ends).
        step(sb & 1, (sb + 1) & 1, offX, offW, 2, T_{}, T_{}, F_{}, F_{});
        for (kt = 1; kt < nk - 2; ++kt) step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, kt + 2, T_{}, F_{}, F_{}, F_{});
        offsets(m0n, n0n, offX, offW);
        step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, 0, T_{}, F_{}, T_{}, F_{});      // K tile 0 of the next tile
        ++kt;
        step((sb + kt) & 1, (sb + kt + 1) & 1, offX, offW, 1, F_{}, F_{}, F_{}, T_{});      // K tile 1 of the next tile: in flight through the epilogue
        tail_fold();
        CFSAR_TRACE1(1);
#ifdef CFSAR_DEV
        if (p.dbg & 4) {                                                // ablation: no epilogue (keep the accumulators live)
            if (acc[0][0][0][0] == 123.456f) reinterpret_cast<float*>(p.out)[0] = acc[1][1][1][3] + acc[0][3][1][2] + acc[1][2][0][1];
        } else
#endif
        {
            const u32x4 rv0[4] = {};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int mb = m0 + wm * WR, nb = n0 + wn * WC + h * 64;
                if constexpr (WIDE) {
                    if (mb + WR <= p.M && nb + 64 <= p.N) epilogue_rows_wide<STORE, true, MIW>(acc[h], p, mb, nb, lane, slab);
                    else epilogue_rows_wide<STORE, false, MIW>(acc[h], p, mb, nb, lane, slab);
                } else if (mb + WR <= p.M && nb + 64 <= p.N) epilogue_rows<TO, ACT, HAS_RES, STORE, true, LNFOLD, false, false, MIW>(acc[h], p, mb, nb, lane, slab, rscale, rv0);
                else epilogue_rows<TO, ACT, HAS_RES, STORE, false, LNFOLD, false, false, MIW>(acc[h], p, mb, nb, lane, slab, rscale, rv0);
            }
        }
        CFSAR_TRACE1(2);
#ifdef CFSAR_DEV
        ++trace_i;
#endif
        if (!has_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the surplus DMA of the last step targets this workgroup's LDS
            break;
        }
        sb = (sb + nk) & 1;
        b = bn;
        m0 = m0n;
        n0 = n0n;
    }
}

constexpr int LDS1W = 2 * (TM + TN) * ROWB + 4 * EPI_SLAB;     // 147 456 B

template <typename TI, typename TO, int ACT, int MODE, int STORE>
int launch1w_inst(const VitGemmArgs& a, hipStream_t s) {
    auto* fn = &vit_gemm1w_kernel<TI, TO, ACT, MODE, STORE>;
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(fn), LDS1W, "cfsar_gemm(vit1w)")) return rc;
    const int full = cfsar_num_cus() & ~7;
    const int grid = a.ntiles < full ? ((a.ntiles + 7) & ~7) : full;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), LDS1W, s, a);
    return cfsar_check_launch("cfsar_gemm(vit1w)");
}

template <int STORE>
int launch1w_path(const VitGemmArgs& a, int mode, bool f16io, hipStream_t s) {
    if (mode == 1 && !f16io) return launch1w_inst<__bf16, _Float16, CFSAR_ACT_NONE, 1, STORE>(a, s);
    if (mode == 2 && !f16io) {
        if (a.act == CFSAR_ACT_QUICKGELU) return launch1w_inst<_Float16, __bf16, CFSAR_ACT_QUICKGELU, 2, STORE>(a, s);
        return launch1w_inst<_Float16, __bf16, CFSAR_ACT_NONE, 2, STORE>(a, s);
    }
    if (mode != 0 || a.relu) return -2;
    if (a.act == CFSAR_ACT_QUICKGELU) return launch1w_inst<__bf16, __bf16, CFSAR_ACT_QUICKGELU, 0, STORE>(a, s);
    return launch1w_inst<__bf16, __bf16, CFSAR_ACT_NONE, 0, STORE>(a, s);
}

}  // namespace

// The one-wave-per-SIMD form of cfsar_gemm_vit_try's launch (256 x 256 tiles; bf16-mode instances); -2 = not covered.
int cfsar_gemm_vit1w_launch(const VitGemmArgs& a0, int mode, bool f16io, int store, hipStream_t s) {
    if (a0.hb_tokens > 0 || a0.ha_tokens > 0 || a0.K < 192 || a0.part != nullptr || a0.relu || mode == 5 || mode == 3 || mode == 6 || f16io) return -2;
    VitGemmArgs a = a0;
    a.miw = 4;
    a.tiles_n = (a.N + TN - 1) / TN;
    a.ntiles = ((a.M + TM - 1) / TM) * a.tiles_n;
    if (store == 2) return launch1w_path<2>(a, mode, f16io, s);
    if (store == 0) return launch1w_path<0>(a, mode, f16io, s);
    return -2;
}
