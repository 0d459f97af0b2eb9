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
    const float* res;
    int M, N, K;
    int lda, ldw, ldo, ldr;
    int act;
    int row_group, row_gap, row_off, res_mod, res_off;
    int tiles_n;
};

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ void glds16(const char* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == CFSAR_ACT_QUICKGELU) return v / (1.0f + __expf(-1.702f * v));
    if (act == CFSAR_ACT_GELU_ERF) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
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

    // ---- epilogue: D[n][m] layout -> lane owns token row m = ..+lr and columns 8g+4hi..+3 of each 32-col tile
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + wm * 64 + mi * 32 + lr;
        if (m >= p.M) continue;
        int orow = m + p.row_off;
        if (p.row_group > 0) orow += (m / p.row_group) * p.row_gap;
        const int rrow = p.res_mod > 0 ? (m % p.res_mod) + p.res_off : orow;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * hi;
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
                    const float4 rv = *reinterpret_cast<const float4*>(p.res + (size_t)rrow * p.ldr + n);
                    v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
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

template <typename TI, typename TO>
int launch(const GemmArgs& a, hipStream_t s) {
    const int tiles_m = (a.M + BM - 1) / BM;
    hipLaunchKernelGGL((gemm_kernel<TI, TO>), dim3(tiles_m * a.tiles_n), dim3(NTHREADS), 0, s, a);
    return cfsar_check_launch("cfsar_gemm");
}

}  // namespace

extern "C" int cfsar_gemm(const void* A, const void* W, void* out, const float* bias, const float* residual, int M,
                          int N, int K, int lda, int ldw, int ldo, int ldr, int in_dtype, int out_dtype, int act,
                          int row_group, int row_gap, int row_off, int res_mod, int res_off, cfsar_stream_t stream) {
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
    GemmArgs a;
    a.A = static_cast<const char*>(A);
    a.W = static_cast<const char*>(W);
    a.out = out;
    a.bias = bias;
    a.res = residual;
    a.M = M; a.N = N; a.K = K;
    a.lda = lda; a.ldw = ldw; a.ldo = ldo; a.ldr = ldr;
    a.act = act;
    a.row_group = row_group; a.row_gap = row_gap; a.row_off = row_off;
    a.res_mod = res_mod; a.res_off = res_off;
    a.tiles_n = (N + BN - 1) / BN;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (in_dtype == CFSAR_BF16)
        return out_dtype == CFSAR_BF16 ? launch<__bf16, __bf16>(a, s) : launch<__bf16, float>(a, s);
    return out_dtype == CFSAR_BF16 ? launch<float, __bf16>(a, s) : launch<float, float>(a, s);
}
