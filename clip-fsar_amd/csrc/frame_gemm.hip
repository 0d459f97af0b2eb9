// The fp16 numerics mode's per-frame GEMMs (round 5): [frames, K] x [K, N] products of per-frame token MEANS with a weight matrix --
// the low-word correction of every block GEMM (corr[f] = mean_t(operand) W_lo^T, fp32 out) and the update of the residual stream's per-frame
// mean (xbar[f] += mean_t(operand) W^T + b, bf16 in place); engine.py::HipViT.forward, six per block (few_shot.py:623, 626-640 are the GEMMs
// they serve).  frames = 80 per episode: at one episode per call these ran on the generic 128 x 128 MFMA kernel at 26-41 us each -- 6-24
// workgroups walking K serially, a third of the fp16 mode's GPU time -- and at 16 episodes on the 256 x 128 kernel at ~25 us.
//
// Shape of the problem: the WEIGHT matrix is the big operand (1.2-4.7 MB), streamed once; the activation side is tiny.  So: one workgroup
// per (16 output columns, 64 frames); its four waves split K between them (wave w takes the 32-wide k steps s = w, w + 4, ...: together they
// read whole 256-byte runs of each weight row) and feed v_mfma_f32_16x16x32_bf16 straight from global memory (16 weight rows x 64 B and
// 16 frame rows x 64 B per load instruction, one step ahead in registers); the four partial tiles meet in LDS and are added in a FIXED
// order.  48 ... 192 workgroups per 64 frames instead of 6 ... 24, each with K / 128 dependent steps instead of K / 64.
//
// Determinism: an output element's value is ((p0 + p1) + p2) + p3 with p_w the MFMA chain over wave w's k steps -- a function of its row
// and column alone, not of how many frames the call carries: an episode's logits stay bit-identical whatever batch it is served in
// (tests/test_gpu_e2e.py::test_fp16_mode_b16_equals_b1), which is why EVERY call of these two GEMMs takes this kernel, whatever M.
#include "common.h"

namespace {

constexpr int FG_MT = 4;                       // 16-frame tiles per workgroup (64 frames)

// NT = 16-column tiles per workgroup: 1 at one or two episodes per call (as many workgroups as possible: the call is latency-bound), 4 at batch
// scale (a frame fragment then serves four weight fragments and vice versa: at 1 280 frames the one-tile form re-read the frame operand once per
// 16 columns -- 94-377 MB through the L2 per call, 42-47 us -- and lost to the generic kernel).  The K split and the MFMA chain of an output
// element are the same in both: the bits do not depend on NT.
template <bool OUT_BF16, int NT>
__global__ __launch_bounds__(256) void frame_gemm_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ W, void* out,
                                                         const float* __restrict__ bias, const __bf16* res, int M, int N, int K) {
    __shared__ f32x4 red[4][FG_MT][64];                               // 16 KiB: every wave's partial tiles of ONE column tile at a time
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT), m0 = blockIdx.y * (16 * FG_MT);
    // MFMA "A" operand = 16 weight rows (output columns), "B" operand = 16 frame rows: D[n = 4 (lane >> 4) + reg][m = lane & 15]
    const __bf16* wp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wp[j] = W + (size_t)(n0 + 16 * j + r) * K + kq * 8;
    const __bf16* ap[FG_MT];
#pragma unroll
    for (int t = 0; t < FG_MT; ++t) {
        int row = m0 + 16 * t + r;
        row = row < M ? row : M - 1;                                  // clamped: rows past M are computed and dropped
        ap[t] = A + (size_t)row * K + kq * 8;
    }
    f32x4 acc[NT][FG_MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < FG_MT; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = K >> 5;                                        // K % 128 == 0 (launcher): every wave gets nsteps / 4 steps
    uint4 wf[NT], af[FG_MT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[j] = *reinterpret_cast<const uint4*>(wp[j] + wave * 32);
#pragma unroll
    for (int t = 0; t < FG_MT; ++t) af[t] = *reinterpret_cast<const uint4*>(ap[t] + wave * 32);
    for (int s = wave; s < nsteps; s += 4) {
        const int sn = s + 4 < nsteps ? s + 4 : s;                    // next step's operands in flight under this step's MFMAs (last: a harmless re-read)
        uint4 wn[NT], an[FG_MT];
#pragma unroll
        for (int j = 0; j < NT; ++j) wn[j] = *reinterpret_cast<const uint4*>(wp[j] + sn * 32);
#pragma unroll
        for (int t = 0; t < FG_MT; ++t) an[t] = *reinterpret_cast<const uint4*>(ap[t] + sn * 32);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int t = 0; t < FG_MT; ++t)
                acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j]), __builtin_bit_cast(bf16x8, af[t]), acc[j][t], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NT; ++j) wf[j] = wn[j];
#pragma unroll
        for (int t = 0; t < FG_MT; ++t) af[t] = an[t];
    }
    // per column tile: the four waves' partials meet in LDS; wave w finishes frame tile w -- the partials in wave order (fixed), then the
    // epilogue; a lane owns 4 consecutive columns of one frame
    const int t = wave;
    const int m = m0 + 16 * t + r;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (j > 0) __syncthreads();                                   // the previous column tile's partials have been read
#pragma unroll
        for (int tt = 0; tt < FG_MT; ++tt) red[wave][tt][lane] = acc[j][tt];
        __syncthreads();
        f32x4 v = red[0][t][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const f32x4 p = red[w][t][lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += p[i];
        }
        const int n = n0 + 16 * j + 4 * kq;
        if (m < M) {
            if constexpr (OUT_BF16) {
                bf16x4 o;
                const bf16x4 rv = res ? *reinterpret_cast<const bf16x4*>(res + (size_t)m * N + n) : bf16x4{0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (__bf16)(v[i] + (bias ? bias[n + i] : 0.f) + (float)rv[i]);
                *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(out) + (size_t)m * N + n) = o;
            } else {
                if (bias) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] += bias[n + i];
                }
                *reinterpret_cast<f32x4*>(static_cast<float*>(out) + (size_t)m * N + n) = v;
            }
        }
    }
}

template <bool OUT_BF16, int NT>
void frame_gemm_launch(const void* A, const void* W, void* out, const float* bias, const void* res, int M, int N, int K, hipStream_t s) {
    const dim3 grid((unsigned)(N / (16 * NT)), (unsigned)((M + 16 * FG_MT - 1) / (16 * FG_MT)));
    hipLaunchKernelGGL((frame_gemm_kernel<OUT_BF16, NT>), grid, dim3(256), 0, s, static_cast<const __bf16*>(A), static_cast<const __bf16*>(W), out,
                       bias, static_cast<const __bf16*>(res), M, N, K);
}

}  // namespace

extern "C" int cfsar_frame_gemm(const void* A, const void* W, void* out, const float* bias, const void* res, int M, int N, int K,
                                int out_dtype, cfsar_stream_t stream) {
    CFSAR_REQUIRE(A && W && out, "cfsar_frame_gemm: null pointer");
    CFSAR_REQUIRE(M > 0 && N > 0 && K >= 128 && K % 128 == 0 && N % 16 == 0, "cfsar_frame_gemm: bad shape M=%d N=%d K=%d (K %% 128, N %% 16)", M, N, K);
    CFSAR_REQUIRE(out_dtype == CFSAR_F32 || out_dtype == CFSAR_BF16, "cfsar_frame_gemm: out_dtype must be fp32 or bf16, got %d", out_dtype);
    CFSAR_REQUIRE(out_dtype == CFSAR_BF16 || res == nullptr, "cfsar_frame_gemm: a residual exists for the bf16 form only");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool wide = M > 128 && N % 64 == 0;          // four column tiles per workgroup from three episodes on (same bits either way)
    if (out_dtype == CFSAR_BF16) {
        if (wide) frame_gemm_launch<true, 4>(A, W, out, bias, res, M, N, K, s);
        else frame_gemm_launch<true, 1>(A, W, out, bias, res, M, N, K, s);
    } else {
        if (wide) frame_gemm_launch<false, 4>(A, W, out, bias, res, M, N, K, s);
        else frame_gemm_launch<false, 1>(A, W, out, bias, res, M, N, K, s);
    }
    return cfsar_check_launch("cfsar_frame_gemm");
}
