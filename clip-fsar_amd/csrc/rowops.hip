// Memory-bound row kernels of the ViT tower: patch gather (im2col), class-token rows, LayerNorm.
#include "common.h"

namespace {

// ---- A2 stage 1: frames [F,3,H,W] f32 -> rows [(f,py,px)] x k_pad, k = c*P*P + dy*P + dx (few_shot.py:659,672-674).
// One thread moves two horizontally adjacent pixels (8-byte aligned because P is even); consecutive threads walk k,
// so stores are fully coalesced and loads come in P*4-byte runs.
template <typename TO>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ frames, TO* __restrict__ out, int F,
                                                     int H, int W, int P, int k_pad, long long total_pairs) {
    const int gw = W / P, gh = H / P;
    const int kp2 = k_pad >> 1;
    const int kreal = 3 * P * P;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_pairs;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long row = idx / kp2;
        const int k = (int)(idx - row * kp2) * 2;
        float2 v = make_float2(0.f, 0.f);
        if (k < kreal) {
            const int px = (int)(row % gw);
            const long long t = row / gw;
            const int py = (int)(t % gh);
            const long long f = t / gh;
            const int c = k / (P * P);
            const int rem = k - c * P * P;
            const int dy = rem / P, dx = rem - dy * P;
            const float* src = frames + ((f * 3 + c) * H + (py * P + dy)) * (long long)W + px * P + dx;
            v = *reinterpret_cast<const float2*>(src);
        }
        if constexpr (sizeof(TO) == 2) {
            typedef TO to2 __attribute__((ext_vector_type(2)));
            to2 o;
            o[0] = (TO)v.x;
            o[1] = (TO)v.y;
            *reinterpret_cast<to2*>(out + row * k_pad + k) = o;
        } else {
            *reinterpret_cast<float2*>(out + row * k_pad + k) = v;
        }
    }
}

// Round 6 (the fp16_strict mode): the same gather with the fp32 pixel kept as TWO fp16 words, laid out for a three-pass GEMM against
// [W_hi | W_hi | W_lo]: out row = [hi(k_pad) | lo(k_pad) | hi(k_pad)], hi = fp16(v), lo = fp16(v - hi).  The product hi W_hi + lo W_hi + hi W_lo
// carries ~22 bits of both operands (the lo x lo term is 2^-24 of the result) in one fp32 accumulation chain of the ordinary fp16 MFMA GEMM.
__global__ __launch_bounds__(256) void im2col_split_kernel(const float* __restrict__ frames, _Float16* __restrict__ out, int F,
                                                           int H, int W, int P, int k_pad, long long total_pairs) {
    const int gw = W / P, gh = H / P;
    const int kp2 = k_pad >> 1;
    const int kreal = 3 * P * P;
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_pairs;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long row = idx / kp2;
        const int k = (int)(idx - row * kp2) * 2;
        float2 v = make_float2(0.f, 0.f);
        if (k < kreal) {
            const int px = (int)(row % gw);
            const long long t = row / gw;
            const int py = (int)(t % gh);
            const long long f = t / gh;
            const int c = k / (P * P);
            const int rem = k - c * P * P;
            const int dy = rem / P, dx = rem - dy * P;
            v = *reinterpret_cast<const float2*>(frames + ((f * 3 + c) * H + (py * P + dy)) * (long long)W + px * P + dx);
        }
        h2 hi, lo;
        hi[0] = (_Float16)v.x;
        hi[1] = (_Float16)v.y;
        lo[0] = (_Float16)(v.x - (float)hi[0]);
        lo[1] = (_Float16)(v.y - (float)hi[1]);
        _Float16* dst = out + row * (3LL * k_pad) + k;
        *reinterpret_cast<h2*>(dst) = hi;
        *reinterpret_cast<h2*>(dst + k_pad) = lo;
        *reinterpret_cast<h2*>(dst + 2 * k_pad) = hi;
    }
}

// P = 16 (ViT-B/16), bf16 rows: one thread moves one image-row segment of a patch, 16 floats (four 16-byte loads = one 64-byte
// sector) -> 16 bf16 (two 16-byte stores).  Lane = dy + 16 * (patch within a group of 4): 16 consecutive lanes write 512 contiguous
// bytes of one output row, a wave instruction writes 4 x 512 B and reads 64 whole 64-byte sectors.  No division in the inner path
// (grid.y = frame x channel, grid.x walks patches).  2.85 -> ~5 TB/s on 1 280 frames.
template <typename TO>
__global__ __launch_bounds__(256) void im2col_p16_bf16_kernel(const float* __restrict__ frames, TO* __restrict__ out, int H, int W,
                                                              int k_pad) {
    typedef typename Vec2B<TO>::v8 TO8;
    const int gw = W >> 4, gh = H >> 4, npatch = gw * gh;
    const int fc = blockIdx.y;                       // frame * 3 + channel
    const int f = fc / 3, c = fc - 3 * f;
    const int dy = threadIdx.x & 15;
    const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= npatch) return;
    const int py = p / gw, px = p - py * gw;
    const float4* src = reinterpret_cast<const float4*>(frames + ((size_t)fc * H + (py * 16 + dy)) * W + px * 16);
    const float4 a = src[0], b = src[1], c4 = src[2], d = src[3];
    TO8 lo, hi8;
    lo[0] = (TO)a.x; lo[1] = (TO)a.y; lo[2] = (TO)a.z; lo[3] = (TO)a.w;
    lo[4] = (TO)b.x; lo[5] = (TO)b.y; lo[6] = (TO)b.z; lo[7] = (TO)b.w;
    hi8[0] = (TO)c4.x; hi8[1] = (TO)c4.y; hi8[2] = (TO)c4.z; hi8[3] = (TO)c4.w;
    hi8[4] = (TO)d.x; hi8[5] = (TO)d.y; hi8[6] = (TO)d.z; hi8[7] = (TO)d.w;
    TO* dst = out + ((size_t)f * npatch + p) * k_pad + c * 256 + dy * 16;
    *reinterpret_cast<TO8*>(dst) = lo;
    *reinterpret_cast<TO8*>(dst + 8) = hi8;
}

template <typename TX>
__global__ __launch_bounds__(256) void cls_rows_kernel(TX* __restrict__ x, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, int F, int ntok, int D) {
    const long long total = (long long)F * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / D;
        const int d = (int)(i - f * D);
        x[f * (long long)ntok * D + d] = (TX)(cls[d] + pos[d]);
    }
}

// ---- N2 (the step before the path): test-time frame transform of the few-shot dataset, reference
// datasets/base/ssv2_few_shot.py:614-642 = ToTensorVideo (uint8 THWC -> float CTHW / 255) -> KineticsResizedCropFewshot
// (datasets/utils/transformations.py:663-716: bilinear resize to (sh, sw), align_corners = False, then a crop window)
// -> NormalizeVideo(mean, std) -> permute to [T, 3, crop, crop].  One thread per output pixel (all 3 channels).
__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* __restrict__ src, float* __restrict__ out, int T,
                                                         int H, int W, int sh, int sw, int crop, int y0, int x0, float m0,
                                                         float m1, float m2, float is0, float is1, float is2) {
    const long long total = (long long)T * crop * crop;
    const float ry = (float)H / (float)sh, rx = (float)W / (float)sw;       // torch: scale = in / out
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % crop);
        const long long r = idx / crop;
        const int y = (int)(r % crop);
        const int t = (int)(r / crop);
        float fy = ry * ((float)(y + y0) + 0.5f) - 0.5f;                     // area_pixel_compute_source_index
        float fx = rx * ((float)(x + x0) + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int iy0 = (int)fy, ix0 = (int)fx;
        const int iy1 = iy0 + (iy0 < H - 1 ? 1 : 0), ix1 = ix0 + (ix0 < W - 1 ? 1 : 0);
        const float ly = fy - (float)iy0, lx = fx - (float)ix0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const unsigned char* f = src + (long long)t * H * W * 3;
        const unsigned char* p00 = f + ((long long)iy0 * W + ix0) * 3;
        const unsigned char* p01 = f + ((long long)iy0 * W + ix1) * 3;
        const unsigned char* p10 = f + ((long long)iy1 * W + ix0) * 3;
        const unsigned char* p11 = f + ((long long)iy1 * W + ix1) * 3;
        const float mean[3] = {m0, m1, m2}, istd[3] = {is0, is1, is2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float inv255 = 1.0f / 255.0f;
            const float v = hy * (hx * ((float)p00[c] * inv255) + lx * ((float)p01[c] * inv255)) +
                            ly * (hx * ((float)p10[c] * inv255) + lx * ((float)p11[c] * inv255));
            out[(((long long)t * 3 + c) * crop + y) * crop + x] = (v - mean[c]) * istd[c];
        }
    }
}

// ---- A3 LayerNorm: one wave per row, row held in registers (<= 16 float4 per lane), two-pass statistics in fp32,
// wavefront-shuffle reductions, vectorised 16-byte loads / 8- or 16-byte stores.  Each wave handles RPW rows at once so
// that 2x the loads are in flight per wave (the kernel is HBM-bound: 4 B in + 2 B out per element in bf16 mode).
constexpr int LN_MAXV = 16;
// One wave normalises RPW rows; a lane owns NV vectors of EPV = 16 / sizeof(TI) consecutive elements per row (16-byte loads:
// 4 floats or 8 halfs), statistics and arithmetic in fp32 (two-pass variance on the registers), 16-byte stores where the
// output type allows.  TI: float | _Float16 (the fp16 residual stream of the bf16 mode); TO: float | __bf16 | _Float16.
template <typename TI, typename TO, int NV, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, long long in_stride,
                                                        TO* __restrict__ out, long long out_stride,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        int rows, int D, float eps) {
    constexpr int EPV = 16 / (int)sizeof(TI);
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int nv = D / EPV;   // vectors per row
    float v[RPW][NV][EPV];
    float s[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r < rows ? row0 + r : rows - 1;
        const TI* xr = x + (long long)row * in_stride;
        s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int j = i * 64 + lane;
            if (j < nv) {
                if constexpr (sizeof(TI) == 2) {
                    const f16x8 hv = *reinterpret_cast<const f16x8*>(xr + EPV * j);
#pragma unroll
                    for (int e = 0; e < EPV; ++e) v[r][i][e] = (float)hv[e];
                } else {
                    const float4 fv = *reinterpret_cast<const float4*>(xr + EPV * j);
                    v[r][i][0] = fv.x; v[r][i][1] = fv.y; v[r][i][2] = fv.z; v[r][i][3] = fv.w;
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i * 64 + lane < nv) {
#pragma unroll
                for (int e = 0; e < EPV; e += 4) s[r] += (v[r][i][e] + v[r][i][e + 1]) + (v[r][i][e + 2] + v[r][i][e + 3]);
            }
    }
    float wv[NV][EPV], bv[NV][EPV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = i * 64 + lane;
        if (j < nv) {
#pragma unroll
            for (int e = 0; e < EPV; e += 4) {
                const float4 w4 = *reinterpret_cast<const float4*>(w + EPV * j + e);
                const float4 b4 = *reinterpret_cast<const float4*>(b + EPV * j + e);
                wv[i][e] = w4.x; wv[i][e + 1] = w4.y; wv[i][e + 2] = w4.z; wv[i][e + 3] = w4.w;
                bv[i][e] = b4.x; bv[i][e + 1] = b4.y; bv[i][e + 2] = b4.z; bv[i][e + 3] = b4.w;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const float mean = wave_sum(s[r]) / (float)D;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (i * 64 + lane < nv) {
#pragma unroll
                for (int e = 0; e < EPV; e += 4) {
                    const float a = v[r][i][e] - mean, bq = v[r][i][e + 1] - mean, c = v[r][i][e + 2] - mean, d = v[r][i][e + 3] - mean;
                    ss += (a * a + bq * bq) + (c * c + d * d);
                }
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
        if (row0 + r >= rows) continue;
        TO* orow = out + (long long)(row0 + r) * out_stride;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int j = i * 64 + lane;
            if (j < nv) {
                float o[EPV];
#pragma unroll
                for (int e = 0; e < EPV; ++e) o[e] = (v[r][i][e] - mean) * rstd * wv[i][e] + bv[i][e];
                if constexpr (sizeof(TO) == 2 && EPV == 8) {
                    typename Vec2B<TO>::v8 ob;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ob[e] = (TO)o[e];
                    *reinterpret_cast<typename Vec2B<TO>::v8*>(orow + EPV * j) = ob;
                } else if constexpr (sizeof(TO) == 2) {
                    typename Vec2B<TO>::v4 ob;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ob[e] = (TO)o[e];
                    *reinterpret_cast<typename Vec2B<TO>::v4*>(orow + EPV * j) = ob;
                } else {
#pragma unroll
                    for (int e = 0; e < EPV; e += 4)
                        *reinterpret_cast<float4*>(orow + EPV * j + e) = make_float4(o[e], o[e + 1], o[e + 2], o[e + 3]);
                }
            }
        }
    }
}

// fp16 rows whose vector count is not a multiple of 64 (D = 768 -> 96 vectors of 8 halfs) leave half of the lanes idle in the
// last load of the kernel above.  Here one wave takes TWO rows as one flat run of 2*nv vectors (192 = 3 x 64 for D = 768): every
// load and store instruction is full; the row statistics are two masked wave reductions.  Needs in_stride == D (dense rows).
template <typename TO, int NVT>
__global__ __launch_bounds__(256) void layernorm_f16_pair_kernel(const _Float16* __restrict__ x, TO* __restrict__ out,
                                                                 long long out_stride, const float* __restrict__ w,
                                                                 const float* __restrict__ b, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= rows) return;
    const int nv = D >> 3;
    const int total = (row0 + 1 < rows ? 2 : 1) * nv;
    const _Float16* xr = x + (long long)row0 * D;
    float v[NVT][8];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NVT; ++i) {
        const int k = i * 64 + lane;
        if (k < total) {
            const f16x8 hv = *reinterpret_cast<const f16x8*>(xr + 8 * k);
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[i][e] = (float)hv[e];
                t += v[i][e];
            }
            if (k < nv) s0 += t;
            else s1 += t;
        }
    }
    const float mean0 = wave_sum(s0) / (float)D, mean1 = wave_sum(s1) / (float)D;
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int i = 0; i < NVT; ++i) {
        const int k = i * 64 + lane;
        if (k < total) {
            const float m = k < nv ? mean0 : mean1;
            float t = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - m;
                t += d * d;
            }
            if (k < nv) q0 += t;
            else q1 += t;
        }
    }
    const float rstd0 = 1.0f / sqrtf(wave_sum(q0) / (float)D + eps), rstd1 = 1.0f / sqrtf(wave_sum(q1) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NVT; ++i) {
        const int k = i * 64 + lane;
        if (k < total) {
            const bool first = k < nv;
            const int c = (first ? k : k - nv) * 8;                         // column of the vector
            const float m = first ? mean0 : mean1, rs = first ? rstd0 : rstd1;
            const float4 w0 = *reinterpret_cast<const float4*>(w + c), w1 = *reinterpret_cast<const float4*>(w + c + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(b + c), b1 = *reinterpret_cast<const float4*>(b + c + 4);
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - m) * rs * wv[e] + bv[e];
            TO* op = out + (long long)(row0 + (first ? 0 : 1)) * out_stride + c;
            if constexpr (sizeof(TO) == 2) {
                typename Vec2B<TO>::v8 ob;
#pragma unroll
                for (int e = 0; e < 8; ++e) ob[e] = (TO)o[e];
                *reinterpret_cast<typename Vec2B<TO>::v8*>(op) = ob;
            } else {
                *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
        }
    }
}

template <typename TI, typename TO>
int launch_ln(const TI* x, long long in_stride, TO* out, long long out_stride, const float* w, const float* b, int rows,
              int D, float eps, hipStream_t s) {
    constexpr int EPV = 16 / (int)sizeof(TI);
    const int nvl = (D / EPV + 63) / 64;          // 16-byte vectors per lane
    if constexpr (sizeof(TI) == 2) {
        const int nv = D / 8;
        if (in_stride == D && nv % 64 != 0 && (2 * nv) % 64 == 0 && 2 * nv / 64 <= 6) {       // e.g. D = 768: 3 full loads per row pair
            const dim3 grid((unsigned)((rows + 7) / 8));
            switch (2 * nv / 64) {
                case 1: hipLaunchKernelGGL((layernorm_f16_pair_kernel<TO, 1>), grid, dim3(256), 0, s, x, out, out_stride, w, b, rows, D, eps); break;
                case 3: hipLaunchKernelGGL((layernorm_f16_pair_kernel<TO, 3>), grid, dim3(256), 0, s, x, out, out_stride, w, b, rows, D, eps); break;
                default: hipLaunchKernelGGL((layernorm_f16_pair_kernel<TO, 5>), grid, dim3(256), 0, s, x, out, out_stride, w, b, rows, D, eps); break;
            }
            return cfsar_check_launch("cfsar_layernorm");
        }
    }
#define CFSAR_LN(NV, RPW)                                                                                             \
    hipLaunchKernelGGL((layernorm_kernel<TI, TO, NV, RPW>), dim3((unsigned)((rows + 4 * RPW - 1) / (4 * RPW))), dim3(256), 0, \
                       s, x, in_stride, out, out_stride, w, b, rows, D, eps)
    constexpr int RM = 1;
    if (nvl <= 1) CFSAR_LN(1, 4 * RM);
    else if (nvl <= 2) CFSAR_LN(2, 2 * RM);
    else if (nvl <= 3) CFSAR_LN(3, 2);
    else if (nvl <= 4) CFSAR_LN(4, 2);
    else if (nvl <= 8) CFSAR_LN(8, 1);
    else CFSAR_LN(16, 1);
#undef CFSAR_LN
    return cfsar_check_launch("cfsar_layernorm");
}

}  // namespace

namespace {
// rowstats[m] = (mean, std, 1 / std, 0) from `slots` partial (sum, sum of squares) pairs per row; biased variance, eps inside the
// square root (few_shot.py:605-611 = F.layer_norm).  One thread per row; a row's partials are one contiguous 8 * slots bytes.
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ partial, float* __restrict__ rowstats,
                                                                int M, int slots, float invD, float eps) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float2* pp = reinterpret_cast<const float2*>(partial) + (size_t)m * slots;
    float s = 0.f, q = 0.f;
    for (int i = 0; i < slots; ++i) {
        const float2 v = pp[i];
        s += v.x;
        q += v.y;
    }
    const float mean = s * invD;
    const float var = fmaxf(q * invD - mean * mean, 0.f);
    const float sd = sqrtf(var + eps);
    *reinterpret_cast<float4*>(rowstats + (size_t)m * 4) = make_float4(mean, sd, 1.0f / sd, 0.f);
}

// Coalesced form for an even number of slots <= 16 (ViT-B: 12, ViT-L: 16): 8 lanes per row, lane q loads slots 2q, 2q + 1 as one
// 16-byte vector (a row's partials are read as whole lines instead of 64 scattered 8-byte pieces per load instruction: 95 -> ~15 us
// at 329 K rows x 16 slots), the 8 lanes are summed with DPP adds.
__global__ __launch_bounds__(256) void ln_stats_finalize8_kernel(const float* __restrict__ partial, float* __restrict__ rowstats,
                                                                 int M, int slots, float invD, float eps) {
    const int q = threadIdx.x & 7;
    int m = blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool rowok = m < M;
    m = rowok ? m : M - 1;
    float s = 0.f, sq = 0.f;
    if (2 * q < slots) {
        const float4 v = *reinterpret_cast<const float4*>(partial + ((size_t)m * slots + 2 * q) * 2);
        s = v.x + v.z;
        sq = v.y + v.w;
    }
    s = cfsar_dpp_sum8(s);
    sq = cfsar_dpp_sum8(sq);
    if (q == 0 && rowok) {
        const float mean = s * invD;
        const float var = fmaxf(sq * invD - mean * mean, 0.f);
        const float sd = sqrtf(var + eps);
        *reinterpret_cast<float4*>(rowstats + (size_t)m * 4) = make_float4(mean, sd, 1.0f / sd, 0.f);
    }
}

// the same statistics straight from fp16 rows (the output of ln_pre: the first LN-folded GEMM of the tower has no producer
// GEMM before it).  One wave per row, 16-byte loads.
__global__ __launch_bounds__(256) void row_stats_f16_kernel(const _Float16* __restrict__ x, float* __restrict__ rowstats, int M,
                                                            int D, long long ld, float eps) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const _Float16* xr = x + (size_t)m * ld;
    float s = 0.f, q = 0.f;
    for (int k = lane * 8; k < D; k += 512) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(xr + k);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            s += f;
            q = fmaf(f, f, q);
        }
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane == 0) {
        const float mean = s / (float)D;
        const float var = fmaxf(q / (float)D - mean * mean, 0.f);
        const float sd = sqrtf(var + eps);
        *reinterpret_cast<float4*>(rowstats + (size_t)m * 4) = make_float4(mean, sd, 1.0f / sd, 0.f);
    }
}
}  // namespace

extern "C" int cfsar_ln_stats_finalize(const float* partial, float* rowstats, int M, int slots, int D, float eps,
                                       cfsar_stream_t stream) {
    CFSAR_REQUIRE(partial && rowstats && M > 0 && slots > 0 && D > 0, "cfsar_ln_stats_finalize: bad argument");
    if (slots % 2 == 0 && slots <= 16)
        hipLaunchKernelGGL(ln_stats_finalize8_kernel, dim3((unsigned)((M + 31) / 32)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           partial, rowstats, M, slots, 1.0f / (float)D, eps);
    else
        hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           partial, rowstats, M, slots, 1.0f / (float)D, eps);
    return cfsar_check_launch("cfsar_ln_stats_finalize");
}

namespace {
// two-word fp16 stream -> fp32 (round 4: in front of ln_post on the class-token rows)
__global__ __launch_bounds__(256) void f16_pair_to_f32_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo,
                                                              float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)hi[i] + (float)lo[i];
}

// Round 6 (fp16_strict): everything between the patch-embed GEMM and the first block in ONE pass, with no 16-bit rounding before the two-word stream:
// row (f, t) = (t == 0 ? cls : tok[f (ntok - 1) + t - 1]) + pos[t]  (few_shot.py:675-676)  ->  ln_pre (:677, fp32, two-pass variance)  ->
// x_hi = fp16(y), x_lo = fp16(y - x_hi).  One wave per row, the row in registers (D <= 1 024: <= 4 float4 per lane).
__global__ __launch_bounds__(256) void embed_finish_pair_kernel(const float* __restrict__ tok, const float* __restrict__ cls,
                                                                const float* __restrict__ pos, const float* __restrict__ w,
                                                                const float* __restrict__ b, _Float16* __restrict__ xhi,
                                                                _Float16* __restrict__ xlo, long long rows, int ntok, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long f = row / ntok;
    const int t = (int)(row - f * ntok);
    const float* src = t == 0 ? cls : tok + (f * (ntok - 1) + (t - 1)) * (long long)D;
    const float* pr = pos + (long long)t * D;
    const int nv = D >> 2;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = i * 64 + lane;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < nv) {
            const float4 a = *reinterpret_cast<const float4*>(src + 4 * j), p4 = *reinterpret_cast<const float4*>(pr + 4 * j);
            v[i] = make_float4(a.x + p4.x, a.y + p4.y, a.z + p4.z, a.w + p4.w);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i * 64 + lane < nv) {
            const float a = v[i].x - mean, bq = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            ss += (a * a + bq * bq) + (c * c + d * d);
        }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = i * 64 + lane;
        if (j < nv) {
            const float4 w4 = *reinterpret_cast<const float4*>(w + 4 * j), b4 = *reinterpret_cast<const float4*>(b + 4 * j);
            const float o[4] = {(v[i].x - mean) * rstd * w4.x + b4.x, (v[i].y - mean) * rstd * w4.y + b4.y,
                                (v[i].z - mean) * rstd * w4.z + b4.z, (v[i].w - mean) * rstd * w4.w + b4.w};
            h4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (_Float16)o[e];
                l[e] = (_Float16)(o[e] - (float)h[e]);
            }
            *reinterpret_cast<h4*>(xhi + row * D + 4 * j) = h;
            *reinterpret_cast<h4*>(xlo + row * D + 4 * j) = l;
        }
    }
}

// strided row gather in 4-byte words: one workgroup per row
__global__ __launch_bounds__(256) void copy_rows_strided_kernel(const char* __restrict__ src, long long src_stride, char* __restrict__ dst,
                                                                long long dst_stride, int words) {
    const unsigned* s = reinterpret_cast<const unsigned*>(src + (long long)blockIdx.x * src_stride);
    unsigned* d = reinterpret_cast<unsigned*>(dst + (long long)blockIdx.x * dst_stride);
    for (int i = threadIdx.x; i < words; i += 256) d[i] = s[i];
}
}  // namespace

namespace {
// out[f][k] = (1 / tokens) sum_t w_t (A[f tokens + t][k] - mu_t), (mu_t, w_t) = (mean, 1 / std) of row t from rowstats [M, 4], or
// (0, 1) without them: the per-frame TOKEN MEAN of a GEMM's activation operand (of LayerNorm(x) without materialising it, for the
// LN-folded GEMMs).  One workgroup per (256-column chunk, frame): 32 lanes x 16 bytes across, 8 row groups down, fixed summation order.
__global__ __launch_bounds__(256) void frame_col_means_kernel(const _Float16* __restrict__ A, long long lda, const float* __restrict__ rowstats,
                                                              __bf16* __restrict__ out, int tokens, int K) {
    __shared__ float red[8][256];
    const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int col = blockIdx.x * 256 + cg * 8;
    const long long row0 = (long long)blockIdx.y * tokens;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < K) {
        for (int t = rg; t < tokens; t += 8) {
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            const h8 a = *reinterpret_cast<const h8*>(A + (row0 + t) * lda + col);
            float mu = 0.f, w = 1.f;
            if (rowstats) {
                mu = rowstats[(row0 + t) * 4];
                w = rowstats[(row0 + t) * 4 + 2];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w * ((float)a[j] - mu);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rg][cg * 8 + j] = acc[j];
    __syncthreads();
    const int c = threadIdx.x;
    if (blockIdx.x * 256 + c < K) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += red[r][c];
        out[(long long)blockIdx.y * K + blockIdx.x * 256 + c] = (__bf16)(s / (float)tokens);
    }
}
}  // namespace

extern "C" int cfsar_frame_col_means(const void* A, int lda, const float* rowstats, void* out, int frames, int tokens, int K,
                                     cfsar_stream_t stream) {
    CFSAR_REQUIRE(A && out && frames > 0 && tokens > 0 && K > 0 && K % 8 == 0 && lda >= K && lda % 8 == 0, "cfsar_frame_col_means: bad argument");
    hipLaunchKernelGGL(frame_col_means_kernel, dim3((unsigned)((K + 255) / 256), (unsigned)frames), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const _Float16*>(A), (long long)lda, rowstats, static_cast<__bf16*>(out), tokens, K);
    return cfsar_check_launch("cfsar_frame_col_means");
}

extern "C" int cfsar_f16_pair_to_f32(const void* hi, const void* lo, float* out, int64_t n, cfsar_stream_t stream) {
    CFSAR_REQUIRE(hi && lo && out && n > 0, "cfsar_f16_pair_to_f32: bad argument");
    hipLaunchKernelGGL(f16_pair_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const _Float16*>(hi), static_cast<const _Float16*>(lo), out, (long long)n);
    return cfsar_check_launch("cfsar_f16_pair_to_f32");
}

extern "C" int cfsar_embed_finish_pair(const float* tok, const float* cls, const float* pos, const float* ln_w, const float* ln_b, void* x_hi,
                                       void* x_lo, int F, int ntok, int D, float eps, cfsar_stream_t stream) {
    CFSAR_REQUIRE(tok && cls && pos && ln_w && ln_b && x_hi && x_lo, "cfsar_embed_finish_pair: null pointer");
    CFSAR_REQUIRE(F > 0 && ntok > 1 && D > 0 && D % 4 == 0 && D <= 1024, "cfsar_embed_finish_pair: bad shape F=%d ntok=%d D=%d (D %% 4 == 0, D <= 1024)", F, ntok, D);
    const long long rows = (long long)F * ntok;
    hipLaunchKernelGGL(embed_finish_pair_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), tok, cls, pos,
                       ln_w, ln_b, static_cast<_Float16*>(x_hi), static_cast<_Float16*>(x_lo), rows, ntok, D, eps);
    return cfsar_check_launch("cfsar_embed_finish_pair");
}

extern "C" int cfsar_copy_rows_strided(const void* src, int64_t src_stride, void* dst, int64_t dst_stride, int rows, int row_bytes,
                                       cfsar_stream_t stream) {
    CFSAR_REQUIRE(src && dst && rows > 0 && row_bytes > 0 && row_bytes % 4 == 0 && src_stride % 4 == 0 && dst_stride % 4 == 0,
                  "cfsar_copy_rows_strided: bad argument (rows=%d row_bytes=%d)", rows, row_bytes);
    hipLaunchKernelGGL(copy_rows_strided_kernel, dim3((unsigned)rows), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const char*>(src), (long long)src_stride, static_cast<char*>(dst), (long long)dst_stride, row_bytes / 4);
    return cfsar_check_launch("cfsar_copy_rows_strided");
}

extern "C" int cfsar_row_stats(const void* x, float* rowstats, int M, int D, int ld, float eps, cfsar_stream_t stream) {
    CFSAR_REQUIRE(x && rowstats && M > 0 && D > 0 && D % 8 == 0 && ld >= D && ld % 8 == 0, "cfsar_row_stats: bad argument");
    hipLaunchKernelGGL(row_stats_f16_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const _Float16*>(x), rowstats, M, D, (long long)ld, eps);
    return cfsar_check_launch("cfsar_row_stats");
}

extern "C" int cfsar_im2col_patches(const float* frames, void* out, int out_dtype, int F, int H, int W, int P,
                                    int k_pad, cfsar_stream_t stream) {
    CFSAR_REQUIRE(frames && out, "cfsar_im2col_patches: null pointer");
    CFSAR_REQUIRE(F > 0 && P > 0 && P % 2 == 0 && H % P == 0 && W % P == 0, "cfsar_im2col_patches: bad geometry");
    CFSAR_REQUIRE(k_pad >= 3 * P * P && k_pad % 2 == 0, "cfsar_im2col_patches: k_pad too small / odd");
    const long long rows = (long long)F * (H / P) * (W / P);
    const long long pairs = rows * (k_pad / 2);
    long long blocks = (pairs + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((out_dtype == CFSAR_BF16 || out_dtype == CFSAR_F16) && P == 16 && k_pad == 768 && W % 16 == 0 && (long long)F * 3 <= 65535) {
        const int npatch = (H / 16) * (W / 16);
        const dim3 grid((unsigned)((npatch + 15) / 16), (unsigned)(F * 3));
        if (out_dtype == CFSAR_BF16)
            hipLaunchKernelGGL(im2col_p16_bf16_kernel<__bf16>, grid, dim3(256), 0, s, frames, static_cast<__bf16*>(out), H, W, k_pad);
        else
            hipLaunchKernelGGL(im2col_p16_bf16_kernel<_Float16>, grid, dim3(256), 0, s, frames, static_cast<_Float16*>(out), H, W, k_pad);
        return cfsar_check_launch("cfsar_im2col_patches");
    }
    if (out_dtype == CFSAR_F16)
        hipLaunchKernelGGL((im2col_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, s, frames,
                           static_cast<_Float16*>(out), F, H, W, P, k_pad, pairs);
    else if (out_dtype == CFSAR_BF16)
        hipLaunchKernelGGL((im2col_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, s, frames,
                           static_cast<__bf16*>(out), F, H, W, P, k_pad, pairs);
    else if (out_dtype == CFSAR_F32)
        hipLaunchKernelGGL((im2col_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, frames,
                           static_cast<float*>(out), F, H, W, P, k_pad, pairs);
    else
        return cfsar_fail("cfsar_im2col_patches: bad dtype %d", out_dtype);
    return cfsar_check_launch("cfsar_im2col_patches");
}

extern "C" int cfsar_im2col_patches_split(const float* frames, void* out, int F, int H, int W, int P, int k_pad, cfsar_stream_t stream) {
    CFSAR_REQUIRE(frames && out, "cfsar_im2col_patches_split: null pointer");
    CFSAR_REQUIRE(F > 0 && P > 0 && P % 2 == 0 && H % P == 0 && W % P == 0, "cfsar_im2col_patches_split: bad geometry");
    CFSAR_REQUIRE(k_pad >= 3 * P * P && k_pad % 2 == 0, "cfsar_im2col_patches_split: k_pad too small / odd");
    const long long rows = (long long)F * (H / P) * (W / P);
    const long long pairs = rows * (k_pad / 2);
    long long blocks = (pairs + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(im2col_split_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), frames,
                       static_cast<_Float16*>(out), F, H, W, P, k_pad, pairs);
    return cfsar_check_launch("cfsar_im2col_patches_split");
}

extern "C" int cfsar_cls_rows(float* x, const float* cls, const float* pos, int F, int ntok, int D,
                              cfsar_stream_t stream) {
    CFSAR_REQUIRE(x && cls && pos && F > 0 && ntok > 0 && D > 0, "cfsar_cls_rows: bad arguments");
    const long long total = (long long)F * D;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(cls_rows_kernel<float>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, cls, pos, F,
                       ntok, D);
    return cfsar_check_launch("cfsar_cls_rows");
}

extern "C" int cfsar_cls_rows_ex(void* x, int x_dtype, const float* cls, const float* pos, int F, int ntok, int D,
                                 cfsar_stream_t stream) {
    CFSAR_REQUIRE(x && cls && pos && F > 0 && ntok > 0 && D > 0, "cfsar_cls_rows_ex: bad arguments");
    const long long total = (long long)F * D;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (x_dtype == CFSAR_F32)
        hipLaunchKernelGGL(cls_rows_kernel<float>, dim3(blocks), dim3(256), 0, s, static_cast<float*>(x), cls, pos, F, ntok, D);
    else if (x_dtype == CFSAR_F16)
        hipLaunchKernelGGL(cls_rows_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, static_cast<_Float16*>(x), cls, pos, F, ntok, D);
    else
        return cfsar_fail("cfsar_cls_rows_ex: bad dtype %d", x_dtype);
    return cfsar_check_launch("cfsar_cls_rows_ex");
}

extern "C" int cfsar_preprocess_frames(const uint8_t* frames, float* out, int T, int H, int W, int scale_h, int scale_w,
                                       int crop, int y0, int x0, const float* mean3, const float* std3,
                                       cfsar_stream_t stream) {
    CFSAR_REQUIRE(frames && out && mean3 && std3, "cfsar_preprocess_frames: null pointer");
    CFSAR_REQUIRE(T > 0 && H > 1 && W > 1 && scale_h >= crop && scale_w >= crop && crop > 0, "cfsar_preprocess_frames: bad geometry");
    CFSAR_REQUIRE(y0 >= 0 && x0 >= 0 && y0 + crop <= scale_h && x0 + crop <= scale_w, "cfsar_preprocess_frames: crop window out of range");
    const long long total = (long long)T * crop * crop;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), frames, out, T, H,
                       W, scale_h, scale_w, crop, y0, x0, mean3[0], mean3[1], mean3[2], 1.0f / std3[0], 1.0f / std3[1],
                       1.0f / std3[2]);
    return cfsar_check_launch("cfsar_preprocess_frames");
}

extern "C" int cfsar_layernorm_ex(const void* x, int in_dtype, int64_t in_stride, void* out, int64_t out_stride, int out_dtype,
                                  const float* weight, const float* bias, int rows, int D, float eps, cfsar_stream_t stream) {
    CFSAR_REQUIRE(x && out && weight && bias, "cfsar_layernorm: null pointer");
    CFSAR_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 4 * 64 * LN_MAXV, "cfsar_layernorm: bad D=%d", D);
    CFSAR_REQUIRE(in_stride % 4 == 0 && out_stride % 4 == 0, "cfsar_layernorm: strides must be multiples of 4");
    CFSAR_REQUIRE(in_dtype != CFSAR_F16 || (D % 8 == 0 && in_stride % 8 == 0 && out_stride % 8 == 0), "cfsar_layernorm: fp16 input needs D and strides to be multiples of 8");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long is = in_stride, os = out_stride;
    if (in_dtype == CFSAR_F32) {
        const float* xi = static_cast<const float*>(x);
        if (out_dtype == CFSAR_BF16) return launch_ln<float, __bf16>(xi, is, static_cast<__bf16*>(out), os, weight, bias, rows, D, eps, s);
        if (out_dtype == CFSAR_F32) return launch_ln<float, float>(xi, is, static_cast<float*>(out), os, weight, bias, rows, D, eps, s);
    } else if (in_dtype == CFSAR_F16) {
        const _Float16* xi = static_cast<const _Float16*>(x);
        if (out_dtype == CFSAR_BF16) return launch_ln<_Float16, __bf16>(xi, is, static_cast<__bf16*>(out), os, weight, bias, rows, D, eps, s);
        if (out_dtype == CFSAR_F16) return launch_ln<_Float16, _Float16>(xi, is, static_cast<_Float16*>(out), os, weight, bias, rows, D, eps, s);
        if (out_dtype == CFSAR_F32) return launch_ln<_Float16, float>(xi, is, static_cast<float*>(out), os, weight, bias, rows, D, eps, s);
    }
    return cfsar_fail("cfsar_layernorm: unsupported dtype pair in=%d out=%d", in_dtype, out_dtype);
}

extern "C" int cfsar_layernorm(const float* x, int64_t in_stride, void* out, int64_t out_stride, int out_dtype,
                               const float* weight, const float* bias, int rows, int D, float eps,
                               cfsar_stream_t stream) {
    return cfsar_layernorm_ex(x, CFSAR_F32, in_stride, out, out_stride, out_dtype, weight, bias, rows, D, eps, stream);
}
