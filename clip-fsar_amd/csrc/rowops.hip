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
            bf16x2 o;
            o[0] = (__bf16)v.x;
            o[1] = (__bf16)v.y;
            *reinterpret_cast<bf16x2*>(out + row * k_pad + k) = o;
        } else {
            *reinterpret_cast<float2*>(out + row * k_pad + k) = v;
        }
    }
}

__global__ __launch_bounds__(256) void cls_rows_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, int F, int ntok, int D) {
    const long long total = (long long)F * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long f = i / D;
        const int d = (int)(i - f * D);
        x[f * (long long)ntok * D + d] = cls[d] + pos[d];
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
template <typename TO, int NV, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long long in_stride,
                                                        TO* __restrict__ out, long long out_stride,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int nv = D >> 2;   // float4 count
    float4 v[RPW][NV];
    float s[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r < rows ? row0 + r : rows - 1;
        const float* xr = x + (long long)row * in_stride;
        s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int j = i * 64 + lane;
            if (j < nv) v[r][i] = *reinterpret_cast<const float4*>(xr + 4 * j);
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i * 64 + lane < nv) s[r] += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
    }
    float4 wv[NV], bv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = i * 64 + lane;
        if (j < nv) {
            wv[i] = *reinterpret_cast<const float4*>(w + 4 * j);
            bv[i] = *reinterpret_cast<const float4*>(b + 4 * j);
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const float mean = wave_sum(s[r]) / (float)D;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (i * 64 + lane < nv) {
                const float a = v[r][i].x - mean, bq = v[r][i].y - mean, c = v[r][i].z - mean, d = v[r][i].w - mean;
                ss += (a * a + bq * bq) + (c * c + d * d);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
        if (row0 + r >= rows) continue;
        TO* orow = out + (long long)(row0 + r) * out_stride;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int j = i * 64 + lane;
            if (j < nv) {
                float4 o;
                o.x = (v[r][i].x - mean) * rstd * wv[i].x + bv[i].x;
                o.y = (v[r][i].y - mean) * rstd * wv[i].y + bv[i].y;
                o.z = (v[r][i].z - mean) * rstd * wv[i].z + bv[i].z;
                o.w = (v[r][i].w - mean) * rstd * wv[i].w + bv[i].w;
                if constexpr (sizeof(TO) == 2) {
                    bf16x4 ob;
                    ob[0] = (__bf16)o.x; ob[1] = (__bf16)o.y; ob[2] = (__bf16)o.z; ob[3] = (__bf16)o.w;
                    *reinterpret_cast<bf16x4*>(orow + 4 * j) = ob;
                } else {
                    *reinterpret_cast<float4*>(orow + 4 * j) = o;
                }
            }
        }
    }
}

template <typename TO>
int launch_ln(const float* x, long long in_stride, TO* out, long long out_stride, const float* w, const float* b, int rows,
              int D, float eps, hipStream_t s) {
    const int nvl = (D / 4 + 63) / 64;          // float4 per lane
#define CFSAR_LN(NV, RPW)                                                                                             \
    hipLaunchKernelGGL((layernorm_kernel<TO, NV, RPW>), dim3((unsigned)((rows + 4 * RPW - 1) / (4 * RPW))), dim3(256), 0, \
                       s, x, in_stride, out, out_stride, w, b, rows, D, eps)
    if (nvl <= 1) CFSAR_LN(1, 4);
    else if (nvl <= 2) CFSAR_LN(2, 2);
    else if (nvl <= 3) CFSAR_LN(3, 2);
    else if (nvl <= 4) CFSAR_LN(4, 2);
    else if (nvl <= 8) CFSAR_LN(8, 1);
    else CFSAR_LN(16, 1);
#undef CFSAR_LN
    return cfsar_check_launch("cfsar_layernorm");
}

}  // namespace

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
    if (out_dtype == CFSAR_BF16)
        hipLaunchKernelGGL((im2col_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, s, frames,
                           static_cast<__bf16*>(out), F, H, W, P, k_pad, pairs);
    else if (out_dtype == CFSAR_F32)
        hipLaunchKernelGGL((im2col_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, frames,
                           static_cast<float*>(out), F, H, W, P, k_pad, pairs);
    else
        return cfsar_fail("cfsar_im2col_patches: bad dtype %d", out_dtype);
    return cfsar_check_launch("cfsar_im2col_patches");
}

extern "C" int cfsar_cls_rows(float* x, const float* cls, const float* pos, int F, int ntok, int D,
                              cfsar_stream_t stream) {
    CFSAR_REQUIRE(x && cls && pos && F > 0 && ntok > 0 && D > 0, "cfsar_cls_rows: bad arguments");
    const long long total = (long long)F * D;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(cls_rows_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, cls, pos, F,
                       ntok, D);
    return cfsar_check_launch("cfsar_cls_rows");
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

extern "C" int cfsar_layernorm(const float* x, int64_t in_stride, void* out, int64_t out_stride, int out_dtype,
                               const float* weight, const float* bias, int rows, int D, float eps,
                               cfsar_stream_t stream) {
    CFSAR_REQUIRE(x && out && weight && bias, "cfsar_layernorm: null pointer");
    CFSAR_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 4 * 64 * LN_MAXV, "cfsar_layernorm: bad D=%d", D);
    CFSAR_REQUIRE(in_stride % 4 == 0 && out_stride % 4 == 0, "cfsar_layernorm: strides must be multiples of 4");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_dtype == CFSAR_BF16)
        return launch_ln<__bf16>(x, (long long)in_stride, static_cast<__bf16*>(out), (long long)out_stride, weight, bias,
                                 rows, D, eps, s);
    if (out_dtype == CFSAR_F32)
        return launch_ln<float>(x, (long long)in_stride, static_cast<float*>(out), (long long)out_stride, weight, bias, rows,
                                D, eps, s);
    return cfsar_fail("cfsar_layernorm: bad dtype %d", out_dtype);
}
