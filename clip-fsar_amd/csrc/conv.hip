// N3: data-movement kernels of the CLIP ModifiedResNet ("RN50") tower (reference few_shot.py:182-227, 542-602).
// Activations are NHWC ([F*H*W, C] row-major) so 1x1 convolutions are plain GEMM rows and 3x3 convolutions become a
// GEMM after a tap-major gather (column = (ky*3+kx)*C + c).  BatchNorm (eval) is folded into the conv weights / bias on
// the host, ReLU and the identity add are GEMM epilogues (cfsar_gemm_ex); what remains here is memory-bound.
#include <utility>

#include "common.h"

namespace {

// frames [F,3,H,W] f32 (the episode layout) -> [F,H,W,4-padded? no: 3] in the compute dtype
template <typename TO>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, TO* __restrict__ out, int C, int H,
                                                           int W, long long total) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        long long r = idx / C;
        const int x = (int)(r % W);
        r /= W;
        const int y = (int)(r % H);
        const long long f = r / H;
        out[idx] = (TO)in[((f * C + c) * H + y) * (long long)W + x];
    }
}

// 3x3 / pad 1 / stride s gather: in [F,H,W,C] -> out [F*Ho*Wo, kpad], column (ky*3+kx)*C + c, zero outside the image and in
// the pad columns.  VEC elements (16 bytes when possible) per thread.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W, int C,
                                                        int Ho, int Wo, int stride, int kpad, long long total_vec) {
    const int kv = kpad / VEC;
    const int kreal = 9 * C;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_vec;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long row = idx / kv;
        const int k = (int)(idx - row * kv) * VEC;
        T v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = (T)0.0f;
        if (k < kreal) {
            const int tap = k / C, c = k - tap * C;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int xo = (int)(row % Wo);
            const long long t = row / Wo;
            const int yo = (int)(t % Ho);
            const long long f = t / Ho;
            const int y = yo * stride + ky - 1, x = xo * stride + kx - 1;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                const T* src = in + ((f * H + y) * (long long)W + x) * C + c;
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[j] = src[j];
            }
        }
        T* dst = out + row * kpad + k;
#pragma unroll
        for (int j = 0; j < VEC; ++j) dst[j] = v[j];
    }
}

// AvgPool2d(2) on NHWC
template <typename T>
__global__ __launch_bounds__(256) void avgpool2_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W, int C,
                                                       long long total) {
    const int Ho = H / 2, Wo = W / 2;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        long long r = idx / C;
        const int xo = (int)(r % Wo);
        r /= Wo;
        const int yo = (int)(r % Ho);
        const long long f = r / Ho;
        const T* p = in + ((f * H + 2 * yo) * (long long)W + 2 * xo) * C + c;
        const float s = ((float)p[0] + (float)p[C]) + ((float)p[(long long)W * C] + (float)p[(long long)W * C + C]);
        out[idx] = (T)(s * 0.25f);
    }
}

// bf16, C % 8 == 0: 8 channels (16 bytes) per thread -- four 16-byte loads, one 16-byte store (the scalar form above moves 2 bytes per
// load: 3.5 TB/s on the RN50 pools; this one streams)
template <typename T>
__global__ __launch_bounds__(256) void avgpool2_x8_kernel(const T* __restrict__ in, T* __restrict__ out, int H, int W, int C8,
                                                              long long total8) {
    typedef typename Vec2B<T>::v8 T8;
    const int Ho = H / 2, Wo = W / 2;
    const long long rowb = (long long)W * C8;                 // one input image row in 16-byte units
    const uint4* in4 = reinterpret_cast<const uint4*>(in);
    uint4* out4 = reinterpret_cast<uint4*>(out);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total8;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C8);
        long long r = idx / C8;
        const int xo = (int)(r % Wo);
        r /= Wo;
        const int yo = (int)(r % Ho);
        const long long f = r / Ho;
        const uint4* q = in4 + ((f * H + 2 * yo) * (long long)W + 2 * xo) * C8 + c;
        const T8 a = __builtin_bit_cast(T8, q[0]), b = __builtin_bit_cast(T8, q[C8]);
        const T8 d = __builtin_bit_cast(T8, q[rowb]), e = __builtin_bit_cast(T8, q[rowb + C8]);
        T8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (T)((((float)a[j] + (float)b[j]) + ((float)d[j] + (float)e[j])) * 0.25f);
        out4[idx] = __builtin_bit_cast(uint4, o);
    }
}

// AttentionPool2d token build (few_shot.py:446-448): tokens[f,0] = mean_hw(x[f]) + pos[0]; tokens[f,1+p] = x[f,p] + pos[1+p]
template <typename T>
__global__ __launch_bounds__(256) void attnpool_tokens_kernel(const T* __restrict__ x, const float* __restrict__ pos,
                                                              T* __restrict__ out, int HW, int C) {
    const int f = blockIdx.x;
    const T* xf = x + (long long)f * HW * C;
    T* of = out + (long long)f * (HW + 1) * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int p = 0; p < HW; ++p) {
            const float v = (float)xf[(long long)p * C + c];
            s += v;
            of[(long long)(p + 1) * C + c] = (T)(v + pos[(long long)(p + 1) * C + c]);
        }
        of[c] = (T)(s / (float)HW + pos[c]);
    }
}


// AttentionPool2d attention for the ONE query the pool keeps (the mean token, few_shot.py:450-469: multi_head_attention_forward
// over [mean ; x], output row 0): one wave per (frame, head).  scores_t = scale * q . k_t (lane = token, chunks of 64),
// softmax over the T tokens (fp32, wave reductions), out_d = sum_t p_t v_td (lane = d).  q [F, C], kv [F*T, 2C] = [k | v].
__global__ __launch_bounds__(64) void attnpool_attend_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                             float* __restrict__ out, int T, int heads, int hd, float scale) {
    __shared__ float sc[512];
    __shared__ float qs[128];
    const int f = blockIdx.x / heads, h = blockIdx.x % heads;
    const int lane = threadIdx.x;
    const int C = heads * hd;
    for (int d = lane; d < hd; d += 64) qs[d] = q[(long long)f * C + h * hd + d] * scale;
    __syncthreads();
    float mx = -INFINITY;
    for (int t = lane; t < T; t += 64) {
        const float* kr = kv + ((long long)f * T + t) * 2 * C + h * hd;
        float s = 0.f;
        for (int d = 0; d < hd; d += 4) {
            const float4 k4 = *reinterpret_cast<const float4*>(kr + d);
            s += qs[d] * k4.x + qs[d + 1] * k4.y + qs[d + 2] * k4.z + qs[d + 3] * k4.w;
        }
        sc[t] = s;
        mx = fmaxf(mx, s);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int t = lane; t < T; t += 64) {
        const float e = __expf(sc[t] - mx);
        sc[t] = e;
        sum += e;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    __syncthreads();
    const float inv = 1.0f / sum;
    for (int d = lane; d < hd; d += 64) {
        const float* vr = kv + (long long)f * T * 2 * C + C + h * hd + d;
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += sc[t] * vr[(long long)t * 2 * C];
        out[(long long)f * C + h * hd + d] = a * inv;
    }
}

// Stem conv1 of the ModifiedResNet (few_shot.py:558-560, 582-586): nn.Conv2d(3, Cout, 3, stride 2, padding 1) + folded
// BatchNorm + ReLU, straight from the fp32 NCHW frames to NHWC activations -- no layout pass, no im2col matrix (K = 27 is
// far too short for the matrix cores; the layer is HBM-bound: 385 MB in, 514 MB out at 640 frames).  One thread per output
// pixel: 27 inputs in registers, weights [27][COUT] broadcast from LDS, fp32 FMAs, COUT outputs stored as 16-byte vectors.
template <typename TO, int COUT>
__global__ __launch_bounds__(256) void stem_conv1_kernel(const float* __restrict__ frames, const float* __restrict__ w,
                                                         const float* __restrict__ bias, TO* __restrict__ out, int H, int W,
                                                         int Ho, int Wo, long long npix, int relu) {
    __shared__ float sw[27 * COUT];
    __shared__ float sb[COUT];
    // w is [COUT][3][3][3] (PyTorch layout: o, c, ky, kx) -> sw[(c*9 + ky*3 + kx) * COUT + o]
    for (int i = threadIdx.x; i < 27 * COUT; i += 256) sw[(i % 27) * COUT + i / 27] = w[i];
    for (int i = threadIdx.x; i < COUT; i += 256) sb[i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    const int xo = (int)(pix % Wo);
    const long long t = pix / Wo;
    const int yo = (int)(t % Ho);
    const long long f = t / Ho;
    float in[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int yy = 2 * yo + ky - 1, xx = 2 * xo + kx - 1;
                float v = 0.f;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = frames[((f * 3 + c) * H + yy) * (long long)W + xx];
                in[c * 9 + ky * 3 + kx] = v;
            }
    TO* op = out + pix * COUT;
#pragma unroll
    for (int o0 = 0; o0 < COUT; o0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = sb[o0 + j];
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const float4 w0 = *reinterpret_cast<const float4*>(&sw[k * COUT + o0]);
            const float4 w1 = *reinterpret_cast<const float4*>(&sw[k * COUT + o0 + 4]);
            acc[0] = fmaf(in[k], w0.x, acc[0]); acc[1] = fmaf(in[k], w0.y, acc[1]);
            acc[2] = fmaf(in[k], w0.z, acc[2]); acc[3] = fmaf(in[k], w0.w, acc[3]);
            acc[4] = fmaf(in[k], w1.x, acc[4]); acc[5] = fmaf(in[k], w1.y, acc[5]);
            acc[6] = fmaf(in[k], w1.z, acc[6]); acc[7] = fmaf(in[k], w1.w, acc[7]);
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
        }
        if constexpr (sizeof(TO) == 2) {
            typename Vec2B<TO>::v8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (TO)acc[j];
            *reinterpret_cast<typename Vec2B<TO>::v8*>(op + o0) = o;
        } else {
            *reinterpret_cast<float4*>(op + o0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4*>(op + o0 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    }
}

template <typename F>
int grid_for(long long total, F) {
    long long b = (total + 255) / 256;
    return (int)(b > 256 * 32 ? 256 * 32 : b);
}

}  // namespace

extern "C" int cfsar_nchw_to_nhwc(const float* frames, void* out, int out_dtype, int F, int C, int H, int W,
                                  cfsar_stream_t stream) {
    CFSAR_REQUIRE(frames && out && F > 0 && C > 0 && H > 0 && W > 0, "cfsar_nchw_to_nhwc: bad arguments");
    const long long total = (long long)F * C * H * W;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_dtype == CFSAR_BF16)
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<__bf16>), dim3(grid_for(total, 0)), dim3(256), 0, s, frames, static_cast<__bf16*>(out), C, H, W, total);
    else if (out_dtype == CFSAR_F16)
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<_Float16>), dim3(grid_for(total, 0)), dim3(256), 0, s, frames, static_cast<_Float16*>(out), C, H, W, total);
    else if (out_dtype == CFSAR_F32)
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), dim3(grid_for(total, 0)), dim3(256), 0, s, frames, static_cast<float*>(out), C, H, W, total);
    else
        return cfsar_fail("cfsar_nchw_to_nhwc: bad dtype %d", out_dtype);
    return cfsar_check_launch("cfsar_nchw_to_nhwc");
}

extern "C" int cfsar_im2col3x3_nhwc(const void* in, void* out, int dtype, int F, int H, int W, int C, int stride, int k_pad,
                                    cfsar_stream_t stream) {
    CFSAR_REQUIRE(in && out && F > 0 && H > 0 && W > 0 && C > 0, "cfsar_im2col3x3_nhwc: bad arguments");
    CFSAR_REQUIRE(stride == 1 || stride == 2, "cfsar_im2col3x3_nhwc: stride must be 1 or 2");
    CFSAR_REQUIRE(k_pad >= 9 * C, "cfsar_im2col3x3_nhwc: k_pad too small");
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long long rows = (long long)F * Ho * Wo;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec8 = (C % 8 == 0) && (k_pad % 8 == 0);
    const bool vec4 = (C % 4 == 0) && (k_pad % 4 == 0);
    if (dtype == CFSAR_BF16 || dtype == CFSAR_F16) {           // a pure mover (zeros are the same bits): fp16 takes the 2-byte instances
        if (vec8) {
            const long long tv = rows * (k_pad / 8);
            hipLaunchKernelGGL((im2col3x3_kernel<__bf16, 8>), dim3(grid_for(tv, 0)), dim3(256), 0, s, static_cast<const __bf16*>(in), static_cast<__bf16*>(out), H, W, C, Ho, Wo, stride, k_pad, tv);
        } else {
            const long long tv = rows * k_pad;
            hipLaunchKernelGGL((im2col3x3_kernel<__bf16, 1>), dim3(grid_for(tv, 0)), dim3(256), 0, s, static_cast<const __bf16*>(in), static_cast<__bf16*>(out), H, W, C, Ho, Wo, stride, k_pad, tv);
        }
    } else if (dtype == CFSAR_F32) {
        if (vec4) {
            const long long tv = rows * (k_pad / 4);
            hipLaunchKernelGGL((im2col3x3_kernel<float, 4>), dim3(grid_for(tv, 0)), dim3(256), 0, s, static_cast<const float*>(in), static_cast<float*>(out), H, W, C, Ho, Wo, stride, k_pad, tv);
        } else {
            const long long tv = rows * k_pad;
            hipLaunchKernelGGL((im2col3x3_kernel<float, 1>), dim3(grid_for(tv, 0)), dim3(256), 0, s, static_cast<const float*>(in), static_cast<float*>(out), H, W, C, Ho, Wo, stride, k_pad, tv);
        }
    } else {
        return cfsar_fail("cfsar_im2col3x3_nhwc: bad dtype %d", dtype);
    }
    return cfsar_check_launch("cfsar_im2col3x3_nhwc");
}

extern "C" int cfsar_avgpool2x2_nhwc(const void* in, void* out, int dtype, int F, int H, int W, int C, cfsar_stream_t stream) {
    CFSAR_REQUIRE(in && out && F > 0 && H > 1 && W > 1 && C > 0, "cfsar_avgpool2x2_nhwc: bad arguments");
    const long long total = (long long)F * (H / 2) * (W / 2) * C;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == CFSAR_BF16 && C % 8 == 0 && ((size_t)in & 15) == 0 && ((size_t)out & 15) == 0)
        hipLaunchKernelGGL(avgpool2_x8_kernel<__bf16>, dim3(grid_for(total / 8, 0)), dim3(256), 0, s, static_cast<const __bf16*>(in), static_cast<__bf16*>(out), H, W, C / 8, total / 8);
    else if (dtype == CFSAR_F16 && C % 8 == 0 && ((size_t)in & 15) == 0 && ((size_t)out & 15) == 0)
        hipLaunchKernelGGL(avgpool2_x8_kernel<_Float16>, dim3(grid_for(total / 8, 0)), dim3(256), 0, s, static_cast<const _Float16*>(in), static_cast<_Float16*>(out), H, W, C / 8, total / 8);
    else if (dtype == CFSAR_F16)
        hipLaunchKernelGGL((avgpool2_kernel<_Float16>), dim3(grid_for(total, 0)), dim3(256), 0, s, static_cast<const _Float16*>(in), static_cast<_Float16*>(out), H, W, C, total);
    else if (dtype == CFSAR_BF16)
        hipLaunchKernelGGL((avgpool2_kernel<__bf16>), dim3(grid_for(total, 0)), dim3(256), 0, s, static_cast<const __bf16*>(in), static_cast<__bf16*>(out), H, W, C, total);
    else if (dtype == CFSAR_F32)
        hipLaunchKernelGGL((avgpool2_kernel<float>), dim3(grid_for(total, 0)), dim3(256), 0, s, static_cast<const float*>(in), static_cast<float*>(out), H, W, C, total);
    else
        return cfsar_fail("cfsar_avgpool2x2_nhwc: bad dtype %d", dtype);
    return cfsar_check_launch("cfsar_avgpool2x2_nhwc");
}

extern "C" int cfsar_attnpool_tokens(const void* x, const float* pos, void* out, int dtype, int F, int HW, int C,
                                     cfsar_stream_t stream) {
    CFSAR_REQUIRE(x && pos && out && F > 0 && HW > 0 && C > 0, "cfsar_attnpool_tokens: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == CFSAR_BF16)
        hipLaunchKernelGGL((attnpool_tokens_kernel<__bf16>), dim3(F), dim3(256), 0, s, static_cast<const __bf16*>(x), pos, static_cast<__bf16*>(out), HW, C);
    else if (dtype == CFSAR_F16)
        hipLaunchKernelGGL((attnpool_tokens_kernel<_Float16>), dim3(F), dim3(256), 0, s, static_cast<const _Float16*>(x), pos, static_cast<_Float16*>(out), HW, C);
    else if (dtype == CFSAR_F32)
        hipLaunchKernelGGL((attnpool_tokens_kernel<float>), dim3(F), dim3(256), 0, s, static_cast<const float*>(x), pos, static_cast<float*>(out), HW, C);
    else
        return cfsar_fail("cfsar_attnpool_tokens: bad dtype %d", dtype);
    return cfsar_check_launch("cfsar_attnpool_tokens");
}

extern "C" int cfsar_attnpool_attend(const float* q, const float* kv, float* out, int F, int T, int heads, int head_dim,
                                     float scale, cfsar_stream_t stream) {
    CFSAR_REQUIRE(q && kv && out && F > 0 && T > 0 && heads > 0, "cfsar_attnpool_attend: bad arguments");
    CFSAR_REQUIRE(T <= 512 && head_dim > 0 && head_dim <= 128 && head_dim % 4 == 0, "cfsar_attnpool_attend: T=%d (<= 512), head_dim=%d (<= 128, %% 4)", T, head_dim);
    hipLaunchKernelGGL(attnpool_attend_kernel, dim3(F * heads), dim3(64), 0, static_cast<hipStream_t>(stream), q, kv, out, T,
                       heads, head_dim, scale);
    return cfsar_check_launch("cfsar_attnpool_attend");
}

template <typename TO>
static int launch_stem(const float* frames, const float* w, const float* bias, void* out, int F, int H, int W, int Cout, int relu,
                       hipStream_t s) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long npix = (long long)F * Ho * Wo;
    const dim3 grid((unsigned)((npix + 255) / 256));
    TO* o = static_cast<TO*>(out);
    switch (Cout) {
        case 8: hipLaunchKernelGGL((stem_conv1_kernel<TO, 8>), grid, dim3(256), 0, s, frames, w, bias, o, H, W, Ho, Wo, npix, relu); break;
        case 16: hipLaunchKernelGGL((stem_conv1_kernel<TO, 16>), grid, dim3(256), 0, s, frames, w, bias, o, H, W, Ho, Wo, npix, relu); break;
        case 32: hipLaunchKernelGGL((stem_conv1_kernel<TO, 32>), grid, dim3(256), 0, s, frames, w, bias, o, H, W, Ho, Wo, npix, relu); break;
        case 64: hipLaunchKernelGGL((stem_conv1_kernel<TO, 64>), grid, dim3(256), 0, s, frames, w, bias, o, H, W, Ho, Wo, npix, relu); break;
        default: return cfsar_fail("cfsar_stem_conv3x3_s2: Cout=%d not in {8, 16, 32, 64}", Cout);
    }
    return cfsar_check_launch("cfsar_stem_conv3x3_s2");
}

extern "C" int cfsar_stem_conv3x3_s2(const float* frames, const float* w, const float* bias, void* out, int out_dtype, int F,
                                     int H, int W, int Cout, int relu, cfsar_stream_t stream) {
    CFSAR_REQUIRE(frames && w && out && F > 0 && H > 0 && W > 0, "cfsar_stem_conv3x3_s2: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_dtype == CFSAR_BF16) return launch_stem<__bf16>(frames, w, bias, out, F, H, W, Cout, relu, s);
    if (out_dtype == CFSAR_F16) return launch_stem<_Float16>(frames, w, bias, out, F, H, W, Cout, relu, s);
    if (out_dtype == CFSAR_F32) return launch_stem<float>(frames, w, bias, out, F, H, W, Cout, relu, s);
    return cfsar_fail("cfsar_stem_conv3x3_s2: bad out_dtype %d", out_dtype);
}

// ============================================================================================================
// Direct 3x3 / pad 1 / stride 1 convolution for NARROW channel counts (Cin, Cout in {32, 64}: the RN50 stem convs 2 / 3 and the
// conv2 of layer1, few_shot.py:549-554, 196-197), bf16 NHWC in, bf16 NHWC out, bias (+ ReLU) fused.
//
// The implicit-GEMM path (gemm.hip, p3 CONV NARROW) gathers every input pixel NINE times from L2 into the operand tile; with 64-128
// bytes per pixel these convs are bound by that gather (1.7 / 1.7 / 0.6 ms per launch at 16 episodes against 0.2-0.6 ms of HBM
// time).  Here:
//   * the NHWC tensor is ONE linear stream of pixels; a persistent workgroup (4 waves; two workgroups per CU) walks a contiguous
//     range of tiles (128 pixels for Cin = 32, 64 for Cin = 64: 8 KiB of the stream) and keeps a RING of 8 tiles' worth of the
//     stream in LDS (64 KiB), filled by LDS-DMA one chunk per tile, SIX chunks ahead: every pixel is fetched once per workgroup
//     (plus the halo at the two ends of its range);
//   * a tap is an LDS read at (pixel + dy W + dx) mod ring -- the MFMA "B" fragment of 32 consecutive pixels x 8 channels is one
//     ds_read_b128 (128-byte rows, chunk ^ ((row >> 1) & 7); for Cin = 32 two pixels share a row and the mask is 3: both layouts
//     are bank-conflict free for every tap offset, checked by brute force over the ds_read_b128 lane groups), lanes whose tap falls
//     outside the image read a 128-byte zero row instead.  The address of a tap advances by a constant per tile and its swizzle
//     term never changes (rows move by multiples of 16): one add + and + select per tap and tile;
//   * the WEIGHTS live in registers for the whole launch: a wave owns 32 output channels (9 taps x Cin/16 fragments = 72 / 144
//     VGPRs) and 32 or 64 of the tile's pixels, so the inner loop is one ds_read_b128 per MFMA and nothing else; the fragments
//     are read 3-6 steps ahead, pinned by scheduling barriers (left alone the compiler serialises read -> wait -> MFMA through
//     one fragment register);
//   * epilogue: bias is the accumulators' initial value; ReLU, bf16 pack, transpose through a 2 KiB wave-private slab, 16-byte
//     stores (4 lanes per pixel: its 64 bytes of this wave's 32 channels).
// Measured at 16 episodes (1 280 frames; tools/rn_conv_ab.py): stem conv2 1 701 -> 447 us (4.6 TB/s of in + out), stem conv3
// 1 723 -> 774 us (4.0 TB/s), layer1 conv2 597 -> 315 us (940 TFLOP/s).
// ============================================================================================================
namespace {

__device__ __forceinline__ void dc_glds16(const char* src, unsigned lds_addr) {
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

template <typename F, int... Is>
__device__ __forceinline__ void dc_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void dc_static_for(F&& f) {
    dc_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

struct DirectConvArgs {
    const char* in;
    const char* w;
    char* out;
    const float* bias;
    int M, H, W;            // pixels in the stream, image height / width
    unsigned invW, invH;    // floor(2^32 / W) + 1, floor(2^32 / H) + 1: exact quotients for operands < 2^25
    int ldw;                // weight row pitch in elements (round_up(9 Cin, 64))
    int relu;
    int ntiles;
#ifdef CFSAR_DEV
    int dbg;                // ablation bits (tools/rn_conv_ab.py): 1 no stores, 2 tap addresses of the first tile only, 4 no MFMAs, 8 no barrier
#endif
};
#ifdef CFSAR_DEV
#define DCDBG(p) ((p).dbg)
#else
#define DCDBG(p) 0
#endif

// pixels per tile = per DMA chunk: 8 KiB of the stream either way (128 pixels of 64 bytes, 64 pixels of 128 bytes), so that the ring
// is 64 KiB for both channel counts and two workgroups share a CU
template <int CIN> struct DcGeom { static constexpr int TILE = CIN == 64 ? 64 : 128; };
constexpr int DC_NCH = 8;           // chunks in the LDS ring

template <int N> __device__ __forceinline__ void dc_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

template <int CIN, int COUT, typename T = __bf16>
__global__ __launch_bounds__(256, 2) void conv3x3_direct_kernel(DirectConvArgs p) {
    constexpr int DC_TILE = DcGeom<CIN>::TILE, DC_RING = DC_NCH * DC_TILE;
    constexpr int PXB = CIN * 2;                  // bytes per pixel
    constexpr int RPC = DC_TILE * PXB / 128;      // 128-byte LDS rows per chunk (64)
    constexpr int DPW = RPC / 32;                 // LDS-DMA instructions per wave and chunk (1 KiB each)
    constexpr int RING_BYTES = DC_RING * PXB;
    constexpr int KC = CIN / 16;
    constexpr int WCO = COUT / 32;                // waves across the output channels (32 each)
    constexpr int WPX = 4 / WCO;                  // waves across the tile's pixels
    constexpr int MI = DC_TILE / (32 * WPX);      // 32-pixel MFMA tiles per wave
    constexpr int NSTEP = 9 * KC;
    constexpr int SWM = CIN == 64 ? 7 : 3;
    constexpr int ZOFF = RING_BYTES;              // 128 bytes of zeros
    constexpr int SLAB0 = RING_BYTES + 128;
    constexpr int SLAB = 32 * 32 * 2;             // 32 pixels x 32 channels
    constexpr int NS = 2 * MI;                    // store instructions per tile and wave (16 rows x 64 bytes each)
    constexpr int NACC = MI == 1 ? 2 : MI;        // one pixel tile: two accumulators break the MFMA dependence chain
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave % WCO, pg = wave / WCO;
    const int lr = lane & 31, hi = lane >> 5;

    const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t0 = (int)blockIdx.x * per;
    const int t1 = t0 + per < p.ntiles ? t0 + per : p.ntiles;
    if (t0 >= t1) return;

    if (tid < 32) *reinterpret_cast<unsigned*>(smem + ZOFF + tid * 4) = 0u;

    // weights -> registers: fragment (tap, kc) = W[32 cg + lr][tap Cin + 16 kc + 8 hi .. + 8]
    uint4 wr[NSTEP];
    dc_static_for<NSTEP>([&](auto S) {
        constexpr int tap = S.value / KC, kc = S.value % KC;
        wr[S.value] = *reinterpret_cast<const uint4*>(p.w + ((size_t)(32 * cg + lr) * p.ldw + tap * CIN + 16 * kc + 8 * hi) * 2);
    });
    // bias in the accumulator layout: element 4 g + j is output channel 32 cg + 8 g + 4 hi + j
    float br[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) br[e] = p.bias ? p.bias[32 * cg + 8 * (e >> 2) + 4 * hi + (e & 3)] : 0.0f;

    // LDS-DMA: instruction j = wave + 4 i of a chunk fills its rows 8 j .. 8 j + 7; lane (rr, pos) brings the 16-byte piece that
    // belongs at position pos of row 8 j + rr, i.e. chunk pos ^ swizzle(row) of that row of the stream
    const long long nrows = (long long)p.M * PXB / 128;
    int srow[DPW], soff[DPW];
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
        const int rowin = 8 * (wave + 4 * i) + (lane >> 3);
        srow[i] = rowin;
        soff[i] = ((lane & 7) ^ ((rowin >> 1) & SWM)) << 4;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto issue = [&](int c) __attribute__((always_inline)) {
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(c & (DC_NCH - 1)) * (RPC * 128) + (unsigned)wave * 1024u);
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            long long g = (long long)c * RPC + srow[i];
            g = g < 0 ? 0 : (g >= nrows ? nrows - 1 : g);
            dc_glds16(p.in + g * 128 + soff[i], base + i * 4096);
        }
    };
    for (int c = t0 - 1; c <= t0 + DC_NCH - 3; ++c) issue(c);

    const int ck0 = hi << 4;
    char* slab = smem + SLAB0 + wave * SLAB;
    const int rsub = lane >> 2, Q = lane & 3;     // read-back: 4 lanes per 64-byte row, 16 rows per instruction

    // Tap addresses.  A tile later the same lane's pixel sits 128 pixels further in the ring: the row address moves by 128 PXB bytes
    // (mod ring) and the swizzle term (row >> 1) & SWM (and, for Cin = 32, which half of the row) does NOT change -- rows move by
    // multiples of 16.  So per tile and tap: one add + and, and one select against the zero row for taps outside the image (the
    // zero row is 128 bytes: any swizzle term stays inside it).
    int rbase[MI][9], aswz[MI][9];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int pix = t0 * DC_TILE + (pg * MI + mi) * 32 + lr;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int byte = ((pix + dy * p.W + dx) & (DC_RING - 1)) * PXB;      // this pixel's bytes in the ring
            const int lrow = byte >> 7;
            rbase[mi][tap] = lrow << 7;
            aswz[mi][tap] = (byte & 64) ^ (((lrow >> 1) & SWM) << 4);           // Cin = 32: odd pixels are chunks 4-7 of their row
        }
    }

    for (int t = t0; t < t1; ++t) {
        int abase[MI][9];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int pix = t * DC_TILE + (pg * MI + mi) * 32 + lr;
            const int pc = pix < p.M ? pix : p.M - 1;
            const int rowi = (int)__umulhi((unsigned)pc, p.invW);
            const int x = pc - rowi * p.W;
            const int y = rowi - (int)__umulhi((unsigned)rowi, p.invH) * p.H;
            const bool yok[3] = {y >= 1, true, y + 1 < p.H}, xok[3] = {x >= 1, true, x + 1 < p.W};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                abase[mi][tap] = (yok[tap / 3] && xok[tap % 3]) ? rbase[mi][tap] : ZOFF;
                rbase[mi][tap] = (rbase[mi][tap] + DC_TILE * PXB) & (RING_BYTES - 1);
            }
        }
        // chunk t + 1 has landed: behind it in the queue are chunks t + 2 ... t + 5 and the stores of up to five tiles
        if (t - t0 < 5) dc_wait_vm<4 * DPW>();
        else dc_wait_vm<4 * DPW + 5 * NS>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(DCDBG(p) & 8)) __builtin_amdgcn_s_barrier();             // every wave is done with tile t - 1: chunk t - 2 is dead
        issue(t + DC_NCH - 2);

        f32x16 acc[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = (a < MI) ? br[e] : 0.0f;

        // K loop: 9 taps x Cin/16 steps; the fragments of step s + 1 are read while the MFMAs of step s run
        constexpr int PD = MI == 2 ? 3 : (CIN == 64 ? 4 : 6);   // steps of read-ahead (~130-200 cycles of MFMA work; Cin = 64: 144 weight registers leave room for 4)
        uint4 af[PD + 1][MI];
        auto read_step = [&](auto S, uint4 (&dst)[MI]) __attribute__((always_inline)) {
            constexpr int tap = S.value / KC, kc = S.value % KC;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                dst[mi] = *reinterpret_cast<const uint4*>(smem + abase[mi][tap] + (aswz[mi][tap] ^ ((2 * kc) << 4) ^ ck0));
        };
        dc_static_for<PD>([&](auto S) { read_step(S, af[S.value]); });
        if (!(DCDBG(p) & 4))
        dc_static_for<NSTEP>([&](auto S) {
            constexpr int s = S.value;
            if constexpr (s + PD < NSTEP) read_step(std::integral_constant<int, s + PD>{}, af[(s + PD) % (PD + 1)]);
            __builtin_amdgcn_sched_barrier(0);    // keep the reads of step s + PD AHEAD of these MFMAs (the scheduler otherwise serialises
#pragma unroll                                    // read -> wait -> MFMA with one fragment register)
            for (int mi = 0; mi < MI; ++mi) {
                const int a = MI == 1 ? (s & 1) : mi;
                acc[a] = cfsar_mfma_32x32x16<T>(wr[s], af[s % (PD + 1)][mi], acc[a]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (MI == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][e] += acc[1][e];
        }

        // epilogue per 32-pixel tile: [ReLU] -> bf16 -> slab (8-byte slot (2 g + hi) ^ (lr & 7) of row lr) -> 16-byte stores, 4 lanes
        // per pixel (its 64 bytes of this wave's 32 channels)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typename Vec2B<T>::v4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[mi][4 * g + j];
                    if (p.relu) v = fmaxf(v, 0.0f);
                    o[j] = (T)v;
                }
                *reinterpret_cast<typename Vec2B<T>::v4*>(slab + lr * 64 + (((2 * g + hi) ^ (lr & 7)) << 3)) = o;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private slab: in-order DS + this wait
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 16 + rsub;
                const int f = row & 7;
                uint4 d = *reinterpret_cast<const uint4*>(slab + row * 64 + ((Q ^ (f >> 1)) << 4));
                if (f & 1) d = uint4{d.z, d.w, d.x, d.y};
                const int opix = t * DC_TILE + (pg * MI + mi) * 32 + row;
                uint4* dst = reinterpret_cast<uint4*>(p.out + (size_t)opix * (COUT * 2) + cg * 64 + Q * 16);
                if (opix < p.M && !(DCDBG(p) & 1)) {
                    if (DCDBG(p) & 16) {
                        typedef unsigned dc_u32x4 __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(dc_u32x4{d.x, d.y, d.z, d.w}, reinterpret_cast<dc_u32x4*>(dst));
                    }
                    else *dst = d;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no LDS-DMA may outlive the workgroup
}

template <int CIN, int COUT, typename T>
int launch_direct_conv(const DirectConvArgs& a, hipStream_t s) {
    constexpr int LDS = DC_NCH * DcGeom<CIN>::TILE * CIN * 2 + 128 + 4 * 2048;
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&conv3x3_direct_kernel<CIN, COUT, T>), LDS, "cfsar_conv3x3_nhwc(direct)")) return rc;
    const int wgs = cfsar_num_cus() * 2;        // 72 KiB of LDS, <= 256 registers: two workgroups per CU
    const int grid = a.ntiles < wgs ? a.ntiles : wgs;
    hipLaunchKernelGGL((conv3x3_direct_kernel<CIN, COUT, T>), dim3(grid), dim3(256), LDS, s, a);
    return cfsar_check_launch("cfsar_conv3x3_nhwc(direct)");
}

}  // namespace

#ifdef CFSAR_DEV
int g_direct_conv_dbg = 0;
#endif
// Called by cfsar_conv3x3_nhwc (gemm.hip) for the shapes this kernel covers; returns -2 when it does not apply.
int cfsar_conv3x3_direct(const void* in, const void* W, void* out, const float* bias, int F, int H, int Wd, int C, int Cout, int ldw,
                         int ldo, int relu, int f16, hipStream_t s) {
    const long long M = (long long)F * H * Wd;
    if (!((C == 32 || C == 64) && (Cout == 32 || Cout == 64) && !(C == 64 && Cout == 32))) return -2;
    const int tile = C == 64 ? DcGeom<64>::TILE : DcGeom<32>::TILE;
    if (ldo != Cout || Wd + 1 > tile || Wd < 2 || H < 2 || M >= (1ll << 25) || (M * C * 2) % 128 != 0 || M < tile) return -2;
    DirectConvArgs a;
    a.in = static_cast<const char*>(in);
    a.w = static_cast<const char*>(W);
    a.out = static_cast<char*>(out);
    a.bias = bias;
    a.M = (int)M; a.H = H; a.W = Wd;
    a.invW = (unsigned)((1ull << 32) / (unsigned)Wd) + 1u;
    a.invH = (unsigned)((1ull << 32) / (unsigned)H) + 1u;
    a.ldw = ldw;
    a.relu = relu;
    a.ntiles = (int)((M + tile - 1) / tile);
#ifdef CFSAR_DEV
    a.dbg = g_direct_conv_dbg;
#endif
    if (f16) {
        if (C == 32 && Cout == 32) return launch_direct_conv<32, 32, _Float16>(a, s);
        if (C == 32 && Cout == 64) return launch_direct_conv<32, 64, _Float16>(a, s);
        return launch_direct_conv<64, 64, _Float16>(a, s);
    }
    if (C == 32 && Cout == 32) return launch_direct_conv<32, 32, __bf16>(a, s);
    if (C == 32 && Cout == 64) return launch_direct_conv<32, 64, __bf16>(a, s);
    return launch_direct_conv<64, 64, __bf16>(a, s);
}
