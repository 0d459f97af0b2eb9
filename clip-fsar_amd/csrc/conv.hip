// N3: data-movement kernels of the CLIP ModifiedResNet ("RN50") tower (reference few_shot.py:182-227, 542-602).
// Activations are NHWC ([F*H*W, C] row-major) so 1x1 convolutions are plain GEMM rows and 3x3 convolutions become a
// GEMM after a tap-major gather (column = (ky*3+kx)*C + c).  BatchNorm (eval) is folded into the conv weights / bias on
// the host, ReLU and the identity add are GEMM epilogues (cfsar_gemm_ex); what remains here is memory-bound.
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
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)acc[j];
            *reinterpret_cast<bf16x8*>(op + o0) = o;
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
    if (dtype == CFSAR_BF16) {
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
    if (dtype == CFSAR_BF16)
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
    if (out_dtype == CFSAR_F32) return launch_stem<float>(frames, w, bias, out, F, H, W, Cout, relu, s);
    return cfsar_fail("cfsar_stem_conv3x3_s2: bad out_dtype %d", out_dtype);
}
