// Shared device/host helpers for libclipfsar_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/clipfsar_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
// 2-byte element type -> its 4- and 8-wide vectors (epilogues are generic over bf16 / fp16 outputs)
template <typename T> struct Vec2B;
template <> struct Vec2B<__bf16> { typedef bf16x4 v4; typedef bf16x8 v8; };
template <> struct Vec2B<_Float16> { typedef f16x4 v4; typedef f16x8 v8; };
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <> struct Vec2B<float> { typedef f32x4 v4; typedef f32x4 v8; };   // placeholder: float paths never use it

#define CFSAR_WAVE 64

// v_mfma_f32_32x32x16_{bf16,f16} on 16-byte operand fragments, selected by the 2-byte element type
#include <type_traits>
template <typename T2>
__device__ __forceinline__ f32x16 cfsar_mfma_32x32x16(uint4 a, uint4 b, f32x16 c) {
    if constexpr (std::is_same<T2, _Float16>::value)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

extern thread_local char cfsar_err_buf[512];
int cfsar_fail(const char* fmt, ...);
int cfsar_check_launch(const char* what);
int cfsar_ensure_lds(const void* fn, int bytes, const char* what);   // per-(device, kernel) dynamic-LDS limit, cached
int cfsar_num_cus();                                                 // compute units of the current device, cached

// conv.hip: direct 3x3 convolution for Cin, Cout in {32, 64} (bf16 or, f16 != 0, fp16 NHWC, no residual); -2 = shape not covered, use the implicit GEMM
int cfsar_conv3x3_direct(const void* in, const void* W, void* out, const float* bias, int F, int H, int Wd, int C, int Cout, int ldw,
                         int ldo, int relu, int f16, hipStream_t s);

#define CFSAR_REQUIRE(cond, ...)                    \
    do {                                            \
        if (!(cond)) return cfsar_fail(__VA_ARGS__); \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// sum over each aligned group of 8 consecutive lanes, result in all 8: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror as
// DPP-modified VALU adds (a __shfl_xor is a ds_bpermute round trip through the LDS pipe)
template <int CTRL>
__device__ __forceinline__ float cfsar_dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float cfsar_dpp_sum8(float v) {
    v += cfsar_dpp_move<0xB1>(v);
    v += cfsar_dpp_move<0x4E>(v);
    v += cfsar_dpp_move<0x141>(v);
    return v;
}
