// Microtest (gfx950): packed-fp32 VALU results consumed by MFMAs while the SIMD's other wave keeps the matrix pipe busy.
// Mirrors the accumulator initialisation of csrc/gemm_vit.hip (LN-folded instance): 128 accumulator registers are written by
// v_pk_mul_f32 (bias x per-row scale), then 8 MFMAs (32x32x16, A = B = 0) read them as SrcC and write them back; the result must be
// bias x scale in every lane.  PACKED = 0 does the same with v_mul_f32.  512 threads per workgroup, 256 VGPRs -> two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_under_mfma.hip -o pk_under_mfma && ./pk_under_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PACKED>
__global__ __launch_bounds__(512, 2) void k(const float* bias, const float* scale, unsigned* bad, unsigned* badlane, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[8];
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    float b[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) b[e] = bias[(threadIdx.x * 16 + e) & 4095];
    unsigned nbad = 0, lanes = 0;
    for (int t = 0; t < iters; ++t) {
        float sc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sc[j] = scale[(blockIdx.x * 8 + j + t) & 1023];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                if (PACKED) {
                    f2 r, x = {b[e], b[e + 1]}, s = {sc[j], sc[j]};
                    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(s));
                    acc[j][e] = r[0];
                    acc[j][e + 1] = r[1];
                } else {
                    float r0, r1;
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(b[e]), "v"(sc[j]));
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r1) : "v"(b[e + 1]), "v"(sc[j]));
                    acc[j][e] = r0;
                    acc[j][e + 1] = r1;
                }
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z, z, acc[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r2 = 0; r2 < 4; ++r2) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(z, z, acc[j], 0, 0, 0);   // keep the pipe busy
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (acc[j][e] != b[e] * sc[j]) { ++nbad; lanes |= 1u << (lane >> 4); }
    }
    if (nbad) { atomicAdd(bad, nbad); atomicOr(badlane, lanes); }
}

int main() {
    float *bias, *scale; unsigned *bad, *badlane;
    hipMalloc(&bias, 4096 * 4); hipMalloc(&scale, 1024 * 4); hipMalloc(&bad, 4); hipMalloc(&badlane, 4);
    static float hb[4096], hs[1024];
    for (int i = 0; i < 4096; ++i) hb[i] = 0.25f + (float)((i * 37) % 1013) / 511.0f;
    for (int i = 0; i < 1024; ++i) hs[i] = 0.5f + (float)((i * 11) % 251) / 97.0f;
    hipMemcpy(bias, hb, sizeof(hb), hipMemcpyHostToDevice); hipMemcpy(scale, hs, sizeof(hs), hipMemcpyHostToDevice);
    for (int packed = 1; packed >= 0; --packed) {
        unsigned tot = 0, lanes = 0;
        for (int rep = 0; rep < 20; ++rep) {
            hipMemset(bad, 0, 4); hipMemset(badlane, 0, 4);
            if (packed) hipLaunchKernelGGL(k<1>, dim3(512), dim3(512), 0, 0, bias, scale, bad, badlane, 400);
            else hipLaunchKernelGGL(k<0>, dim3(512), dim3(512), 0, 0, bias, scale, bad, badlane, 400);
            hipDeviceSynchronize();
            unsigned h1, h2; hipMemcpy(&h1, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&h2, badlane, 4, hipMemcpyDeviceToHost);
            tot += h1; lanes |= h2;
        }
        printf("%s -> mfma SrcC: %u wrong accumulator elements of %.3g (lane-quarter mask 0x%x)\n", packed ? "v_pk_mul_f32" : "v_mul_f32   ", tot,
               20.0 * 512 * 512 * 400 * 128, lanes);
    }
    return 0;
}
