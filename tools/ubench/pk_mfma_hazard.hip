// Microtest (gfx950): how many wait states does "VALU writes a VGPR -> MFMA reads it as SrcC" need, for a plain v_mul_f32 and for the
// two-pass packed v_pk_mul_f32, and which lanes / which half of the packed pair go stale when there are too few?
// (csrc/gemm_vit.hip initialises accumulators with packed multiplies right before the first MFMA of a tile.)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_mfma_hazard.hip -o pk_mfma_hazard && ./pk_mfma_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define STR2(x) #x
#define STR(x) STR2(x)
// NOPS < 0: no s_nop at all; otherwise "s_nop NOPS" = NOPS + 1 wait states
template <int PACKED, int NOPS>
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters, int n) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    const float a = in[i];
    h4 z = {0, 0, 0, 0};
    for (int t = 0; t < iters; ++t) {
        const float x = a + 0.001f * (float)t;
        f2 xx = {x, x + 1.0f};
        float* q = out + 2 * ((size_t)t * n + i);
#define BODY(NOPSTR)                                                                                                                   \
        if (PACKED)                                                                                                                    \
            asm volatile("v_mov_b32 v44, 0x7149f2ca\n\tv_mov_b32 v45, 0x7149f2ca\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\ts_nop 7\n\ts_nop 7\n\t" \
                         "v_pk_mul_f32 v[44:45], %1, %1\n\t" NOPSTR "v_mfma_f32_4x4x4_16b_f16 v[44:47], %2, %2, v[44:47]\n\ts_nop 7\n\ts_nop 7\n\t"      \
                         "global_store_dwordx2 %0, v[44:45], off" : : "v"(q), "v"(xx), "v"(z) : "memory", "v44", "v45", "v46", "v47");          \
        else                                                                                                                           \
            asm volatile("v_mov_b32 v44, 0x7149f2ca\n\tv_mov_b32 v45, 0x7149f2ca\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\ts_nop 7\n\ts_nop 7\n\t" \
                         "v_mul_f32 v44, %1, %1\n\tv_mul_f32 v45, %3, %3\n\t" NOPSTR "v_mfma_f32_4x4x4_16b_f16 v[44:47], %2, %2, v[44:47]\n\ts_nop 7\n\ts_nop 7\n\t" \
                         "global_store_dwordx2 %0, v[44:45], off" : : "v"(q), "v"(xx[0]), "v"(z), "v"(xx[1]) : "memory", "v44", "v45", "v46", "v47");
        if constexpr (NOPS < 0) { BODY("") }
        else if constexpr (NOPS == 0) { BODY("s_nop 0\n\t") }
        else if constexpr (NOPS == 1) { BODY("s_nop 1\n\t") }
        else if constexpr (NOPS == 2) { BODY("s_nop 2\n\t") }
        else if constexpr (NOPS == 3) { BODY("s_nop 3\n\t") }
        else if constexpr (NOPS == 4) { BODY("s_nop 4\n\t") }
        else { BODY("s_nop 7\n\t") }
    }
}

template <int PACKED, int NOPS>
static void run(const char* name) {
    const int blocks = 1024, iters = 32, n = blocks * 512;
    float *in, *o;
    hipMalloc(&in, n * 4); hipMalloc(&o, (size_t)n * iters * 8);
    float* h = (float*)malloc(n * 4);
    for (int i = 0; i < n; ++i) h[i] = 0.5f + (float)(i % 977) / 977.0f;
    hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    float* ho = (float*)malloc((size_t)n * iters * 8);
    long bad_lo = 0, bad_hi = 0, lanes[4] = {0, 0, 0, 0}, total = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipMemset(o, 0xff, (size_t)n * iters * 8);
        hipLaunchKernelGGL((k<PACKED, NOPS>), dim3(blocks), dim3(512), 0, 0, in, o, iters, n);
        hipDeviceSynchronize();
        hipMemcpy(ho, o, (size_t)n * iters * 8, hipMemcpyDeviceToHost);
        for (int t = 0; t < iters; ++t)
            for (int i = 0; i < n; ++i) {
                const float x = h[i] + 0.001f * (float)t, e0 = x * x, e1 = (x + 1.0f) * (x + 1.0f);
                const float* q = ho + 2 * ((size_t)t * n + i);
                ++total;
                if (q[0] != e0) { ++bad_lo; ++lanes[(i % 64) / 16]; }
                if (q[1] != e1) { ++bad_hi; ++lanes[(i % 64) / 16]; }
            }
    }
    printf("%-34s wrong: low reg %9ld  high reg %9ld of %ld   by lane quarter: %ld %ld %ld %ld\n", name, bad_lo, bad_hi, total, lanes[0], lanes[1],
           lanes[2], lanes[3]);
    hipFree(in); hipFree(o); free(h); free(ho);
}

int main() {
    run<0, -1>("v_mul x2 -> mfma, 0 wait states");
    run<0, 0>("v_mul x2 -> mfma, 1 wait state");
    run<0, 1>("v_mul x2 -> mfma, 2 wait states");
    run<0, 2>("v_mul x2 -> mfma, 3 wait states");
    run<1, -1>("v_pk_mul -> mfma, 0 wait states");
    run<1, 0>("v_pk_mul -> mfma, 1 wait state");
    run<1, 1>("v_pk_mul -> mfma, 2 wait states");
    run<1, 2>("v_pk_mul -> mfma, 3 wait states");
    run<1, 3>("v_pk_mul -> mfma, 4 wait states");
    run<1, 4>("v_pk_mul -> mfma, 5 wait states");
    run<1, 7>("v_pk_mul -> mfma, 8 wait states");
    return 0;
}
