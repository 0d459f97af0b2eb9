// Microtest (gfx950): transcendental-source write-after-read.
//
// v_exp_f32 / v_rcp_f32 run in the quarter-rate transcendental pipe.  Question: if the NEXT instruction is an independent
// full-rate VALU op that overwrites the transcendental's SOURCE register, can the transcendental still read the old value in
// every lane?  Three instruction shapes, each as ONE asm block so the order is exactly as written:
//   S  single:      v_rcp r, s ; v_mov s, junk                                  (what hipcc's integer division does)
//   G  GELU-like:   v_exp ; v_exp ; v_rcp r0, s ; v_add s, 1, e1 ; v_rcp r1, s ; v_add s, 1, junk     (csrc/gemm_vit.hip before the fix)
//   A  softmax-like: v_fma z ; v_exp e0, z ; v_fma z ; v_exp e1, z ; ... x 8     (csrc/attention.hip)
// Each is checked against the same math done with separate registers.  512-thread blocks, 2 blocks per CU -> 4 waves per SIMD
// compete for the transcendental pipe.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/trans_war.hip -o trans_war && ./trans_war
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    float a = in[i], acc = 0.f;
    for (int t = 0; t < iters; ++t) {
        const float x0 = a + 0.01f * (float)t, x1 = a * 0.5f + 0.02f * (float)t;
        if constexpr (SHAPE == 0) {                // S
            float s = x0, r, junk = 12345.0f;
            asm volatile("v_rcp_f32 %0, %1\n\tv_mov_b32 %1, %2" : "=&v"(r), "+v"(s) : "v"(junk));
            acc += r;
        } else if constexpr (SHAPE == 1) {         // G: r0 = 1/(1+2^x0), r1 = 1/(1+2^x1)
            float e0 = x0, e1 = x1, s, r0, r1, junk = 777.0f;
            asm volatile(
                "v_exp_f32 %0, %0\n\t"
                "v_exp_f32 %1, %1\n\t"
                "v_add_f32 %2, 1.0, %0\n\t"
                "v_rcp_f32 %3, %2\n\t"
                "v_add_f32 %2, 1.0, %1\n\t"        // overwrites the source of the v_rcp right before it
                "v_rcp_f32 %4, %2\n\t"
                "v_add_f32 %2, 1.0, %5"            // ... and again
                : "+v"(e0), "+v"(e1), "=&v"(s), "=&v"(r0), "=&v"(r1) : "v"(junk));
            acc += r0 + 2.0f * r1 + 0.0f * s;
        } else {                                   // A: sum_j 2^(c * y_j - m), z register recycled after every v_exp
            float z, e[8], y[8];
            for (int j = 0; j < 8; ++j) y[j] = x0 * (0.1f * j) - x1;
            const float c = 0.7f, m = 1.5f;
            asm volatile(
                "v_fma_f32 %8, %9, %16, %17\n\tv_exp_f32 %0, %8\n\t"
                "v_fma_f32 %8, %10, %16, %17\n\tv_exp_f32 %1, %8\n\t"
                "v_fma_f32 %8, %11, %16, %17\n\tv_exp_f32 %2, %8\n\t"
                "v_fma_f32 %8, %12, %16, %17\n\tv_exp_f32 %3, %8\n\t"
                "v_fma_f32 %8, %13, %16, %17\n\tv_exp_f32 %4, %8\n\t"
                "v_fma_f32 %8, %14, %16, %17\n\tv_exp_f32 %5, %8\n\t"
                "v_fma_f32 %8, %15, %16, %17\n\tv_exp_f32 %6, %8\n\t"
                "v_fma_f32 %8, %18, %16, %17\n\tv_exp_f32 %7, %8\n\t"
                "v_mov_b32 %8, %17"
                : "=&v"(e[0]), "=&v"(e[1]), "=&v"(e[2]), "=&v"(e[3]), "=&v"(e[4]), "=&v"(e[5]), "=&v"(e[6]), "=&v"(e[7]), "=&v"(z)
                : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(c), "v"(-m), "v"(y[7]));
            for (int j = 0; j < 8; ++j) acc += e[j] * (1.0f + 0.125f * j);
            acc += 0.0f * z;
        }
    }
    out[i] = acc;
}

static double ref(int shape, float a, int iters) {
    double acc = 0;
    for (int t = 0; t < iters; ++t) {
        const float x0 = a + 0.01f * (float)t, x1 = a * 0.5f + 0.02f * (float)t;
        if (shape == 0) acc += 1.0 / x0;
        else if (shape == 1) acc += 1.0 / (1.0 + exp2((double)x0)) + 2.0 / (1.0 + exp2((double)x1));
        else for (int j = 0; j < 8; ++j) acc += exp2(0.7 * (double)(x0 * (0.1f * j) - x1) - 1.5) * (1.0 + 0.125 * j);
    }
    return acc;
}

int main() {
    const int n = 512 * 2048, iters = 256;
    float *in, *out, *h = (float*)malloc(n * 4), *ho = (float*)malloc(n * 4);
    for (int i = 0; i < n; ++i) h[i] = 0.25f + (float)((i * 2654435761u) % 4099u) * 0.001f;
    (void)hipMalloc(&in, n * 4); (void)hipMalloc(&out, n * 4);
    (void)hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    const char* names[3] = {"S single v_rcp + overwrite", "G exp,exp,rcp,add(overwrite),rcp,add(overwrite)", "A fma/exp alternating, z recycled"};
    for (int v = 0; v < 3; ++v) {
        for (int rep = 0; rep < 3; ++rep) {
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(n / 512), dim3(512), 0, 0, in, out, iters);
            if (v == 1) hipLaunchKernelGGL(k<1>, dim3(n / 512), dim3(512), 0, 0, in, out, iters);
            if (v == 2) hipLaunchKernelGGL(k<2>, dim3(n / 512), dim3(512), 0, 0, in, out, iters);
        }
        (void)hipMemcpy(ho, out, n * 4, hipMemcpyDeviceToHost);
        long bad = 0, grp[4] = {0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            const double r = ref(v, h[i], iters);
            if (fabs(ho[i] - r) > 2e-4 * fabs(r) + 1e-6) { ++bad; ++grp[(i & 63) >> 4]; }
        }
        printf("%-52s: %ld wrong of %d threads; by 16-lane group: %ld %ld %ld %ld\n", names[v], bad, n, grp[0], grp[1], grp[2], grp[3]);
    }
    return 0;
}
