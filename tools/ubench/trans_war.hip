// Microtest (gfx950): does a VALU write to the SOURCE register of the immediately preceding transcendental instruction race
// with the transcendental's operand read?  v_rcp_f32 is quarter rate (16 lanes per cycle); the next VALU instruction may
// overwrite the source before the last lane group has been read.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/trans_war.hip -o trans_war && ./trans_war
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
template <int NOPS>
__global__ void k(const float* in, float* out, int iters) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    float a = in[i], acc = 0.f;
    for (int t = 0; t < iters; ++t) {
        float src = a + (float)t, r, junk = 12345.0f;
        if constexpr (NOPS == 0)
            asm volatile("v_rcp_f32 %0, %1\n\tv_mov_b32 %1, %2" : "=&v"(r), "+v"(src) : "v"(junk));
        else if constexpr (NOPS == 1)
            asm volatile("v_rcp_f32 %0, %1\n\ts_nop 0\n\tv_mov_b32 %1, %2" : "=&v"(r), "+v"(src) : "v"(junk));
        else
            asm volatile("v_rcp_f32 %0, %1\n\tv_mov_b32 %3, %2\n\tv_mov_b32 %1, %2" : "=&v"(r), "+v"(src), "+v"(junk) : "v"(junk));
        acc += r + src * 0.0f;
    }
    out[i] = acc;
}
int main() {
    const int n = 256 * 1024, iters = 64;
    float *in, *out, *h = (float*)malloc(n * 4), *ho = (float*)malloc(n * 4);
    for (int i = 0; i < n; ++i) h[i] = 1.0f + (i % 977) * 0.01f;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4);
    hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    for (int v = 0; v < 3; ++v) {
        if (v == 0) hipLaunchKernelGGL(k<0>, dim3(n / 256), dim3(256), 0, 0, in, out, iters);
        if (v == 1) hipLaunchKernelGGL(k<1>, dim3(n / 256), dim3(256), 0, 0, in, out, iters);
        if (v == 2) hipLaunchKernelGGL(k<2>, dim3(n / 256), dim3(256), 0, 0, in, out, iters);
        hipMemcpy(ho, out, n * 4, hipMemcpyDeviceToHost);
        int bad = 0, badlane[64] = {0};
        for (int i = 0; i < n; ++i) {
            double ref = 0; for (int t = 0; t < iters; ++t) ref += 1.0 / (double)(h[i] + (float)t);
            if (fabs(ho[i] - ref) > 1e-3 * fabs(ref)) { ++bad; ++badlane[i & 63]; }
        }
        printf("variant %d (0 = overwrite right after v_rcp, 1 = s_nop 0 between, 2 = one unrelated VALU between): %d wrong of %d; by lane group:", v, bad, n);
        for (int g = 0; g < 4; ++g) { int s = 0; for (int l = 0; l < 16; ++l) s += badlane[g * 16 + l]; printf(" %d", s); }
        printf("\n");
    }
    return 0;
}
