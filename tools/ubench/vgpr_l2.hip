// Microbenchmark: plain global_load_dwordx4 -> VGPR (optionally -> ds_write_b128) streaming from an L2-resident region,
// same access pattern as dma_l2.hip, to compare the register path with the LDS-DMA path.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int TOLDS, int NPER, int HALF = 0>
__global__ __launch_bounds__(512) void k(const char* buf, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const size_t region = 2u << 20;
    const char* rbase = buf + (size_t)(blockIdx.x & 7) * region;
    // HALF = 0: 8 rows x 128 B per instruction (whole cache lines); HALF = 1: 16 rows x 64 B (half lines, BK = 32 slices)
    const size_t lane_off = HALF ? (size_t)(lane >> 2) * 1536 + (lane & 3) * 16 : (size_t)(lane >> 3) * 1536 + (lane & 7) * 16;
    size_t pos = ((size_t)(blockIdx.x >> 3) * 61 + wave * 7) * 1536 * 16;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    f4* lds = (f4*)(smem + wave * 8192) + lane;
    for (int it = 0; it < iters; ++it) {
        f4 v[NPER];
#pragma unroll
        for (int j = 0; j < NPER; ++j) {
            const size_t off = (pos + (size_t)j * (HALF ? 16 : 8) * 1536) % (region - 65536);
            v[j] = __builtin_nontemporal_load((const f4*)(rbase + off + lane_off));
        }
        pos += (size_t)NPER * (HALF ? 16 : 8) * 1536 + (HALF ? 64 : 128);
#pragma unroll
        for (int j = 0; j < NPER; ++j) {
            if (TOLDS) lds[(j & 7) * 64] = v[j];
            else acc += v[j];
        }
    }
    __syncthreads();
    if (TOLDS) acc = lds[0];
    if (acc.x + acc.y + acc.z + acc.w == 123.f) sink[0] = 1.f;
}
int main() {
    char* buf; float* sink;
    hipMalloc(&buf, 17u << 20); hipMemset(buf, 1, 17u << 20); hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2048;
    for (int mode = 0; mode < 5; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL((k<0, 4>), dim3(256), dim3(512), 65536, 0, buf, iters, sink);
            else if (mode == 1) hipLaunchKernelGGL((k<1, 4>), dim3(256), dim3(512), 65536, 0, buf, iters, sink);
            else if (mode == 4) hipLaunchKernelGGL((k<0, 8, 1>), dim3(256), dim3(512), 65536, 0, buf, iters / 2, sink);
            else if (mode == 2) hipLaunchKernelGGL((k<0, 8>), dim3(256), dim3(512), 65536, 0, buf, iters / 2, sink);
            else hipLaunchKernelGGL((k<1, 8>), dim3(256), dim3(512), 65536, 0, buf, iters / 2, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double bytes = 256.0 * 8 * iters * 4 * 1024;
            printf("L2-stream vgpr mode %d (%s, %d loads in flight): %.3f ms  %.2f TB/s total  %.1f GB/s per CU\n", mode,
                   mode == 4 ? "-> regs, 16 rows x 64 B" : (mode & 1) ? "-> ds_write_b128" : "-> regs", mode < 2 ? 4 : 8, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
        }
    return 0;
}
