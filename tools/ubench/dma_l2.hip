// Microbenchmark: LDS-DMA streaming from an L2-resident (per-XCD 2 MiB) region with NO L1 reuse, GEMM-like row patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ void glds16_asm(const char* src, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
}
// MODE 0: 8 rows x 128 B per instruction (full lines), MODE 1: 16 rows x 64 B (half lines); row stride 1536 B (K = 768 bf16)
template <int MODE, int NPER>
__global__ __launch_bounds__(512) void k(const char* buf, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const size_t region = 2u << 20;                                   // per-XCD region (L2-resident)
    const char* rbase = buf + (size_t)(blockIdx.x & 7) * region;
    const int rows_per_instr = MODE == 0 ? 8 : 16;
    const size_t lane_off = MODE == 0 ? (size_t)(lane >> 3) * 1536 + (lane & 7) * 16 : (size_t)(lane >> 2) * 1536 + (lane & 3) * 16;
    size_t pos = ((size_t)(blockIdx.x >> 3) * 61 + wave * 7) * 1536 * 16;   // spread WGs/waves over the region
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NPER; ++j) {
            const size_t off = (pos + (size_t)j * rows_per_instr * 1536) % (region - 65536);
            glds16_asm(rbase + off + lane_off, __builtin_amdgcn_readfirstlane(lds0 + wave * 8192 + (j & 7) * 1024));
        }
        pos += (size_t)NPER * rows_per_instr * 1536 + 128;              // next rows, next k columns
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (((float*)smem)[threadIdx.x] == 123.f) sink[0] = 1.f;
}
int main() {
    char* buf; float* sink;
    hipMalloc(&buf, 17u << 20); hipMemset(buf, 1, 17u << 20); hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2048;
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL((k<0, 4>), dim3(256), dim3(512), 65536, 0, buf, iters, sink);
            else hipLaunchKernelGGL((k<1, 4>), dim3(256), dim3(512), 65536, 0, buf, iters, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double bytes = 256.0 * 8 * iters * 4 * 1024;
            printf("L2-stream %s: %.3f ms  %.2f TB/s total  %.1f GB/s per CU\n", mode ? "16 rows x 64 B" : "8 rows x 128 B", ms,
                   bytes / ms / 1e9, bytes / ms / 1e6 / 256);
        }
    return 0;
}
