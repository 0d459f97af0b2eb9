// Microbenchmark (gfx950): the output burst of a 256 x 256 GEMM tile -- 8 waves per CU, 16 x 16-byte-per-lane stores per wave
// (8 rows x 128 B per wave-instruction at the output's row stride), i.e. 128 KiB per CU per burst -- issued the way the
// epilogue of csrc/gemm_vit.hip issues it.  Questions: (1) how long does ONE CU need for its burst when it is alone, when one CU
// per XCD bursts, and when all 256 CUs burst together (per-CU limit vs per-XCD / chip limit)?  (2) is the time spent ISSUING
// (t_issue: first store -> last store accepted) or DRAINING (t_drain: last store accepted -> vmcnt(0))?  (3) what do the store
// shape (8 x 128 B, 4 x 256 B, 1 x 1 KiB per instruction) and the cache policy (plain / nt / sc1) change?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_burst.hip -o tools/ubench/bin/store_burst && tools/ubench/bin/store_burst
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ void st16(char* dst, u32x4 v) {
    if constexpr (POLICY == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
    else if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
    else *reinterpret_cast<u32x4*>(dst) = v;
}

// SHAPE 0: 8 rows x 128 B per instruction (wave tile 128 rows x 64 two-byte columns; the product epilogue)
// SHAPE 1: 4 rows x 256 B (two waves' columns combined), SHAPE 2: 1 row x 1 KiB (a whole 512-column row band per instruction... x2)
template <int POLICY, int SHAPE>
__global__ __launch_bounds__(512) void burst(char* out, int ld_bytes, int tiles, int active_mod, int gap, long long* times) {
    if ((int)(blockIdx.x % active_mod) != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    u32x4 v = {(unsigned)threadIdx.x, (unsigned)blockIdx.x, 3u, 4u};
    long long t_issue = 0, t_drain = 0;
    for (int t = 0; t < tiles; ++t) {
        // tile origin: row band = block, consecutive tiles move along the columns (512 B per tile) like a column walk
        char* base = out + ((size_t)blockIdx.x * 256) * (size_t)ld_bytes + (size_t)(t % 8) * 512;
        __syncthreads();
        const long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            char* p;
            if constexpr (SHAPE == 0) p = base + (size_t)(wm * 128 + s * 8 + (lane >> 3)) * ld_bytes + wn * 128 + (lane & 7) * 16;
            else if constexpr (SHAPE == 1) p = base + (size_t)(wm * 128 + (wn >> 1) * 64 + s * 4 + (lane >> 4)) * ld_bytes + (wn & 1) * 256 + (lane & 15) * 16;
            else p = base + (size_t)(wave * 32 + s * 2 + (lane >> 5)) * ld_bytes + (lane & 31) * 16;
            v[0] += s;
            st16<POLICY>(p, v);
        }
        const long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t2 = __builtin_readcyclecounter();
        t_issue += t1 - t0;
        t_drain += t2 - t1;
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(127);      // spacing between bursts (~8 K cycles per unit)
    }
    if (lane == 0) {
        times[((size_t)blockIdx.x * 8 + wave) * 2] = t_issue / tiles;
        times[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = t_drain / tiles;
    }
}

template <int POLICY, int SHAPE>
static void run(const char* name, char* out, long long* dt, int ld_bytes, int active_mod, int gap) {
    const int tiles = 64, nb = 256;
    std::vector<long long> h(nb * 16);
    (void)hipMemset(dt, 0, nb * 16 * sizeof(long long));
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((burst<POLICY, SHAPE>), dim3(nb), dim3(512), 0, 0, out, ld_bytes, tiles, active_mod, gap, dt);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((burst<POLICY, SHAPE>), dim3(nb), dim3(512), 0, 0, out, ld_bytes, tiles, active_mod, gap, dt);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipMemcpy(h.data(), dt, nb * 16 * sizeof(long long), hipMemcpyDeviceToHost);
    // per CU: the burst ends when its slowest wave has drained
    double sum_i = 0, sum_t = 0; int n = 0; long long worst = 0;
    for (int blk = 0; blk < nb; blk += active_mod) {
        long long mi = 0, mt = 0;
        for (int w = 0; w < 8; ++w) {
            mi = std::max(mi, h[(blk * 8 + w) * 2]);
            mt = std::max(mt, h[(blk * 8 + w) * 2] + h[(blk * 8 + w) * 2 + 1]);
        }
        sum_i += mi; sum_t += mt; ++n; worst = std::max(worst, mt);
    }
    // s_memtime / readcyclecounter ticks at 100 MHz on gfx950?  report raw ticks and the event-based rate as well
    const double bytes = (double)n * tiles * 131072.0;
    printf("%-34s active CUs %3d gap %d: issue %7.0f ticks, issue+drain %7.0f ticks (worst CU %lld) | kernel %.3f ms -> %.2f TB/s, %.1f GB/s per active CU\n",
           name, n, gap, sum_i / n, sum_t / n, worst, ms, bytes / ms / 1e9, bytes / ms / 1e6 / n);
}

int main() {
    const int ld_bytes = 2304 * 2;                         // QKV output row
    const size_t bytes = (size_t)256 * 256 * ld_bytes;     // 256 row bands of 256 rows
    char* out; long long* dt;
    (void)hipMalloc(&out, bytes);
    (void)hipMalloc(&dt, 256 * 16 * sizeof(long long));
    for (int gap = 0; gap < 2; ++gap)
        for (int am : {1, 8, 256}) {                       // all CUs | 32 blocks (block b -> XCD b % 8: 4 CUs per XCD)... | one CU
            run<0, 0>("plain 8x128B", out, dt, ld_bytes, am, gap);
            run<2, 0>("sc1   8x128B", out, dt, ld_bytes, am, gap);
            run<1, 0>("nt    8x128B", out, dt, ld_bytes, am, gap);
            run<2, 1>("sc1   4x256B", out, dt, ld_bytes, am, gap);
            run<2, 2>("sc1   2x512B", out, dt, ld_bytes, am, gap);
            run<0, 2>("plain 2x512B", out, dt, ld_bytes, am, gap);
        }
    return 0;
}
