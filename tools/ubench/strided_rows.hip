// Microbenchmark: HBM read rate of the attention staging pattern -- each workgroup (frame f, head h) reads 197 rows x 128 B for
// K and for V out of a [F*197, 2304] bf16 matrix (row stride 4608 B) -- against the same bytes laid out contiguously per
// (frame, head) (25 KB blocks).  256-thread workgroups, 16-byte loads, data summed into a sink.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(const uint4* buf, int mode, int ntok, float* sink) {
    const int h = blockIdx.x, f = blockIdx.y;
    unsigned acc = 0;
    const size_t row_chunks = 3 * 768 * 2 / 16;      // 288 16-byte chunks per token row
    for (int idx = threadIdx.x; idx < ntok * 8 * 3; idx += 256) {      // q, k, v: 8 chunks per row each
        const int which = idx / (ntok * 8), rem = idx % (ntok * 8);
        const int r = rem >> 3, c = rem & 7;
        size_t off;
        if (mode == 0) off = ((size_t)f * ntok + r) * row_chunks + which * 96 + h * 8 + c;        // strided rows
        else off = (((size_t)f * 36 + which * 12 + h) * ntok + r) * 8 + c;                        // head-major blocks
        const uint4 v = buf[off];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = 1.f;
}
int main() {
    const int F = 640, ntok = 197;
    const size_t bytes = (size_t)F * ntok * 2304 * 2;
    uint4* buf; float* sink;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(12, F), dim3(256), 0, 0, buf, mode, ntok, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%s: %.3f ms  %.2f TB/s\n", mode ? "head-major 25 KB blocks" : "rows of 128 B at stride 4608 B", ms, bytes / ms / 1e9);
        }
    return 0;
}
