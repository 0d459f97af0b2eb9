#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(int* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 64];
    const int l = threadIdx.x;
    for (int i = l; i < 64 * 64; i += 64) lds[i] = (short)i;     // element value = its index: row = i / 64, col = i % 64 (128-B rows)
    __syncthreads();
    // lane i of each 16-lane group g: row = 4 g + (i >> 2), cols 4 (i & 3) .. +3  (a [4][16] block at rows 4g.., cols 0..15)
    const int i = l & 15, g = l >> 4;
    const short* p = &lds[(4 * g + (i >> 2)) * 64 + 4 * (i & 3)];
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
    int* d; hipMalloc(&d, 256 * 4); int h[256];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l*4+j] / 64, h[l*4+j] % 64); printf("\n"); }
    return 0;
}
