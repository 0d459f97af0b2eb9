// Microtest (gfx950): does a VMEM store issued right behind a VALU / transcendental / packed-fp32 / MFMA write of its DATA register
// see the new value in every lane?  (Hunting the "lanes 48-63 of one register are stale" fault seen in csrc/gemm_vit.hip in kernels
// that spill: the 4th quarter of a wave64 VALU op is written last.)  Every shape is ONE asm block: producer, then the store with no
// instruction in between; the register holds 0 before.  A second, safe copy (s_nop 7 twice before the store) goes to another buffer.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_vmem_hazard.hip -o valu_vmem_hazard && ./valu_vmem_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const float* in, float* outA, float* outB, int iters, int n) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    const float a = in[i];
    for (int t = 0; t < iters; ++t) {
        const float x = a + 0.001f * (float)t;
        float* pa = outA + (size_t)t * n + i;
        float* pb = outB + (size_t)t * n + i;
        float r = 0.f, r2 = 0.f;
        if constexpr (SHAPE == 0) {            // transcendental -> store
            asm volatile("v_mov_b32 %0, 0\n\tv_exp_f32 %0, %2\n\tglobal_store_dword %1, %0, off" : "=&v"(r) : "v"(pa), "v"(x) : "memory");
            asm volatile("v_mov_b32 %0, 0\n\tv_exp_f32 %0, %2\n\ts_nop 7\n\ts_nop 7\n\tglobal_store_dword %1, %0, off" : "=&v"(r2) : "v"(pb), "v"(x) : "memory");
        } else if constexpr (SHAPE == 1) {     // rcp -> store
            asm volatile("v_mov_b32 %0, 0\n\tv_rcp_f32 %0, %2\n\tglobal_store_dword %1, %0, off" : "=&v"(r) : "v"(pa), "v"(x) : "memory");
            asm volatile("v_mov_b32 %0, 0\n\tv_rcp_f32 %0, %2\n\ts_nop 7\n\ts_nop 7\n\tglobal_store_dword %1, %0, off" : "=&v"(r2) : "v"(pb), "v"(x) : "memory");
        } else if constexpr (SHAPE == 2) {     // plain VALU -> store
            asm volatile("v_mov_b32 %0, 0\n\tv_mul_f32 %0, %2, %2\n\tglobal_store_dword %1, %0, off" : "=&v"(r) : "v"(pa), "v"(x) : "memory");
            asm volatile("v_mov_b32 %0, 0\n\tv_mul_f32 %0, %2, %2\n\ts_nop 7\n\ts_nop 7\n\tglobal_store_dword %1, %0, off" : "=&v"(r2) : "v"(pb), "v"(x) : "memory");
        } else if constexpr (SHAPE == 3) {     // packed fp32 -> store of the HIGH register
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 xx = {x, x + 1.0f}, rr = {0.f, 0.f}, rr2 = {0.f, 0.f};
            float* qa = outA + 2 * ((size_t)t * n + i);
            float* qb = outB + 2 * ((size_t)t * n + i);
            asm volatile("v_pk_mul_f32 %0, %2, %2\n\tglobal_store_dwordx2 %1, %0, off" : "+v"(rr) : "v"(qa), "v"(xx) : "memory");
            asm volatile("v_pk_mul_f32 %0, %2, %2\n\ts_nop 7\n\ts_nop 7\n\tglobal_store_dwordx2 %1, %0, off" : "+v"(rr2) : "v"(qb), "v"(xx) : "memory");
            r = rr[1]; r2 = rr2[1];
        } else if constexpr (SHAPE == 4) {     // exp -> mul (trans result forwarded) -> store
            float e = 0.f;
            asm volatile("v_mov_b32 %0, 0\n\tv_exp_f32 %1, %3\n\tv_mul_f32 %0, %1, %3\n\tglobal_store_dword %2, %0, off" : "=&v"(r), "=&v"(e) : "v"(pa), "v"(x) : "memory");
            asm volatile("v_mov_b32 %0, 0\n\tv_exp_f32 %1, %3\n\ts_nop 7\n\tv_mul_f32 %0, %1, %3\n\ts_nop 7\n\ts_nop 7\n\tglobal_store_dword %2, %0, off" : "=&v"(r2), "=&v"(e) : "v"(pb), "v"(x) : "memory");
        } else if constexpr (SHAPE == 6) {     // packed fp32 add -> transcendental reading the HIGH half (QuickGELU: 1 + e, then rcp)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 xx = {x, x + 1.0f}, one = {1.0f, 1.0f};
            // fixed registers v[40:41] so that the HIGH half can be named; it holds 1e30 (0x7149f2ca) before: a stale read gives rcp = 1e-30
            asm volatile("v_mov_b32 v41, 0x7149f2ca\n\tv_mov_b32 v40, 0x7149f2ca\n\ts_nop 7\n\tv_pk_add_f32 v[40:41], %2, %3\n\tv_rcp_f32 %0, v41\n\ts_nop 7\n\tglobal_store_dword %1, %0, off"
                         : "=&v"(r) : "v"(pa), "v"(xx), "v"(one) : "memory", "v40", "v41");
            asm volatile("v_mov_b32 v41, 0x7149f2ca\n\tv_mov_b32 v40, 0x7149f2ca\n\ts_nop 7\n\tv_pk_add_f32 v[40:41], %2, %3\n\ts_nop 7\n\ts_nop 7\n\tv_rcp_f32 %0, v41\n\ts_nop 7\n\tglobal_store_dword %1, %0, off"
                         : "=&v"(r2) : "v"(pb), "v"(xx), "v"(one) : "memory", "v40", "v41");
        } else if constexpr (SHAPE == 7) {     // packed fp32 mul -> MFMA reading the pair as SrcC (accumulator initialisation, then the first MFMA)
            typedef float f2 __attribute__((ext_vector_type(2)));
            typedef float f4 __attribute__((ext_vector_type(4)));
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            f2 xx = {x, x + 1.0f};
            h4 z = {0, 0, 0, 0};
            (void)sizeof(f4);
            // v[44:47] = accumulator (1e30 before); A = B = 0 so the MFMA returns SrcC: a stale read shows 1e30 in the stored pair
            asm volatile("v_mov_b32 v44, 0x7149f2ca\n\tv_mov_b32 v45, 0x7149f2ca\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\ts_nop 7\n\t"
                         "v_pk_mul_f32 v[44:45], %1, %1\n\tv_mfma_f32_4x4x4_16b_f16 v[44:47], %2, %2, v[44:47]\n\ts_nop 7\n\ts_nop 7\n\tglobal_store_dwordx2 %0, v[44:45], off"
                         : : "v"(outA + 2 * ((size_t)t * n + i)), "v"(xx), "v"(z) : "memory", "v44", "v45", "v46", "v47");
            asm volatile("v_mov_b32 v44, 0x7149f2ca\n\tv_mov_b32 v45, 0x7149f2ca\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\ts_nop 7\n\t"
                         "v_pk_mul_f32 v[44:45], %1, %1\n\ts_nop 7\n\ts_nop 7\n\tv_mfma_f32_4x4x4_16b_f16 v[44:47], %2, %2, v[44:47]\n\ts_nop 7\n\ts_nop 7\n\tglobal_store_dwordx2 %0, v[44:45], off"
                         : : "v"(outB + 2 * ((size_t)t * n + i)), "v"(xx), "v"(z) : "memory", "v44", "v45", "v46", "v47");
        } else {                               // cvt_pk (the epilogue's pack) -> store
            asm volatile("v_mov_b32 %0, 0\n\tv_cvt_pk_bf16_f32 %0, %2, %2\n\tglobal_store_dword %1, %0, off" : "=&v"(r) : "v"(pa), "v"(x) : "memory");
            asm volatile("v_mov_b32 %0, 0\n\tv_cvt_pk_bf16_f32 %0, %2, %2\n\ts_nop 7\n\ts_nop 7\n\tglobal_store_dword %1, %0, off" : "=&v"(r2) : "v"(pb), "v"(x) : "memory");
        }
        if (r != r2 && r == 12345.678f) outA[0] = r2;      // keep the values live
    }
}

template <int SHAPE>
static int run(const char* name, int blocks, int iters) {
    const int n = blocks * 512;
    float *in, *a, *b;
    hipMalloc(&in, n * 4); hipMalloc(&a, (size_t)n * iters * 8); hipMalloc(&b, (size_t)n * iters * 8);
    float* h = (float*)malloc(n * 4);
    for (int i = 0; i < n; ++i) h[i] = 0.5f + (float)(i % 977) / 977.0f;
    hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    long bad = 0, badhi = 0, total = 0;
    float* ha = (float*)malloc((size_t)n * iters * 8); float* hb = (float*)malloc((size_t)n * iters * 8);
    for (int rep = 0; rep < 8; ++rep) {
        hipMemset(a, 0xff, (size_t)n * iters * 8); hipMemset(b, 0xff, (size_t)n * iters * 8);
        hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(512), 0, 0, in, a, b, iters, n);
        hipDeviceSynchronize();
        hipMemcpy(ha, a, (size_t)n * iters * 8, hipMemcpyDeviceToHost); hipMemcpy(hb, b, (size_t)n * iters * 8, hipMemcpyDeviceToHost);
        const size_t words = (size_t)n * iters * ((SHAPE == 3 || SHAPE == 7) ? 2 : 1);
        for (size_t j = 0; j < words; ++j) {
            ++total;
            const size_t lane = ((SHAPE == 3 || SHAPE == 7) ? j / 2 : j) % 64;
            if (memcmp(&ha[j], &hb[j], 4) != 0) { ++bad; if (lane >= 48) ++badhi; }
        }
    }
    printf("%-28s %ld / %ld stores differ from the safe copy (%ld of them in lanes 48-63)\n", name, bad, total, badhi);
    hipFree(in); hipFree(a); hipFree(b); free(h); free(ha); free(hb);
    return bad != 0;
}

int main() {
    int rc = 0;
    rc |= run<0>("v_exp_f32 -> store", 1024, 64);
    rc |= run<1>("v_rcp_f32 -> store", 1024, 64);
    rc |= run<2>("v_mul_f32 -> store", 1024, 64);
    rc |= run<3>("v_pk_mul_f32 -> store", 1024, 64);
    rc |= run<4>("v_exp -> v_mul -> store", 1024, 64);
    rc |= run<5>("v_cvt_pk_bf16_f32 -> store", 1024, 64);
    rc |= run<6>("v_pk_add_f32 -> v_rcp(high)", 1024, 64);
    rc |= run<7>("v_pk_mul_f32 -> v_mfma SrcC", 1024, 64);
    return rc;
}
