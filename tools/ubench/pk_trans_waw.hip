// Microtest (gfx950), round 3 of the stale-lanes hunt (docs/history/design_r01-r03.md "A fault worth recording"): a quarter-rate transcendental
// (v_exp_f32 / v_rcp_f32) is still executing when a PACKED fp32 instruction writes (WAW) or reads (RAW) the same register as
// the HIGH or LOW half of its 64-bit operand.  The failing builds of csrc/gemm_vit.hip had exactly this neighbourhood in their
// epilogues (v_rcp_f32 v137 ... v_pk_mul_f32 v[136:137]) and lost lanes 48-63 of the HIGH register of one pair -- the lanes a
// 4-pass instruction writes last.  Earlier microtests covered trans-source WAR and VALU -> MFMA wait states with ONE kind of
// instruction stream per SIMD; here the two waves of a SIMD run DIFFERENT streams (test wave next to a transcendental-heavy or an
// MFMA-heavy partner), which is the condition under which the fault appeared in the product ("only with a second kernel on the chip").
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_trans_waw.hip -o tools/ubench/bin/pk_trans_waw && tools/ubench/bin/pk_trans_waw
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// FILL independent VALU instructions between the transcendental and the packed instruction
#define FILL0 ""
#define FILL1 "v_mov_b32 v30, v31\n\t"
#define FILL2 FILL1 FILL1
#define FILL3 FILL2 FILL1
#define FILL4 FILL2 FILL2
#define FILL6 FILL4 FILL2

// SHAPE 0: WAW high: v_exp v11 ; fill ; v_pk_mul v[10:11] = a * b          expect v11 == a.y * b.y in every lane
// SHAPE 1: WAW low : v_exp v10 ; fill ; v_pk_mul v[10:11]                   expect v10 == a.x * b.x
// SHAPE 2: RAW high: v_rcp v11 ; fill(>= 1: the documented trans -> VALU wait state) ; v_pk_mul v[12:13] = v[10:11] * b
// SHAPE 3: WAW high with a plain multiply as the writer (control): v_exp v11 ; fill ; v_mul v11 = a.y * b.y
// SHAPE 4: two transcendentals back to back into the pair, then the packed overwrite (the GELU neighbourhood)
#define BODY(FILL, SHAPE)                                                                                          \
    if constexpr (SHAPE == 0)                                                                                      \
        asm volatile("v_exp_f32 v11, %6\n\t" FILL "v_pk_mul_f32 v[10:11], %2, %3\n\tv_mov_b32 %0, v10\n\tv_mov_b32 %1, v11" \
                     : "=v"(lo), "=v"(hi) : "v"(a), "v"(b), "v"(a), "v"(b), "v"(x) : "v10", "v11", "v30", "v31");   \
    else if constexpr (SHAPE == 1)                                                                                 \
        asm volatile("v_exp_f32 v10, %6\n\t" FILL "v_pk_mul_f32 v[10:11], %2, %3\n\tv_mov_b32 %0, v10\n\tv_mov_b32 %1, v11" \
                     : "=v"(lo), "=v"(hi) : "v"(a), "v"(b), "v"(a), "v"(b), "v"(x) : "v10", "v11", "v30", "v31");   \
    else if constexpr (SHAPE == 2)                                                                                 \
        asm volatile("v_mov_b32 v10, %4\n\tv_rcp_f32 v11, %6\n\tv_nop\n\t" FILL                                    \
                     "v_pk_mul_f32 v[12:13], v[10:11], %3\n\tv_mov_b32 %0, v12\n\tv_mov_b32 %1, v13"               \
                     : "=v"(lo), "=v"(hi) : "v"(a), "v"(b), "v"(a.x), "v"(b), "v"(x) : "v10", "v11", "v12", "v13", "v30", "v31"); \
    else if constexpr (SHAPE == 3)                                                                                 \
        asm volatile("v_exp_f32 v11, %6\n\t" FILL "v_mul_f32 v11, %4, %5\n\tv_mov_b32 %0, v11\n\tv_mov_b32 %1, v11" \
                     : "=v"(lo), "=v"(hi) : "v"(a), "v"(b), "v"(a.y), "v"(b.y), "v"(x) : "v10", "v11", "v30", "v31"); \
    else                                                                                                           \
        asm volatile("v_exp_f32 v10, %6\n\tv_exp_f32 v11, %6\n\t" FILL "v_pk_mul_f32 v[10:11], %2, %3\n\tv_mov_b32 %0, v10\n\tv_mov_b32 %1, v11" \
                     : "=v"(lo), "=v"(hi) : "v"(a), "v"(b), "v"(a), "v"(b), "v"(x) : "v10", "v11", "v30", "v31");

typedef float f2 __attribute__((ext_vector_type(2)));

// PARTNER: what the odd waves of the workgroup do while the even waves run the test: 0 = the test too, 1 = transcendentals only,
// 2 = back-to-back MFMAs.  512 threads, 2 workgroups per CU -> 4 waves per SIMD.
template <int SHAPE, int NF, int PARTNER>
__global__ __launch_bounds__(512) void k(const float* in, unsigned* bad, float* sink, int iters) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    const int wave = threadIdx.x >> 6;
    float x = in[tid & 65535];
    if (PARTNER != 0 && (wave & 1)) {
        if constexpr (PARTNER == 1) {
            float s = x;
            for (int t = 0; t < iters * 8; ++t) asm volatile("v_exp_f32 %0, %0\n\tv_rcp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\tv_rcp_f32 %0, %0" : "+v"(s));
            if (s == 123.f) sink[0] = s;
        } else {
            f32x16 acc = {};
            f16x8 fa = {(_Float16)x, 1, 2, 3, 4, 5, 6, 7}, fb = {(_Float16)1, 1, 1, 1, 1, 1, 1, (_Float16)x};
            for (int t = 0; t < iters * 2; ++t) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, acc, 0, 0, 0);
            }
            if (acc[0] == 123.f) sink[0] = acc[3];
        }
        return;
    }
    unsigned nbad = 0;
    for (int t = 0; t < iters; ++t) {
        const f2 a = {1.25f + (float)(t & 7), 2.5f + x}, b = {3.0f, 0.5f + (float)(t & 3)};
        x = x * 0.999f + 0.001f;                       // exp2(x) in [1, 2]: never equal to a product below
        float lo, hi;
        if constexpr (NF == 0) { BODY(FILL0, SHAPE) }
        else if constexpr (NF == 1) { BODY(FILL1, SHAPE) }
        else if constexpr (NF == 2) { BODY(FILL2, SHAPE) }
        else if constexpr (NF == 3) { BODY(FILL3, SHAPE) }
        else if constexpr (NF == 4) { BODY(FILL4, SHAPE) }
        else { BODY(FILL6, SHAPE) }
        float elo, ehi;
        if constexpr (SHAPE == 2) { elo = a.x * b.x; ehi = __builtin_amdgcn_rcpf(x) * b.y; }
        else if constexpr (SHAPE == 3) { elo = a.y * b.y; ehi = a.y * b.y; }
        else { elo = a.x * b.x; ehi = a.y * b.y; }
        if (lo != elo || hi != ehi) ++nbad;
    }
    if (nbad) atomicAdd(&bad[(threadIdx.x & 63) >> 4], nbad);        // histogram by quarter of the wave (lanes 0-15, ..., 48-63)
}

template <int SHAPE, int NF, int PARTNER>
static void run(const float* in, unsigned* bad, float* sink) {
    unsigned h[4] = {0, 0, 0, 0};
    (void)hipMemset(bad, 0, 16);
    hipLaunchKernelGGL((k<SHAPE, NF, PARTNER>), dim3(512), dim3(512), 0, 0, in, bad, sink, 4096);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    const char* shapes[5] = {"WAW hi (exp -> pk_mul)", "WAW lo (exp -> pk_mul)", "RAW hi (rcp -> pk_mul src)", "WAW (exp -> v_mul) control", "WAW pair (exp,exp -> pk_mul)"};
    const char* partners[3] = {"same", "trans", "mfma"};
    printf("%-30s fill %d partner %-5s : wrong per lane quarter %u %u %u %u %s\n", shapes[SHAPE], NF, partners[PARTNER], h[0], h[1], h[2], h[3],
           (h[0] | h[1] | h[2] | h[3]) ? "<-- FAULT" : "");
}

template <int SHAPE, int PARTNER>
static void run_fills(const float* in, unsigned* bad, float* sink) {
    run<SHAPE, 0, PARTNER>(in, bad, sink);
    run<SHAPE, 1, PARTNER>(in, bad, sink);
    run<SHAPE, 2, PARTNER>(in, bad, sink);
    run<SHAPE, 3, PARTNER>(in, bad, sink);
    run<SHAPE, 4, PARTNER>(in, bad, sink);
    run<SHAPE, 6, PARTNER>(in, bad, sink);
}

int main() {
    float *in, *sink; unsigned* bad;
    float* h = (float*)malloc(65536 * 4);
    for (int i = 0; i < 65536; ++i) h[i] = 0.01f + (float)((i * 2654435761u) % 977u) * 0.001f;
    (void)hipMalloc(&in, 65536 * 4); (void)hipMalloc(&sink, 4); (void)hipMalloc(&bad, 16);
    (void)hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    run_fills<0, 0>(in, bad, sink); run_fills<0, 1>(in, bad, sink); run_fills<0, 2>(in, bad, sink);
    run_fills<1, 0>(in, bad, sink); run_fills<1, 1>(in, bad, sink);
    run_fills<2, 0>(in, bad, sink); run_fills<2, 1>(in, bad, sink); run_fills<2, 2>(in, bad, sink);
    run_fills<3, 0>(in, bad, sink); run_fills<3, 1>(in, bad, sink);
    run_fills<4, 0>(in, bad, sink); run_fills<4, 1>(in, bad, sink); run_fills<4, 2>(in, bad, sink);
    return 0;
}
