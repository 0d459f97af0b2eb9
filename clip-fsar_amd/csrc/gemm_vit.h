// Internal interface between the cfsar_gemm_ex dispatcher (gemm.hip) and the persistent ViT GEMM kernel (gemm_vit.hip).
#pragma once
#include <hip/hip_runtime.h>

struct VitGemmArgs {          // kernel argument block
    const char* A;            // activations [M, lda] bf16
    const char* W;            // weights [N, ldw] bf16 (nn.Linear layout)
    void* out;                // [M, ldo] bf16 (no residual) or fp16 (residual stream)
    const float* bias;        // [N]
    const void* res;          // fp16 residual [M, ldr] or NULL
    const float* rowstats;    // LN-folded mode: [M, 4] = (mean, std, 1 / std, -) per row of A, else NULL
    const float* cvec;        // LN-folded mode: c_n = sum_k W'_nk, [N]
    // LN-folded mode, optional: the statistics come straight from the producer GEMM's partials [M, part_slots, 2] (sum, sum of squares per
    // 64 columns) and are finalized inside this kernel (no cfsar_ln_stats_finalize launch); rowstats is NULL then.  part_slots in {12, 16}.
    const float* part;
    int part_slots;
    float part_invD, part_eps;
    float* stats_out;         // residual mode, optional: [M, stats_slots, 2] partial (sum, sum of squares) of the stored rows
    int stats_slots;          // N / 64
    int M, N, K;
    int lda, ldw, ldo, ldr;
    int act;
    int relu;                 // bf16 residual instance only (RN50 conv3 + identity): ReLU applied last
    int tiles_n, ntiles;      // 64 miw x 256 output tiles
    int miw;                  // 4 = 256-row tiles, 3 = 192-row tiles (small M: fewer wasted rounds of the persistent grid)
    int group, colfast;       // tile walk inside an XCD's range (see tile_of)
    // head-blocked layouts (tokens per frame T >= 128; 0 = row-major).  hb_tokens / hb_heads: the LN-folded QKV instance writes
    // out[((f H + h) T + t) * 192 + 64 which + c] for row m = f T + t, column n = 64 (which H + h) + c (which = q / k / v), i.e. 75 KB
    // contiguous per (frame, head) -- what the attention kernel reads.  ha_tokens: the residual instance reads its A operand from
    // A[((f H + h) T + t) * 64 + c] (K tile kt = head kt), the attention kernel's output in the same blocking.
    int hb_tokens, hb_heads, ha_tokens;
    // Round 4 (the fp16 numerics mode).  nka: K tiles of A.  nka < K / 64 = "split weights": W is [N, 2 ka] = [W_hi | W_lo] (the fp16
    // rounding of the fp32 weight and the fp16 rounding of its remainder) and K tile kt >= nka of A is K tile kt - nka again, i.e. the
    // kernel accumulates A W_hi^T + A W_lo^T in one fp32 chain: the weight carries ~22 bits at twice the MFMA work.
    // res_lo (wide residual instance, MODE 6): second fp16 word of the residual stream (the rounding remainder of the first) or NULL.
    int nka;
    void* res_lo;
    // Per-frame low-word correction (round 4, fp16-output instances): corr [ceil(M / corr_tokens), N] fp32 is added to every row of its frame
    // -- corr[f] = (token mean of frame f's operand rows) x W_lo^T, the part of the weights' fp16 rounding error that is common to a frame's
    // tokens.  In the LN-folded instances it is added in normalised units (before the activation).  It rides in the tail MFMA: the k
    // slot 3 of the hi == 0 lanes carries frame f0 = first row of the wave's tile / corr_tokens, that of the hi == 1 lanes frame f0 + 1
    // (corr_tokens >= rows of a wave's tile: a tile spans at most two frames).  NULL = off.
    const float* corr;
    int corr_tokens;
    int corr_raw;             // LN-folded instances: 1 = corr is in RAW-stream units (added before the division by the row's std), 0 = normalised units
    // Per-frame column sums of the OUTPUT (the fp16 LN-folded QuickGELU instance = c_fc, whose output is c_proj's operand: its per-frame
    // token means feed c_proj's correction without another pass over the 1.5 GB hidden).  colsum [ceil(M / rows of a wave tile), 2, N] fp16:
    // for the wave tile starting at row r0, slot 0 = sum over its rows in frame f0 = r0 / corr_tokens, slot 1 = those in f0 + 1.  NULL = off.
    void* colsum;
#ifdef CFSAR_DEV
    int dbg;                  // ablations: 4 = no epilogue, 8 = every workgroup reads tile (0, 0), 16 = epilogue without its global stores,
                              // 128 = start-time stagger: workgroup b sleeps ((b >> 3) & 31) * stagger_unit * 64 cycles before its first tile
    int stagger_unit;
    long long* trace;         // NULL or [grid][64 tiles][4]: s_memrealtime (100 MHz) at tile start, K loop end, epilogue end, spare
#endif
};

struct VitGemmCall {          // host-side request
    const void* A;
    const void* W;
    void* out;
    const float* bias;
    const void* res;
    const float* rowstats;
    const float* cvec;
    float* stats_out;
    const float* part;        // see VitGemmArgs (NULL: rowstats)
    int part_slots;
    float part_eps;
    int M, N, K, lda, ldw, ldo, ldr;
    int in_dtype, out_dtype, res_dtype, act, relu;      // in_dtype: A / W (bf16, or fp16 in the fp16 numerics mode and LN-folded)
    int opath, store;         // operand path (0 register-staged, 1 LDS-DMA), store policy (0 default, 1 nt, 2 sc1; dev builds)
    int group, colfast;
    int dbg;
    int hb_tokens, hb_heads, ha_tokens;
    int ka = 0;               // K of A when the weights are split ([N, 2 ka]); 0 = K
    int wide = 0;             // residual call: fp32 residual add before the ONE rounding to the fp16 stream (MODE 6), res_lo optional
    void* res_lo = nullptr;
    const float* corr = nullptr;    // see VitGemmArgs
    int corr_tokens = 0;
    int corr_raw = 0;
    void* colsum = nullptr;
    int* out_miw = nullptr;         // receives the tile form the launcher picked (4: 128-row wave tiles, 3: 96-row)
};

// 0 = launched, > 0 = error (cfsar_last_error), -2 = outside this kernel's contract (caller falls back)
int cfsar_gemm_vit_try(const VitGemmCall& c, hipStream_t s);
// operand path the policy gives a launch with this K (0 register-staged, 1 LDS-DMA, 2 LDS-DMA issued one barrier earlier, 4 = the
// two-workgroups-per-CU kernel of gemm_vit4.hip)
int cfsar_vit_policy_opath(int K);
// gemm_vit4.hip: the launch `a` (as cfsar_gemm_vit_try filled it) on 192 x 128 tiles, two 4-wave workgroups per CU; -2 = not covered
int cfsar_gemm_vit4_launch(const VitGemmArgs& a, int mode, bool f16io, int store, hipStream_t s);
// gemm_vit1w.hip: the same launch with one wave per SIMD (4 waves, 128 x 128 wave tiles, 256 x 256 tiles); opath 5; -2 = not covered
int cfsar_gemm_vit1w_launch(const VitGemmArgs& a, int mode, bool f16io, int store, hipStream_t s);
