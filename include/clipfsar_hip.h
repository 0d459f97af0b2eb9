/*
 * clipfsar_hip.h -- C ABI of libclipfsar_hip.so: the hand-written HIP (gfx950 / CDNA4) kernels of the
 * CLIP-FSAR episodic-inference hot path.
 *
 * The reference (alibaba-mmai-research/CLIP-FSAR) is pure Python on stock torch.nn modules; it has no FFI for
 * this path.  Each entry point below replaces the library-dispatched op(s) named in its comment
 * (file:line in /root/reference/models/base/few_shot.py unless stated otherwise).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocates; `tensor.data_ptr()`),
 *     including all workspaces: the library allocates no device memory and owns no stream.  Its only process-wide
 *     state is a cache of per-device facts (CU count, the dynamic-LDS limit already raised for a kernel), keyed by
 *     device ordinal, so one process may drive several GPUs; each call acts on the CURRENT HIP device, which must be
 *     the one the operands and the stream belong to;
 *   - `stream` is a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); all work is enqueued on it and
 *     no entry point synchronises with the host;
 *   - return value 0 = success; non-zero = error, message via cfsar_last_error() (thread-local);
 *   - dtype codes: CFSAR_F32 = 0, CFSAR_BF16 = 1, CFSAR_F16 = 2 (where stated);  matrices are row-major with explicit leading dimensions
 *     (in elements).
 */
#ifndef CLIPFSAR_HIP_H
#define CLIPFSAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared between this push and the pop at the end
 * of the header are exported (tests/test_abi.py compares `nm -D` with this file). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define CFSAR_F32 0
#define CFSAR_BF16 1
#define CFSAR_F16 2 /* IEEE half: GEMM output / residual and LayerNorm input only (the residual stream of the bf16 mode) */

#define CFSAR_ACT_NONE 0
#define CFSAR_ACT_QUICKGELU 1 /* x*sigmoid(1.702x), few_shot.py:614-616 */
#define CFSAR_ACT_GELU_ERF 2  /* nn.GELU() exact, few_shot.py:1647 */

typedef void* cfsar_stream_t;

/* library version (major*10000 + minor*100 + patch) and last error text of the calling thread */
int cfsar_version(void);
/* ABI revision: bumped whenever an exported signature changes or is added; a binding compares it with the CFSAR_ABI_VERSION it was
 * written against at load time (clip-fsar_amd/hip.py does) instead of calling through a stale prototype. */
#define CFSAR_ABI_VERSION 9
int cfsar_abi_version(void);
const char* cfsar_last_error(void);

/* ---- N2 test-time frame transform (the step BEFORE the path; reference datasets/base/ssv2_few_shot.py:614-642,
 * datasets/utils/transformations.py:663-716): uint8 frames [T,H,W,3] (device) -> /255 -> bilinear resize to
 * (scale_h, scale_w) with align_corners=False -> crop window [y0, y0+crop) x [x0, x0+crop) -> (v - mean[c]) / std[c]
 * -> out [T,3,crop,crop] fp32.  mean3 / std3 are HOST pointers to 3 floats (DATA.MEAN / DATA.STD). */
int cfsar_preprocess_frames(const uint8_t* frames, float* out, int T, int H, int W, int scale_h, int scale_w, int crop,
                            int y0, int x0, const float* mean3, const float* std3, cfsar_stream_t stream);

/* ---- A2 patch embedding, stage 1: gather non-overlapping PxP patches of NCHW fp32 frames into GEMM rows.
 * Replaces the im2col implied by nn.Conv2d(3, D, kernel=P, stride=P, bias=False) (few_shot.py:659,672-674).
 * frames [F,3,H,W] fp32 -> out [F*(H/P)*(W/P), k_pad] (dtype out_dtype); column k = c*P*P + dy*P + dx,
 * columns >= 3*P*P are zero-filled.  P must be even. */
int cfsar_im2col_patches(const float* frames, void* out, int out_dtype, int F, int H, int W, int P, int k_pad,
                         cfsar_stream_t stream);

/* Round 6 (the fp16_strict numerics mode): cfsar_im2col_patches with each fp32 pixel kept as TWO fp16 words, laid out for ONE three-pass fp16 GEMM
 * against [W_hi | W_hi | W_lo]:  out [F*(H/P)*(W/P), 3 k_pad] fp16, row = [hi | lo | hi], hi = fp16(v), lo = fp16(v - hi) (pad columns zero).
 * conv1 (few_shot.py:659, 672-674) then reaches the stream with ~22-bit operands instead of 11: the patch embedding is a quarter of the fp16
 * mode's logits error (profiles/r06_strict_budget.md) and 0.7 % of the tower's FLOPs. */
int cfsar_im2col_patches_split(const float* frames, void* out, int F, int H, int W, int P, int k_pad, cfsar_stream_t stream);

/* ---- A2 in ONE launch (few_shot.py:659, 672-676) for 16 x 16 and (round 6) 14 x 14 patches and 16-bit operands: conv1 as a GEMM whose rows
 * are gathered straight from the fp32 NCHW frames (no patch matrix), + pos[1 + p], scattered behind each frame's class-token row, and the class-token rows
 * cls + pos[0] themselves:  x[f*ntok + 1 + p, :] = patch(f, p) @ W.T + pos[1 + p],  x[f*ntok, :] = cls + pos[0],  ntok = (H/P)*(W/P) + 1.
 * frames [F,3,H,W] fp32 (8-byte aligned); pos [ntok, D] fp32, cls [D] fp32; x [F*ntok, D] (x_dtype CFSAR_F16: the 16-bit modes' residual stream).
 * W [D, ldw] (w_dtype CFSAR_BF16 | CFSAR_F16):
 *   P = 16: ldw >= 768, column k = c*256 + dy*16 + dx = conv1.weight.reshape(D, 768);
 *   P = 14: ldw >= 704, PADDED ROWS: column k' = (c*14 + dy)*16 + dx for dx < 14, columns with dx in {14, 15} and k' >= 672 zero (a patch row of 14
 *           pixels is 56 bytes: the 64-slot K slices hold four 16-slot rows each; the caller lays the weights out once).
 * Frames are rounded to w_dtype in registers; every output element is bit-identical to cfsar_gemm_ex (row remap, pos residual) on a patch matrix in the
 * same column layout + cfsar_cls_rows_ex (P = 16: cfsar_im2col_patches makes that matrix).  Other patch sizes and fp32 operands keep
 * cfsar_im2col_patches + cfsar_gemm_ex + cfsar_cls_rows_ex.  Replaces: self.conv1(x) ... x + positional_embedding (few_shot.py:672-676). */
int cfsar_patch_embed(const float* frames, const void* W, int w_dtype, const float* pos, const float* cls, void* x, int x_dtype,
                      int F, int H, int Wd, int P, int D, int ldw, cfsar_stream_t stream);

/* ---- A2 stage 3: class-token rows.  x[f*ntok*D + d] = cls[d] + pos[d]  (few_shot.py:675-676). */
int cfsar_cls_rows(float* x, const float* cls, const float* pos, int F, int ntok, int D, cfsar_stream_t stream);
/* Same for a residual stream of dtype x_dtype (CFSAR_F32 | CFSAR_F16). */
int cfsar_cls_rows_ex(void* x, int x_dtype, const float* cls, const float* pos, int F, int ntok, int D, cfsar_stream_t stream);

/* ---- A3 LayerNorm over the last dim (fp32 statistics, biased variance, eps inside the sqrt), few_shot.py:605-611
 * (ln_pre/ln_1/ln_2/ln_post) and :971-977 (context2 pre-norm).  x rows at stride in_stride (elements), out rows at
 * out_stride; out dtype f32 or bf16.  D % 4 == 0, D <= 4096.  In-place (out == x, f32) is allowed. */
int cfsar_layernorm(const float* x, int64_t in_stride, void* out, int64_t out_stride, int out_dtype,
                    const float* weight, const float* bias, int rows, int D, float eps, cfsar_stream_t stream);
/* Same with the input dtype explicit: in_dtype CFSAR_F32 (out F32 | BF16) or CFSAR_F16 (out BF16 | F16 | F32) -- the fp16
 * residual stream of the bf16 mode (statistics and arithmetic stay fp32). */
int cfsar_layernorm_ex(const void* x, int in_dtype, int64_t in_stride, void* out, int64_t out_stride, int out_dtype,
                       const float* weight, const float* bias, int rows, int D, float eps, cfsar_stream_t stream);

/* ---- A5/A6/A8/A11 dense projections: out = act(A . W^T + bias) + residual, MFMA (v_mfma_f32_32x32x16_bf16 for
 * bf16 inputs, v_mfma_f32_32x32x2_f32 for f32 inputs), fp32 accumulation.
 * Replaces nn.Linear / MultiheadAttention in_proj+out_proj / conv1-as-GEMM / `@ proj`
 * (few_shot.py:623,626-628,635,672,686,1046-1053,1646-1650).
 *   A [M,K] (in_dtype, lda), W [N,K] (in_dtype, ldw; nn.Linear weight layout), bias [N] f32 or NULL,
 *   residual f32 (ldr) or NULL, out [.,N] (out_dtype, ldo).
 *   Output row of GEMM row m:  orow = m + (m / row_group) * row_gap + row_off   (row_group == 0: orow = m);
 *   residual row: res_mod > 0 ? (m % res_mod) + res_off : orow.   (Used to scatter patch rows behind the class
 *   token and add the positional embedding in the patch-embed epilogue.)
 *   K % 64 == 0 (bf16) / K % 32 == 0 (f32); N % 4 == 0; A and W 16-byte aligned rows. */
int cfsar_gemm(const void* A, const void* W, void* out, const float* bias, const float* residual, int M, int N,
               int K, int lda, int ldw, int ldo, int ldr, int in_dtype, int out_dtype, int act, int row_group,
               int row_gap, int row_off, int res_mod, int res_off, cfsar_stream_t stream);

/* Same as cfsar_gemm with two extras used by the RN50 tower (N3): the residual may be bf16 / fp16 (res_dtype) and, when
 * relu != 0, max(.,0) is applied LAST (after bias, activation and residual): conv+BN(+identity)+ReLU, few_shot.py:213-226.
 * fp16 operands (in_dtype CFSAR_F16: the tower's fp16 numerics mode) write fp16 or fp32 outputs; bias, residual and ReLU are
 * applied to the fp32 accumulator and the result is rounded ONCE. */
int cfsar_gemm_ex(const void* A, const void* W, void* out, const float* bias, const void* residual, int M, int N,
                  int K, int lda, int ldw, int ldo, int ldr, int in_dtype, int out_dtype, int act, int row_group,
                  int row_gap, int row_off, int res_mod, int res_off, int res_dtype, int relu, cfsar_stream_t stream);

/* ---- N3 ModifiedResNet ("RN50") tower helpers (few_shot.py:182-227, 542-602); activations are NHWC; dtype / out_dtype is
 * CFSAR_BF16, CFSAR_F16 (the tower's fp16 numerics mode) or CFSAR_F32 (validation mode) on every one of them.
 * cfsar_nchw_to_nhwc: frames [F,C,H,W] f32 -> [F,H,W,C] (out_dtype).
 * cfsar_im2col3x3_nhwc: 3x3 / pad 1 / stride 1|2 gather, in [F,H,W,C] -> out [F*Ho*Wo, k_pad], column (ky*3+kx)*C + c,
 *   zeros outside the image and in pad columns (nn.Conv2d(k=3, padding=1) as a GEMM with tap-major weights).
 * cfsar_avgpool2x2_nhwc: nn.AvgPool2d(2).
 * cfsar_attnpool_tokens: AttentionPool2d token build (:446-448): [mean_hw(x) ; x] + positional_embedding. */
int cfsar_nchw_to_nhwc(const float* frames, void* out, int out_dtype, int F, int C, int H, int W, cfsar_stream_t stream);
int cfsar_im2col3x3_nhwc(const void* in, void* out, int dtype, int F, int H, int W, int C, int stride, int k_pad,
                         cfsar_stream_t stream);
int cfsar_avgpool2x2_nhwc(const void* in, void* out, int dtype, int F, int H, int W, int C, cfsar_stream_t stream);
int cfsar_attnpool_tokens(const void* x, const float* pos, void* out, int dtype, int F, int HW, int C,
                          cfsar_stream_t stream);

/* Stem conv1 of the ModifiedResNet (few_shot.py:558-560, 582-586): nn.Conv2d(3, Cout, 3, stride=2, padding=1, bias=False) +
 * folded BatchNorm (+ ReLU when relu != 0), straight from fp32 NCHW frames [F,3,H,W] to NHWC activations
 * out [F, Ho, Wo, Cout] (out_dtype), Ho = (H-1)/2+1.  w fp32 [Cout, 3, 3, 3] (PyTorch layout, BN scale folded in), bias fp32
 * [Cout] or NULL; Cout in {8, 16, 32, 64}.  fp32 arithmetic. */
int cfsar_stem_conv3x3_s2(const float* frames, const float* w, const float* bias, void* out, int out_dtype, int F, int H,
                          int W, int Cout, int relu, cfsar_stream_t stream);

/* AttentionPool2d attention for the single query it keeps (the mean token; few_shot.py:450-469): q [F, C] fp32 (q_proj of
 * token 0, bias included, NOT yet scaled), kv [F*T, 2C] fp32 = [k_proj | v_proj] of all T = HW+1 tokens, out [F, C] fp32 =
 * softmax(scale * q k^T) v per head (C = heads * head_dim, head_dim <= 128, T <= 512); c_proj follows as a cfsar_gemm. */
int cfsar_attnpool_attend(const float* q, const float* kv, float* out, int F, int T, int heads, int head_dim, float scale,
                          cfsar_stream_t stream);

/* nn.Conv2d(C, Cout, 3, padding=1, bias=False) + folded BatchNorm (+ identity) + ReLU of the RN50 tower (few_shot.py:196,
 * 213-226) as ONE launch on 16-bit NHWC activations: the 3x3 patches are gathered inside the GEMM's operand staging (no
 * im2col matrix in HBM), or, for Cin, Cout in {32, 64}, read from an LDS ring of the pixel stream (csrc/conv.hip).
 * in [F,H,W,C] bf16, C a power of two >= 8; W [Cout, ldw] bf16, tap-major columns (ky*3+kx)*C + c, zero-padded to
 * ldw = round_up(9*C, 64); out [F*H*W, ldo] (out_dtype bf16 or fp32); bias fp32 [Cout] or NULL; residual [F*H*W, ldr]
 * (res_dtype) or NULL; relu != 0 applies max(., 0) last.  out_dtype CFSAR_F16 selects the fp16 form: `in` and W are IEEE
 * half too (the tower's fp16 numerics mode). */
int cfsar_conv3x3_nhwc(const void* in, const void* W, void* out, const float* bias, const void* residual, int F, int H,
                       int Wd, int C, int Cout, int ldw, int ldo, int ldr, int out_dtype, int res_dtype, int relu,
                       cfsar_stream_t stream);

/* ---- A5 scaled-dot-product attention of nn.MultiheadAttention for the ViT (no mask, no dropout), head_dim 64.
 * qkv [F*ntok, 3*D] packed as [q | k | v], head h at columns h*64 of each third; out [F*ntok, D].
 * dtype bf16: MFMA kernel (K and V^T of one (frame, head) staged in LDS, single-pass softmax in registers);
 * dtype f32: fp32 VALU kernel (validation mode).  ntok <= 288. */
int cfsar_vit_attention(const void* qkv, void* out, int dtype, int F, int ntok, int D, int heads,
                        cfsar_stream_t stream);

/* The class-token form of the same attention: VisionTransformer.forward keeps only x[:, 0] after the LAST block
 * (few_shot.py:683), so that block needs the attention output of ONE query row per frame.  q: row f at q + f*ldq (elements);
 * k / v: token t of frame f at k / v + (f*ntok + t)*ldkv, head h at columns 64h.  With the packed matrix of
 * cfsar_vit_attention: q = qkv, ldq = ntok*3D, k = qkv + D, v = qkv + 2D, ldkv = 3D.  out [F, D] (row f = frame f).
 * dtype bf16 | fp16 | f32; ntok <= 320; rows 16-byte aligned. */
int cfsar_vit_attention_cls(const void* q, long long ldq, const void* k, const void* v, int ldkv, void* out, int dtype, int F,
                            int ntok, int D, int heads, cfsar_stream_t stream);

/* ---- A15b aux class logits: cos_sim(mean_T(feats), text_train) * scale  (few_shot.py:2937-2939, cos_sim :1115-1124).
 * feats [n_videos, T, E] f32, text [n_cls, E] f32, scale [1] f32 (device), out [n_videos, n_cls] f32. */
int cfsar_class_text_logits(const float* feats, const float* text, const float* scale, float* out, int n_videos,
                            int T, int E, int n_cls, cfsar_stream_t stream);

/* ---- A10/A12 build the temporal-transformer input sequences of a batch of episodes (few_shot.py:2946-2955).
 * feats [B, S+Q, T, E] f32 (support videos first), text_test [n_test, E], support_labels / real_support_labels
 * [B, S] f32.  Output tokens X (row-major [rows, E]):
 *    rows [0, B*Q*T)                       : query sequences, (b, q, t)
 *    rows [B*Q*T, B*Q*T + B*Sp*(T+1))      : support sequences (b, s', 0..T): T frame tokens then the text token,
 * Sp = S, or `way` when merge_before != 0 (class means of frames and of the text rows, classes in ascending label
 * order as torch.unique sorts them).  Returns non-zero if way*shot != S. */
int cfsar_build_sequences(const float* feats, const float* text_test, const float* support_labels,
                          const float* real_support_labels, float* X, int B, int S, int Q, int T, int E, int way,
                          int n_test, int merge_before, cfsar_stream_t stream);

/* ---- A11 attention of Attention_qkv on short sequences (few_shot.py:1056-1073): softmax(q k^T * scale) v per
 * (sequence, head).  qkv [rows, 3*inner] f32 packed [q | k | v]; out [rows, inner].  Sequences: n_a sequences of
 * length len_a starting at row 0, then n_b sequences of length len_b.  len <= 128, head_dim <= 128.
 * causal != 0: keys j > i are masked (the CLIP text transformer's additive -inf mask, few_shot.py:778-784; N1). */
int cfsar_seq_attention(const float* qkv, float* out, int n_a, int len_a, int n_b, int len_b, int heads,
                        int head_dim, float scale, int causal, cfsar_stream_t stream);

/* ---- N1 (init-time text tower) token embedding lookup + positional embedding, CLIP.encode_text few_shot.py:794-796.
 * tokens [n_seq, L] int32, table [vocab, W], pos [L, W] -> out [n_seq*L, W] f32. */
int cfsar_embed_tokens(const int32_t* tokens, const float* table, const float* pos, float* out, int n_seq, int L, int W,
                       int vocab, cfsar_stream_t stream);

/* ---- N1 row gather: out[i] = x[idx[i]] (EOT-token pooling x[arange, text.argmax(-1)], few_shot.py:804). */
int cfsar_gather_rows(const float* x, const int32_t* idx, float* out, int n, int D, int rows_in, cfsar_stream_t stream);

/* ---- N4 EVAL_TEXT branch (few_shot.py:2835-2852) and first half of COMBINE (:2855-2870): per query,
 * softmax_c(scale * cos(mean_T(target feats), class-mean_c(text_test[real_support_labels]))) with both vectors
 * L2-normalised (no epsilon), classes in ascending support-label order.  probs [B, Q, way]. */
int cfsar_text_match_probs(const float* feats, const float* text_test, const float* support_labels,
                           const float* real_support_labels, const float* scale, float* probs, int B, int S, int Q,
                           int T, int E, int way, int n_test, cfsar_stream_t stream);

/* ---- N4 COMBINE (few_shot.py:2921-2926): out = p_text^coff * softmax_c((8 - cum)/8)^(1 - coff) with cum = -visual_logits
 * (out is the reference's `logits` = -cum_dists). */
int cfsar_combine_logits(const float* text_probs, const float* visual_logits, float* out, int n_queries, int way,
                         float text_coff, cfsar_stream_t stream);

/* ---- A12 prototypes: first T tokens of each support sequence, class-mean over shots unless merged before
 * (few_shot.py:2956-2962).  Xs = support part of the context2 output [B, Sp, T+1, E]; protos [B, way, T, E]. */
int cfsar_prototypes(const float* Xs, const float* support_labels, float* protos, int B, int S, int Sp, int T, int E,
                     int way, int merge_before, cfsar_stream_t stream);

/* ---- A13/A14/A15 cos_sim (eps = 0.01 added to the product of norms) -> 1 - sim -> OTAM soft-DTW (lambda 0.5,
 * zero-padded columns, both directions unless single_direct) -> logits = -cum_dists
 * (few_shot.py:1115-1124, 2657-2687, 2970-2990).  Xq [B, Q, T, E], protos [B, way, T, E] f32;
 * logits [B, Q, way]; dists_out (optional, may be NULL) [B, Q, way, T, T].  T <= 32, E % 4 == 0, E <= 2048. */
int cfsar_cos_otam_logits(const float* Xq, const float* protos, float* logits, float* dists_out, int B, int Q,
                          int way, int T, int E, float lambda, int single_direct, cfsar_stream_t stream);

/* ---- A17 per-episode top-1 accuracy (utils/metrics.py:100-138 topks_correct with k = 1, per episode as runs/test_net_few_shot.py:120-147
 * uses it): acc[e] = fraction of episode e's Q queries whose first-maximum class equals target_labels[e, q] (class index as float).
 * logits [episodes, Q, way] f32, target_labels [episodes, Q] f32, acc [episodes] f32. */
int cfsar_episode_top1(const float* logits, const float* target_labels, float* acc, int episodes, int Q, int way, cfsar_stream_t stream);

/* ---- A3 + A5/A6 fused: LayerNorm folded into the GEMM that consumes it (few_shot.py:605-611 with :626-628 / :636-640).
 * out[m,n] = act(sum_k LN(x)[m,k] W[n,k] + bias[n]) computed on the RAW fp16 residual stream, without materialising LN(x):
 *     LN(x) W^T + bias = (x Wg^T - mean_m c_n) / std_m + d_n,   Wg = W diag(gamma),  c_n = sum_k Wg[n,k],
 *     d_n = sum_k beta_k W[n,k] + bias_n      (Wg, c, d are prepared once by the caller; c must be the sum of the ROUNDED Wg).
 * x [M, lda] fp16, Wg [N, ldw] fp16, out [M, ldo] out_dtype (bf16; fp16 in the fp16 numerics mode), cvec / dvec [N] fp32, rowstats [M, 4] fp32 = (mean, std, 1/std, -) of each
 * row of x (cfsar_row_stats / cfsar_ln_stats_finalize).  K % 64 == 0, K >= 128, N % 64 == 0; act = NONE | QUICKGELU. */
int cfsar_gemm_lnfold(const void* x, const void* Wg, void* out, const float* cvec, const float* dvec, const float* rowstats,
                      int M, int N, int K, int lda, int ldw, int ldo, int act, int out_dtype, cfsar_stream_t stream);

/* The QKV form of cfsar_gemm_lnfold with HEAD-BLOCKED output (act = NONE, N = 192 heads: q | k | v, tokens per frame >= 128,
 * M a multiple of tokens): row m = f tokens + t, column n = 64 (which heads + h) + c is written to
 *     out[((f heads + h) tokens + t) * 192 + 64 which + c]
 * -- the q, k and v rows of one (frame, head) form one contiguous 75 KB block (few_shot.py:626-628 produces the packed qkv the
 * attention of :623 splits per head).  cfsar_vit_attention reads such a buffer when called with D = 64, heads = 1 and
 * frames x heads "frames"; it then writes its output blocked as well ([(f heads + h) tokens + t][64]). */
int cfsar_gemm_lnfold_heads(const void* x, const void* Wg, void* out, const float* cvec, const float* dvec, const float* rowstats,
                            int M, int N, int K, int lda, int ldw, int tokens, int heads, cfsar_stream_t stream);

/* cfsar_gemm_lnfold (tokens = 0) / cfsar_gemm_lnfold_heads (tokens > 0) with the row statistics taken STRAIGHT from the producer's
 * partials -- partial [M, slots, 2] = (sum, sum of squares) per 64 columns, as written by cfsar_gemm_residual_stats -- and
 * finalized inside the GEMM (mean, sqrt(biased variance + eps), reciprocal: the arithmetic of cfsar_ln_stats_finalize), so that no
 * kernel runs between the producing and the consuming GEMM of a LayerNorm (few_shot.py:605-611 between :635 and :639, :640 and :626).
 * The in-kernel form exists for K = 64 slots in {768, 1024} (ViT-B/16, ViT-L/14) on the 192-row-tile instances the launcher picks for
 * small M (one or two episodes per call, where the finalize launches are 8 % of the step); in every other case this entry point runs
 * cfsar_ln_stats_finalize into rowstats_ws [M, 4] and then the plain form itself -- same result, one call for the caller. */
int cfsar_gemm_lnfold_partials(const void* x, const void* Wg, void* out, const float* cvec, const float* dvec, const float* partial,
                               int slots, float eps, float* rowstats_ws, int M, int N, int K, int lda, int ldw, int ldo, int act,
                               int out_dtype, int tokens, int heads, cfsar_stream_t stream);

/* ---- A5/A6 residual update + the statistics of the next LayerNorm (few_shot.py:633-635 / :639-640 followed by :636 / :626).
 * x[m,n] = x[m,n] + sum_k A[m,k] W[n,k] + bias[n] in place on the fp16 residual stream (A, W of in_dtype: bf16 or fp16).  If stats_partial != NULL it
 * receives, per row m and 64-column slot s = n / 64, (sum, sum of squares) of the NEW (rounded) x[m, 64 s .. 64 s + 63]:
 * [M, N / 64, 2] fp32.  Deterministic (no atomics).  K % 64 == 0, K >= 128, N % 64 == 0. */
int cfsar_gemm_residual_stats(const void* A, const void* W, void* x, const float* bias, float* stats_partial, int M, int N,
                              int K, int lda, int ldw, int ldx, int in_dtype, cfsar_stream_t stream);

/* The out_proj form of cfsar_gemm_residual_stats whose A operand is the head-blocked attention output
 * A[((f heads + h) tokens + t) * 64 + c], heads = K / 64 (K tile kt of the GEMM = head kt), M a multiple of tokens. */
int cfsar_gemm_residual_stats_heads(const void* A, const void* W, void* x, const float* bias, float* stats_partial, int M, int N,
                                    int K, int ldw, int ldx, int tokens, cfsar_stream_t stream);

/* ---- Round 4: the fp16 numerics mode's forms of the two GEMMs above (VIDEO.HEAD.PRECISION = "fp16"; same reference ops).
 * SPLIT WEIGHTS: a weight matrix is handed over as [N, 2 K] = [hi | lo], hi = fp16(w), lo = fp16(w - hi) (~22 bits per weight); the
 * kernel walks the K tiles of the activation operand twice inside ONE fp32 accumulation chain (A hi^T + A lo^T): twice the MFMA work of
 * the GEMM it is applied to, no second pass over the output.
 * WIDE RESIDUAL: the GEMM result is added to the residual stream in fp32 and rounded ONCE (cfsar_gemm_residual_stats rounds the GEMM
 * result to fp16 first and adds in fp16); with x_lo != NULL the stream carries two fp16 words per element, x = x_hi + x_lo, x_lo the
 * rounding remainder of x_hi: the update reads and writes both, LN-folded consumers read x_hi (and its statistics) only.
 *
 * PER-FRAME LOW-WORD CORRECTION (corr != NULL; frames of corr_tokens >= 128 rows): the part of the weights' fp16 rounding error that is
 * the same for all tokens of a frame -- (token mean of the frame's operand rows) x W_lo^T, a [frames, K] x [K, N] product the caller forms
 * with cfsar_frame_col_means + cfsar_gemm -- is added to every row of its frame inside the GEMM (in normalised units, before the
 * activation, for the LN-folded form): most of what split weights buy at none of the second MFMA pass (profiles/r04_parity_table.md).
 *
 * cfsar_gemm_lnfold_hp: cfsar_gemm_lnfold with fp16 output; wsplit = 1: Wg [N, ldw >= 2 K] split, cvec = sum_k (hi + lo).  Statistics
 * either finalized (rowstats [M, 4], partial = NULL) or the producer's partials (partial [M, slots, 2], rowstats = NULL, rowstats_ws
 * [M, 4] as in cfsar_gemm_lnfold_partials).  corr [ceil(M / |corr_tokens|), N] fp32 or NULL; corr_tokens < 0 selects the RAW-STREAM form of the
 * correction: corr = (token mean of the frame's raw x rows) x W_lo^T, added before the division by the row's std, with cvec = the exact column
 * sums of W gamma (the row mean's share of the low word) -- the caller then needs no pass over x: the stream's per-frame mean follows its
 * updates x += A W^T + b linearly (engine.py: mean_update_gemm).  colmean_out (act = QUICKGELU only; NULL = off):
 * receives the per-frame token means of the OUTPUT, [ceil(M / corr_tokens), N] bf16 -- the c_fc GEMM hands the next GEMM (c_proj) the
 * means its own correction needs without another pass over the hidden; colsum_ws: workspace of (M / 96 + 2) x 2 x N int32 (the sums are taken in fixed point: an episode's
 * result does not depend on the batch it is served in).
 * cfsar_gemm_residual_wide: x = x + A W^T + bias, A [M, lda] fp16, W [N, ldw] fp16 (wsplit = 0) or [N, ldw >= 2 K] split (wsplit = 1),
 * x_hi [M, ldx] fp16 in place, x_lo [M, ldx] fp16 in place or NULL; stats_partial as in cfsar_gemm_residual_stats (of the new x_hi);
 * corr as above.  wsplit = 2 (round 6, the fp16_strict mode's out_proj): BOTH operands carry two words -- A [M, lda >= 2 K] = [a_hi | a_lo]
 * (cfsar_vit_attention_pair's output), W [N, ldw >= 3 K] = [w_hi | w_hi | w_lo]; the kernel accumulates a_hi w_hi + a_lo w_hi + a_hi w_lo in one fp32
 * chain (A's K tiles wrap behind 2 K: the third segment reads a_hi again), three times the MFMA work of the smallest block GEMM. */
int cfsar_gemm_lnfold_hp(const void* x, const void* Wg, void* out, const float* cvec, const float* dvec, const float* rowstats,
                         const float* partial, int slots, float eps, float* rowstats_ws, int M, int N, int K, int lda, int ldw, int ldo,
                         int act, int out_dtype, int wsplit, const float* corr, int corr_tokens, void* colmean_out, void* colsum_ws,
                         cfsar_stream_t stream);
int cfsar_gemm_residual_wide(const void* A, const void* W, void* x_hi, void* x_lo, const float* bias, float* stats_partial, int M,
                             int N, int K, int wsplit, int lda, int ldw, int ldx, const float* corr, int corr_tokens,
                             cfsar_stream_t stream);
/* cfsar_vit_attention on fp16 q / k / v that ALSO writes the per-frame token means of its output, omean [F, D] bf16 (omean[f][64 h + d] =
 * mean over the frame's tokens of out[f tokens + t][64 h + d]): what out_proj's per-frame correction needs, from the one workgroup that sees
 * all output rows of a (frame, head) -- no pass over `out`. */
int cfsar_vit_attention_means(const void* qkv, void* out, void* omean, int F, int ntok, int D, int heads, cfsar_stream_t stream);
/* Round 6 (fp16_strict): cfsar_vit_attention_means whose output keeps TWO fp16 words per element: out_pair [F ntok, 2 D] = [o_hi | o_lo],
 * o_hi = fp16(o), o_lo = fp16(o - o_hi) -- the K = 2 D operand of cfsar_gemm_residual_wide (wsplit = 2), so that out_proj (few_shot.py:635) no
 * longer sees the 11-bit rounding of the attention output.  omean [F, D] bf16 as above (of the unrounded output). */
int cfsar_vit_attention_pair(const void* qkv, void* out_pair, void* omean, int F, int ntok, int D, int heads, cfsar_stream_t stream);
/* out[f][k] = mean over the frame's tokens t of w_t (A[f tokens + t][k] - mu_t), (mu_t, w_t) = (mean, 1 / std) of row t from rowstats
 * [frames tokens, 4] (the token mean of LayerNorm(x) before its affine, few_shot.py:605-611) or (0, 1) when rowstats == NULL.
 * A [frames tokens, lda] fp16, out [frames, K] bf16. */
int cfsar_frame_col_means(const void* A, int lda, const float* rowstats, void* out, int frames, int tokens, int K, cfsar_stream_t stream);
/* Round 5: the per-frame GEMMs of the fp16 numerics mode -- out[f][n] = sum_k A[f][k] W[n][k] (+ bias[n]) (+ res[f][n]) with A [M, K] bf16 the
 * per-frame token means of a block GEMM's operand (cfsar_frame_col_means, cfsar_vit_attention_means, cfsar_gemm_lnfold_hp's colmean_out), W [N, K]
 * bf16 in nn.Linear layout.  out_dtype CFSAR_F32: the low-word correction corr [M, N] fp32 that cfsar_gemm_lnfold_hp / cfsar_gemm_residual_wide
 * add to every row of a frame (W = the weights' second word; bias, res NULL).  out_dtype CFSAR_BF16: out bf16 = A W^T + bias + res, res [M, N] bf16
 * or NULL and allowed to alias out: the residual stream's per-frame mean following the update x += A W^T + b of few_shot.py:637-640.
 * K % 128 == 0, N % 16 == 0.  A row's result does not depend on M (fixed K-summation order): batch-size invariance of the mode. */
int cfsar_frame_gemm(const void* A, const void* W, void* out, const float* bias, const void* res, int M, int N, int K, int out_dtype,
                     cfsar_stream_t stream);
/* out[i] = (float)hi[i] + (float)lo[i], i < n (the two-word stream -> fp32, e.g. in front of ln_post, few_shot.py:683). */
int cfsar_f16_pair_to_f32(const void* hi, const void* lo, float* out, int64_t n, cfsar_stream_t stream);
/* Round 6 (fp16_strict): class token + positional embedding + ln_pre (few_shot.py:675-677) in one pass with NO 16-bit rounding in front of the
 * two-word stream: row (f, t) = (t == 0 ? cls : tok[f (ntok - 1) + t - 1]) + pos[t] -> LayerNorm(ln_w, ln_b; fp32) -> x_hi = fp16(y),
 * x_lo = fp16(y - x_hi).  tok [F (ntok - 1), D] fp32 = the patch-embed GEMM's output (cfsar_im2col_patches_split + cfsar_gemm), cls [D],
 * pos [ntok, D], x_hi / x_lo [F ntok, D] fp16.  D % 4 == 0, D <= 1024. */
int cfsar_embed_finish_pair(const float* tok, const float* cls, const float* pos, const float* ln_w, const float* ln_b, void* x_hi, void* x_lo,
                            int F, int ntok, int D, float eps, cfsar_stream_t stream);
/* dst[r][0 .. row_bytes) = src[r][0 .. row_bytes), rows at byte strides src_stride / dst_stride (row_bytes % 4 == 0): the
 * class-token rows x[:, 0, :] of few_shot.py:683 and their statistics, gathered for the last block's class-token-only tail. */
int cfsar_copy_rows_strided(const void* src, int64_t src_stride, void* dst, int64_t dst_stride, int rows, int row_bytes,
                            cfsar_stream_t stream);

/* rowstats[m] = (mean, std, 1/std, 0) with std = sqrt(biased variance + eps) from the partials above (D = row length). */
int cfsar_ln_stats_finalize(const float* partial, float* rowstats, int M, int slots, int D, float eps, cfsar_stream_t stream);
/* The same statistics directly from fp16 rows x [M, ld] (first LN-folded GEMM of a tower: its input comes from ln_pre). */
int cfsar_row_stats(const void* x, float* rowstats, int M, int D, int ld, float eps, cfsar_stream_t stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* CLIPFSAR_HIP_H */
