// The few-shot tail of CNN_OTAM_CLIPFSAR.forward (eval default branch, few_shot.py:2932-2990), fp32 throughout:
// aux class logits, sequence building (text-token concat, optional class merge), short-sequence attention of the
// temporal transformer, prototypes, and cosine + OTAM -> logits.  All kernels take a batch of episodes.
#include "common.h"

namespace {

constexpr int MAX_S = 128;   // max support videos per episode handled by the label-rank helper

// rank[s] = number of distinct label values smaller than labels[s]  (torch.unique sorts ascending: few_shot.py:2950)
// returns number of distinct labels.  Executed by one thread; S is tiny (way*shot).
__device__ int label_ranks(const float* labels, int S, int* rank) {
    int distinct = 0;
    for (int s = 0; s < S; ++s) {
        int r = 0;
        bool first = true;
        for (int t = 0; t < S; ++t) {
            if (labels[t] < labels[s]) {
                bool seen = false;
                for (int u = 0; u < t; ++u)
                    if (labels[u] == labels[t]) { seen = true; break; }
                if (!seen) ++r;
            }
            if (t < s && labels[t] == labels[s]) first = false;
        }
        rank[s] = r;
        if (first) ++distinct;
    }
    return distinct;
}

// ---- A15b  cos_sim(mean_T(feats[v]), text) * scale    (few_shot.py:2937-2939; cos_sim :1115-1124)
__global__ __launch_bounds__(256) void class_text_logits_kernel(const float* __restrict__ feats,
                                                                const float* __restrict__ text,
                                                                const float* __restrict__ scale,
                                                                float* __restrict__ out, int T, int E, int n_cls) {
    // grid (video, group of 16 classes): every workgroup rebuilds the video's frame mean (T x E floats, L2-resident), then each
    // wave takes 4 classes at once so that their row loads overlap (a one-class-at-a-time loop is a chain of global latencies).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* mean = reinterpret_cast<float*>(smem);       // [E]
    __shared__ float red[4];
    const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* fv = feats + (size_t)v * T * E;
    float ss = 0.f;
    for (int e = tid; e < E; e += 256) {
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += fv[(size_t)t * E + e];
        a = a / (float)T;
        mean[e] = a;
        ss += a * a;
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float xnorm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    const float sc = scale[0];
    const int c0 = blockIdx.y * 16 + wave * 4;
    float dot[4] = {0.f, 0.f, 0.f, 0.f}, yy[4] = {0.f, 0.f, 0.f, 0.f};
    for (int e = lane; e < E; e += 64) {
        const float m = mean[e];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j < n_cls ? c0 + j : n_cls - 1;
            const float y = text[(size_t)c * E + e];
            dot[j] = fmaf(m, y, dot[j]);
            yy[j] = fmaf(y, y, yy[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float d = wave_sum(dot[j]), y2 = wave_sum(yy[j]);
        if (lane == 0 && c0 + j < n_cls) out[(size_t)v * n_cls + c0 + j] = d / (xnorm * sqrtf(y2) + 0.01f) * sc;
    }
}

// ---- A10/A12  sequence building (few_shot.py:2946-2955).  One workgroup per output token row.
__global__ __launch_bounds__(128) void build_sequences_kernel(const float* __restrict__ feats,
                                                              const float* __restrict__ text_test,
                                                              const float* __restrict__ support_labels,
                                                              const float* __restrict__ real_labels,
                                                              float* __restrict__ X, int B, int S, int Q, int T, int E,
                                                              int way, int n_test, int merge_before) {
    __shared__ int rank[MAX_S];
    const int Sp = merge_before ? way : S;
    const int q_rows = B * Q * T;
    const int row = blockIdx.x, tid = threadIdx.x;
    float* xr = X + (size_t)row * E;
    if (row < q_rows) {                                   // query token (b, q, t)
        const int t = row % T, bq = row / T;
        const int q = bq % Q, b = bq / Q;
        const float* src = feats + (((size_t)b * (S + Q) + S + q) * T + t) * E;
        for (int e = tid; e < E; e += 128) xr[e] = src[e];
        return;
    }
    const int r2 = row - q_rows;
    const int tt = r2 % (T + 1), bs = r2 / (T + 1);
    const int sp = bs % Sp, b = bs / Sp;
    const float* lab = support_labels + (size_t)b * S;
    const float* rl = real_labels + (size_t)b * S;
    if (!merge_before) {
        const float* src;
        if (tt < T) {
            src = feats + (((size_t)b * (S + Q) + sp) * T + tt) * E;
        } else {
            const int cls = (int)rl[sp];                   // .long() truncation (few_shot.py:2946)
            // an out-of-range class id is an IndexError in the reference (:2946).  A kernel cannot raise: the row is poisoned
            // with NaN (never silently repaired) and the host wrapper validates the labels (models/base/few_shot.py)
            src = (cls >= 0 && cls < n_test) ? text_test + (size_t)cls * E : nullptr;
        }
        for (int e = tid; e < E; e += 128) xr[e] = src ? src[e] : __builtin_nanf("");
        return;
    }
    if (tid == 0) label_ranks(lab, S, rank);
    __syncthreads();
    int cnt = 0;
    for (int s = 0; s < S; ++s) cnt += (rank[s] == sp);
    // cnt == 0: the episode has fewer than `way` distinct labels (the reference would build fewer prototypes and fail on
    // shapes downstream): NaN row, validated on the host
    const float inv = cnt > 0 ? 1.0f / (float)cnt : __builtin_nanf("");
    for (int e = tid; e < E; e += 128) {
        float a = 0.f;
        for (int s = 0; s < S; ++s) {
            if (rank[s] != sp) continue;
            if (tt < T) {
                a += feats[(((size_t)b * (S + Q) + s) * T + tt) * E + e];
            } else {
                const int cls = (int)rl[s];
                a += (cls >= 0 && cls < n_test) ? text_test[(size_t)cls * E + e] : __builtin_nanf("");
            }
        }
        xr[e] = a * inv;
    }
}

// ---- A11 attention on short sequences (few_shot.py:1056-1073); with causal != 0 it is also the masked attention of
// the CLIP text transformer (N1: few_shot.py:778-784 additive -inf mask above the diagonal).  One 128-thread workgroup
// per (sequence, head); thread i owns query position i (len <= 128).
__global__ __launch_bounds__(128) void seq_attention_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            int n_a, int len_a, int n_b, int len_b, int heads, int hd,
                                                            float scale, int causal) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int seq = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    int start, L;
    if (seq < n_a) { start = seq * len_a; L = len_a; }
    else { start = n_a * len_a + (seq - n_a) * len_b; L = len_b; }
    const int inner = heads * hd;
    const size_t ld = (size_t)3 * inner;
    const int hp = hd + 1;                           // padded row: conflict-free column walks
    float* sq = reinterpret_cast<float*>(smem);      // [L][hp]
    float* sk = sq + L * hp;
    float* sv = sk + L * hp;
    float* sp = sv + L * hp;                         // [L][L+1] probabilities
    const int lp = L + 1;
    for (int idx = tid; idx < L * hd; idx += 128) {
        const int i = idx / hd, d = idx - i * hd;
        const float* rowp = qkv + (size_t)(start + i) * ld + h * hd + d;
        sq[i * hp + d] = rowp[0];
        sk[i * hp + d] = rowp[inner];
        sv[i * hp + d] = rowp[2 * inner];
    }
    __syncthreads();
    if (tid < L) {                                    // thread = query position i
        const int jmax = causal ? tid + 1 : L;
        float mx = -1e30f;
        for (int j = 0; j < jmax; ++j) {
            float dot = 0.f;
            for (int d = 0; d < hd; ++d) dot = fmaf(sq[tid * hp + d], sk[j * hp + d], dot);
            dot *= scale;
            sp[tid * lp + j] = dot;
            mx = fmaxf(mx, dot);
        }
        float sum = 0.f;
        for (int j = 0; j < jmax; ++j) {
            const float e = expf(sp[tid * lp + j] - mx);
            sp[tid * lp + j] = e;
            sum += e;
        }
        const float inv = 1.0f / sum;
        for (int j = 0; j < jmax; ++j) sp[tid * lp + j] *= inv;
        for (int j = jmax; j < L; ++j) sp[tid * lp + j] = 0.f;
    }
    __syncthreads();
    for (int idx = tid; idx < L * hd; idx += 128) {
        const int i = idx / hd, d = idx - i * hd;
        const int jmax = causal ? i + 1 : L;
        float a = 0.f;
        for (int j = 0; j < jmax; ++j) a = fmaf(sp[i * lp + j], sv[j * hp + d], a);
        out[(size_t)(start + i) * inner + h * hd + d] = a;
    }
}

// ---- N1 token embedding + positional embedding (few_shot.py:794-796) and row gather (EOT pooling, :804)
__global__ __launch_bounds__(128) void embed_tokens_kernel(const int* __restrict__ tokens, const float* __restrict__ table,
                                                           const float* __restrict__ pos, float* __restrict__ out, int L,
                                                           int W, int vocab) {
    const int row = blockIdx.x;                       // (sequence, position)
    int t = tokens[row];
    t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
    const float* e = table + (size_t)t * W;
    const float* pe = pos + (size_t)(row % L) * W;
    for (int d = threadIdx.x; d < W; d += 128) out[(size_t)row * W + d] = e[d] + pe[d];
}

__global__ __launch_bounds__(128) void gather_rows_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                                          float* __restrict__ out, int D, int rows_in) {
    int r = idx[blockIdx.x];
    r = r < 0 ? 0 : (r >= rows_in ? rows_in - 1 : r);
    for (int d = threadIdx.x; d < D; d += 128) out[(size_t)blockIdx.x * D + d] = x[(size_t)r * D + d];
}

// ---- N4 text-matching logits of the EVAL_TEXT / COMBINE eval branches (few_shot.py:2835-2849, 2855-2870):
// softmax_c( scale * <norm(mean_T target feats), norm(class-mean of text_test[real_support_labels])> ).
// One workgroup per (b, q); classes in ascending label order.
__global__ __launch_bounds__(256) void text_match_kernel(const float* __restrict__ feats, const float* __restrict__ text_test,
                                                         const float* __restrict__ support_labels,
                                                         const float* __restrict__ real_labels,
                                                         const float* __restrict__ scale, float* __restrict__ probs, int S,
                                                         int Q, int T, int E, int way, int n_test) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* img = reinterpret_cast<float*>(smem);           // [E]
    float* txt = img + E;                                  // [E]
    __shared__ int rank[MAX_S];
    __shared__ float red[4];
    __shared__ float logit[64];
    const int bq = blockIdx.x, b = bq / Q, q = bq - b * Q;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) label_ranks(support_labels + (size_t)b * S, S, rank);
    const float* fq = feats + (((size_t)b * (S + Q) + S + q) * T) * E;
    float ss = 0.f;
    for (int e = tid; e < E; e += 256) {
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += fq[(size_t)t * E + e];
        a /= (float)T;
        img[e] = a;
        ss += a * a;
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float inorm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    const float* rl = real_labels + (size_t)b * S;
    for (int c = 0; c < way; ++c) {
        __syncthreads();
        int cnt = 0;
        for (int s = 0; s < S; ++s) cnt += (rank[s] == c);
        float tt = 0.f, dot = 0.f;
        for (int e = tid; e < E; e += 256) {
            float a = 0.f;
            for (int s = 0; s < S; ++s)
                if (rank[s] == c) {
                    const int cls = (int)rl[s];                                  // out of range: NaN, see build_sequences_kernel
                    a += (cls >= 0 && cls < n_test) ? text_test[(size_t)cls * E + e] : __builtin_nanf("");
                }
            a /= (float)cnt;                                                          // cnt == 0 -> NaN (0 / 0)
            tt += a * a;
            dot += a * img[e];
        }
        tt = wave_sum(tt);
        dot = wave_sum(dot);
        __syncthreads();
        if (lane == 0) { red[wave] = tt; txt[wave] = dot; }
        __syncthreads();
        if (tid == 0) {
            const float tn = sqrtf(red[0] + red[1] + red[2] + red[3]);
            const float d = txt[0] + txt[1] + txt[2] + txt[3];
            logit[c] = scale[0] * (d / inorm / tn);
        }
    }
    __syncthreads();
    if (tid == 0) {
        float mx = -1e30f, sum = 0.f;
        for (int c = 0; c < way; ++c) mx = fmaxf(mx, logit[c]);
        for (int c = 0; c < way; ++c) sum += expf(logit[c] - mx);
        for (int c = 0; c < way; ++c) probs[(size_t)bq * way + c] = expf(logit[c] - mx) / sum;
    }
}

// ---- N4 COMBINE (few_shot.py:2921-2926): -( p_text^coff * softmax_c((8 - cum)/8)^(1-coff) ), cum = -visual logits.
__global__ __launch_bounds__(64) void combine_kernel(const float* __restrict__ text_probs, const float* __restrict__ vis_logits,
                                                     float* __restrict__ out, int way, float coff) {
    const int bq = blockIdx.x;
    if (threadIdx.x != 0) return;
    const float* p = text_probs + (size_t)bq * way;
    const float* v = vis_logits + (size_t)bq * way;
    float mx = -1e30f, sum = 0.f;
    for (int c = 0; c < way; ++c) mx = fmaxf(mx, (8.0f + v[c]) / 8.0f);          // (8 - cum)/8 with cum = -v
    for (int c = 0; c < way; ++c) sum += expf((8.0f + v[c]) / 8.0f - mx);
    for (int c = 0; c < way; ++c) {
        const float soft = expf((8.0f + v[c]) / 8.0f - mx) / sum;
        out[(size_t)bq * way + c] = powf(p[c], coff) * powf(soft, 1.0f - coff);   // logits = -cum_dists
    }
}

// ---- A12 prototypes (few_shot.py:2956-2962).  One workgroup per (b, class, t).
__global__ __launch_bounds__(128) void prototypes_kernel(const float* __restrict__ Xs,
                                                         const float* __restrict__ support_labels,
                                                         float* __restrict__ protos, int S, int Sp, int T, int E, int way,
                                                         int merge_before) {
    __shared__ int rank[MAX_S];
    const int t = blockIdx.x % T, bc = blockIdx.x / T;
    const int c = bc % way, b = bc / way, tid = threadIdx.x;
    float* pr = protos + (((size_t)b * way + c) * T + t) * E;
    if (merge_before) {
        const float* src = Xs + (((size_t)b * Sp + c) * (T + 1) + t) * E;
        for (int e = tid; e < E; e += 128) pr[e] = src[e];
        return;
    }
    if (tid == 0) label_ranks(support_labels + (size_t)b * S, S, rank);
    __syncthreads();
    int cnt = 0;
    for (int s = 0; s < S; ++s) cnt += (rank[s] == c);
    const float inv = cnt > 0 ? 1.0f / (float)cnt : __builtin_nanf("");      // see build_sequences_kernel
    for (int e = tid; e < E; e += 128) {
        float a = 0.f;
        for (int s = 0; s < S; ++s)
            if (rank[s] == c) a += Xs[(((size_t)b * Sp + s) * (T + 1) + t) * E + e];
        pr[e] = a * inv;
    }
}

// ---- A13/A14/A15 cos_sim + OTAM both directions -> logits.  One workgroup per (b, q).
//   LDS: query rows [T][E], sim [way][T][T].  Each wave computes dot products (wave-shuffle reduction over E),
//   then 2*way threads run the sequential soft-min DP (few_shot.py:2657-2687), un-stabilised like the reference.
constexpr int MAX_T = 32;
// TT > 0: T is the compile-time constant TT, every loop unrolls and the two DP rows live in registers.  TT == 0: run-time T, the
// rows live in the caller's LDS scratch `rows` (2 x (MAX_T + 2) floats per thread) -- never in scratch memory: the recurrence is
// one dependent chain of T*T cells, and a scratch round trip per cell made this kernel 140 us for an 8x8 problem.
template <int TT>
__device__ __forceinline__ float otam_dp(const float* d /*[T][T] row-major, row stride rs, col stride cs*/, int rs, int cs, int Trt,
                                         float lbda, float* rows) {
    const int T = TT > 0 ? TT : Trt;
    // padded width M = T+2; columns 0 and T+1 are zero padding (few_shot.py:2663)
    float regs[TT > 0 ? 2 * (TT + 2) : 1];
    float* prev = TT > 0 ? regs : rows;
    float* cur = TT > 0 ? regs + (TT + 2) : rows + (MAX_T + 2);
    const float il = 1.0f / lbda;
    prev[0] = 0.f;
#pragma unroll
    for (int m = 1; m <= T + 1; ++m) {                      // first row: running sum (:2668-2671)
        const float dv = (m <= T) ? d[0 * rs + (m - 1) * cs] : 0.f;
        prev[m] = dv + prev[m - 1];
    }
#pragma unroll
    for (int l = 1; l < T; ++l) {
        cur[0] = 0.f;
        {   // first non-zero column (:2675)
            const float dv = d[l * rs + 0 * cs];
            cur[1] = dv - lbda * logf(expf(-prev[0] * il) + expf(-prev[1] * il) + expf(-cur[0] * il));
        }
#pragma unroll
        for (int m = 2; m <= T; ++m) {                      // middle columns (:2678-2679)
            const float dv = d[l * rs + (m - 1) * cs];
            cur[m] = dv - lbda * logf(expf(-prev[m - 1] * il) + expf(-cur[m - 1] * il));
        }
        // last (padding) column (:2683)
        cur[T + 1] = 0.f - lbda * logf(expf(-prev[T] * il) + expf(-prev[T + 1] * il) + expf(-cur[T] * il));
#pragma unroll
        for (int m = 0; m <= T + 1; ++m) prev[m] = cur[m];
    }
    return prev[T + 1];
}

// One workgroup per (query video, class): LDS holds the query's T frames [T][E], their norms and the T x T distance block.
// Phase 1: each wave takes support frames j = wave, wave + 4, ...: the frame's E values sit in registers (float4 per lane per
// 256 columns, E <= 2048), its T dot products accumulate side by side from ds_read_b128s of the query rows and are reduced at
// the end -- no dependent global load inside the loops.  Phase 2: two threads run the two sequential soft-min DPs.
template <int TT>
__global__ __launch_bounds__(256) void cos_otam_kernel(const float* __restrict__ Xq, const float* __restrict__ protos,
                                                       float* __restrict__ logits, float* __restrict__ dists_out, int Q,
                                                       int way, int Trt, int E, float lbda, int single_direct) {
    constexpr int TMAXI = TT > 0 ? TT : MAX_T;
    const int T = TT > 0 ? TT : Trt;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sq = reinterpret_cast<float*>(smem);            // [T][E]
    float* qn = sq + (size_t)T * E;                          // [T] query norms
    float* sd = qn + MAX_T;                                  // [T][T] distances (1 - sim) of this class
    float* dprows = sd + T * T;                              // TT == 0 only: [2][2][MAX_T + 2]
    const int bq = blockIdx.x, c = blockIdx.y, b = bq / Q;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4* xq4 = reinterpret_cast<const float4*>(Xq + (size_t)bq * T * E);
    for (int i = tid; i < T * E / 4; i += 256) reinterpret_cast<float4*>(sq)[i] = xq4[i];
    __syncthreads();
    for (int t = wave; t < T; t += 4) {
        float ss = 0.f;
        for (int e = lane; e < E; e += 64) ss = fmaf(sq[t * E + e], sq[t * E + e], ss);
        ss = wave_sum(ss);
        if (lane == 0) qn[t] = sqrtf(ss);
    }
    __syncthreads();
    const float* pb = protos + ((size_t)b * way + c) * T * E;
    for (int j = wave; j < T; j += 4) {
        const float* pr = pb + (size_t)j * E;
        float4 pv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = 4 * lane + 256 * k;
            pv[k] = e < E ? *reinterpret_cast<const float4*>(pr + e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float yy = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) yy += pv[k].x * pv[k].x + pv[k].y * pv[k].y + pv[k].z * pv[k].z + pv[k].w * pv[k].w;
        const float yn = sqrtf(wave_sum(yy));
        float part[TMAXI];
#pragma unroll
        for (int i = 0; i < TMAXI; ++i) part[i] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = 4 * lane + 256 * k;
            if (256 * k < E) {                                  // wave-uniform
#pragma unroll
                for (int i = 0; i < TMAXI; ++i) {
                    if (i < T && e < E) {
                        const float4 q4 = *reinterpret_cast<const float4*>(sq + (size_t)i * E + e);
                        part[i] += q4.x * pv[k].x + q4.y * pv[k].y + q4.z * pv[k].z + q4.w * pv[k].w;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TMAXI; ++i) {
            if (i < T) {
                const float dot = wave_sum(part[i]);
                if (lane == 0) sd[i * T + j] = 1.0f - dot / (qn[i] * yn + 0.01f);
            }
        }
    }
    __syncthreads();
    if (dists_out)
        for (int i = tid; i < T * T; i += 256) dists_out[((size_t)bq * way + c) * T * T + i] = sd[i];
    __shared__ float res[2];
    if (tid < 2) {
        float v = 0.f;
        float* rows = dprows + (TT > 0 ? 0 : tid * 2 * (MAX_T + 2));
        if (tid == 0) v = otam_dp<TT>(sd, T, 1, T, lbda, rows);
        else if (!single_direct) v = otam_dp<TT>(sd, 1, T, T, lbda, rows);        // transposed distances (:2982)
        res[tid] = v;
    }
    __syncthreads();
    if (tid == 0) logits[(size_t)bq * way + c] = -(res[0] + res[1]);
}

}  // namespace

extern "C" int cfsar_class_text_logits(const float* feats, const float* text, const float* scale, float* out,
                                       int n_videos, int T, int E, int n_cls, cfsar_stream_t stream) {
    CFSAR_REQUIRE(feats && text && scale && out, "cfsar_class_text_logits: null pointer");
    CFSAR_REQUIRE(n_videos > 0 && T > 0 && E > 0 && n_cls > 0 && E * 4 <= 60000, "cfsar_class_text_logits: bad shape");
    hipLaunchKernelGGL(class_text_logits_kernel, dim3(n_videos, (n_cls + 15) / 16), dim3(256), E * sizeof(float),
                       static_cast<hipStream_t>(stream), feats, text, scale, out, T, E, n_cls);
    return cfsar_check_launch("cfsar_class_text_logits");
}

extern "C" int cfsar_build_sequences(const float* feats, const float* text_test, const float* support_labels,
                                     const float* real_support_labels, float* X, int B, int S, int Q, int T, int E,
                                     int way, int n_test, int merge_before, cfsar_stream_t stream) {
    CFSAR_REQUIRE(feats && text_test && support_labels && real_support_labels && X, "cfsar_build_sequences: null pointer");
    CFSAR_REQUIRE(B > 0 && S > 0 && Q > 0 && T > 0 && E > 0 && way > 0 && n_test > 0, "cfsar_build_sequences: bad shape");
    CFSAR_REQUIRE(S <= MAX_S, "cfsar_build_sequences: S=%d > %d", S, MAX_S);
    CFSAR_REQUIRE(S % way == 0, "cfsar_build_sequences: S=%d is not a multiple of way=%d", S, way);
    const int Sp = merge_before ? way : S;
    const long long rows = (long long)B * Q * T + (long long)B * Sp * (T + 1);
    hipLaunchKernelGGL(build_sequences_kernel, dim3((unsigned)rows), dim3(128), 0, static_cast<hipStream_t>(stream),
                       feats, text_test, support_labels, real_support_labels, X, B, S, Q, T, E, way, n_test,
                       merge_before);
    return cfsar_check_launch("cfsar_build_sequences");
}

extern "C" int cfsar_seq_attention(const float* qkv, float* out, int n_a, int len_a, int n_b, int len_b, int heads,
                                   int head_dim, float scale, int causal, cfsar_stream_t stream) {
    CFSAR_REQUIRE(qkv && out, "cfsar_seq_attention: null pointer");
    CFSAR_REQUIRE(n_a >= 0 && n_b >= 0 && n_a + n_b > 0 && heads > 0 && head_dim > 0, "cfsar_seq_attention: bad shape");
    const int L = len_a > len_b ? len_a : len_b;
    CFSAR_REQUIRE(L <= 128 && head_dim <= 128, "cfsar_seq_attention: len=%d (max 128) head_dim=%d (max 128)", L, head_dim);
    const int lds = (3 * L * (head_dim + 1) + L * (L + 1)) * (int)sizeof(float);
    CFSAR_REQUIRE(lds <= 160 * 1024, "cfsar_seq_attention: len x head_dim too large for LDS");
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&seq_attention_kernel), lds, "cfsar_seq_attention")) return rc;
    hipLaunchKernelGGL(seq_attention_kernel, dim3(n_a + n_b, heads), dim3(128), lds, static_cast<hipStream_t>(stream), qkv,
                       out, n_a, len_a, n_b, len_b, heads, head_dim, scale, causal);
    return cfsar_check_launch("cfsar_seq_attention");
}

extern "C" int cfsar_embed_tokens(const int32_t* tokens, const float* table, const float* pos, float* out, int n_seq,
                                  int L, int W, int vocab, cfsar_stream_t stream) {
    CFSAR_REQUIRE(tokens && table && pos && out, "cfsar_embed_tokens: null pointer");
    CFSAR_REQUIRE(n_seq > 0 && L > 0 && W > 0 && vocab > 0, "cfsar_embed_tokens: bad shape");
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)(n_seq * L)), dim3(128), 0, static_cast<hipStream_t>(stream),
                       tokens, table, pos, out, L, W, vocab);
    return cfsar_check_launch("cfsar_embed_tokens");
}

extern "C" int cfsar_gather_rows(const float* x, const int32_t* idx, float* out, int n, int D, int rows_in,
                                 cfsar_stream_t stream) {
    CFSAR_REQUIRE(x && idx && out && n > 0 && D > 0 && rows_in > 0, "cfsar_gather_rows: bad arguments");
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n), dim3(128), 0, static_cast<hipStream_t>(stream), x, idx, out, D,
                       rows_in);
    return cfsar_check_launch("cfsar_gather_rows");
}

extern "C" int cfsar_text_match_probs(const float* feats, const float* text_test, const float* support_labels,
                                      const float* real_support_labels, const float* scale, float* probs, int B, int S,
                                      int Q, int T, int E, int way, int n_test, cfsar_stream_t stream) {
    CFSAR_REQUIRE(feats && text_test && support_labels && real_support_labels && scale && probs,
                  "cfsar_text_match_probs: null pointer");
    CFSAR_REQUIRE(B > 0 && S > 0 && S <= MAX_S && Q > 0 && T > 0 && E > 0 && way > 0 && way <= 64 && n_test > 0,
                  "cfsar_text_match_probs: bad shape");
    hipLaunchKernelGGL(text_match_kernel, dim3((unsigned)(B * Q)), dim3(256), 2 * E * sizeof(float),
                       static_cast<hipStream_t>(stream), feats, text_test, support_labels, real_support_labels, scale, probs, S,
                       Q, T, E, way, n_test);
    return cfsar_check_launch("cfsar_text_match_probs");
}

namespace {
// A17: per-episode top-1 accuracy.  One thread per episode: the first maximum of each query's `way` logits (torch.argmax / topk order) against
// the query's label.
__global__ __launch_bounds__(64) void episode_top1_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                                          float* __restrict__ acc, int episodes, int Q, int way) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= episodes) return;
    int hit = 0;
    for (int q = 0; q < Q; ++q) {
        const float* l = logits + ((size_t)e * Q + q) * way;
        int best = 0;
        float bv = l[0];
        for (int c = 1; c < way; ++c)
            if (l[c] > bv) { bv = l[c]; best = c; }
        hit += (best == (int)labels[(size_t)e * Q + q]);
    }
    acc[e] = (float)hit / (float)Q;
}
}  // namespace

extern "C" int cfsar_episode_top1(const float* logits, const float* target_labels, float* acc, int episodes, int Q, int way,
                                  cfsar_stream_t stream) {
    CFSAR_REQUIRE(logits && target_labels && acc && episodes > 0 && Q > 0 && way > 0, "cfsar_episode_top1: bad arguments");
    hipLaunchKernelGGL(episode_top1_kernel, dim3((unsigned)((episodes + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream), logits,
                       target_labels, acc, episodes, Q, way);
    return cfsar_check_launch("cfsar_episode_top1");
}

extern "C" int cfsar_combine_logits(const float* text_probs, const float* visual_logits, float* out, int n_queries, int way,
                                    float text_coff, cfsar_stream_t stream) {
    CFSAR_REQUIRE(text_probs && visual_logits && out && n_queries > 0 && way > 0, "cfsar_combine_logits: bad arguments");
    hipLaunchKernelGGL(combine_kernel, dim3((unsigned)n_queries), dim3(64), 0, static_cast<hipStream_t>(stream), text_probs,
                       visual_logits, out, way, text_coff);
    return cfsar_check_launch("cfsar_combine_logits");
}

extern "C" int cfsar_prototypes(const float* Xs, const float* support_labels, float* protos, int B, int S, int Sp, int T,
                                int E, int way, int merge_before, cfsar_stream_t stream) {
    CFSAR_REQUIRE(Xs && support_labels && protos, "cfsar_prototypes: null pointer");
    CFSAR_REQUIRE(B > 0 && S > 0 && S <= MAX_S && T > 0 && E > 0 && way > 0, "cfsar_prototypes: bad shape");
    CFSAR_REQUIRE(Sp == (merge_before ? way : S), "cfsar_prototypes: Sp=%d inconsistent with merge_before", Sp);
    hipLaunchKernelGGL(prototypes_kernel, dim3((unsigned)(B * way * T)), dim3(128), 0, static_cast<hipStream_t>(stream), Xs,
                       support_labels, protos, S, Sp, T, E, way, merge_before);
    return cfsar_check_launch("cfsar_prototypes");
}

extern "C" int cfsar_cos_otam_logits(const float* Xq, const float* protos, float* logits, float* dists_out, int B, int Q,
                                     int way, int T, int E, float lambda, int single_direct, cfsar_stream_t stream) {
    CFSAR_REQUIRE(Xq && protos && logits, "cfsar_cos_otam_logits: null pointer");
    CFSAR_REQUIRE(B > 0 && Q > 0 && way > 0 && way <= 65535 && T > 0 && T <= MAX_T && E > 0 && E % 4 == 0 && E <= 2048,
                  "cfsar_cos_otam_logits: bad shape (T <= 32, E %% 4 == 0, E <= 2048)");
    const bool fixed_t = T == 8 || T == 16;                  // DP rows in registers; otherwise 2 rows per DP thread in LDS
    const int lds = (T * E + MAX_T + T * T + (fixed_t ? 0 : 2 * 2 * (MAX_T + 2))) * (int)sizeof(float);
    CFSAR_REQUIRE(lds <= 150 * 1024, "cfsar_cos_otam_logits: T*E too large for LDS");
    auto launch = [&](auto kern) -> int {
        if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(kern), lds, "cfsar_cos_otam_logits")) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)(B * Q), (unsigned)way), dim3(256), lds, static_cast<hipStream_t>(stream), Xq, protos, logits,
                           dists_out, Q, way, T, E, lambda, single_direct);
        return cfsar_check_launch("cfsar_cos_otam_logits");
    };
    if (T == 8) return launch(&cos_otam_kernel<8>);          // DATA.NUM_INPUT_FRAMES of the shipped configs
    if (T == 16) return launch(&cos_otam_kernel<16>);
    return launch(&cos_otam_kernel<0>);
}
