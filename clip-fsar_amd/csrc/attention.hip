// A5: scaled-dot-product attention inside nn.MultiheadAttention of the CLIP ViT (few_shot.py:623,633-635):
// per (frame, head): softmax(q k^T / sqrt(64)) v, no mask, no dropout.  Sequence = 197 (B/16) or 257 (L/14) tokens,
// head_dim = 64, so K and V of one (frame, head) fit in LDS and the softmax is single-pass.
#include <cstdlib>

#include "common.h"

namespace {

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
typedef unsigned att_u32x4 __attribute__((ext_vector_type(4)));

// O^T tile -> global, 16 bytes per lane.  A lane (q = lane&15, g = lane>>4) holds d = 16dt + 4g + r of its query row; lanes g
// and g^1 hold the two halves of each 8-wide d chunk.  After one exchange with lane^16 (4 dwords each way) the even-g lane owns
// the chunks of dt 0,1 and the odd-g lane those of dt 2,3: two dwordx4 stores per lane instead of four dwordx2 (the
// attention epilogue is store-ISSUE bound: 59 of 273 us at 640 frames were the 8-byte stores).
__device__ __forceinline__ void pack_o_tile(const f32x4 (&o)[4], float inv, int g, uint4 (&val)[2]) {
    unsigned pk[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        bf16x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (__bf16)(o[dt][r] * inv);
        const uint2 u = __builtin_bit_cast(uint2, v);
        pk[dt][0] = u.x;
        pk[dt][1] = u.y;
    }
    const bool odd = g & 1;
    // send the two dt chunks the partner will store, keep the two this lane stores
    unsigned recv[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int w = 0; w < 2; ++w) recv[i][w] = __shfl_xor(odd ? pk[i][w] : pk[2 + i][w], 16, 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int dt = odd ? 2 + i : i;
        // chunk of 8 d values starting at 16dt + 8(g>>1): low half from the even-g lane, high half from the odd-g lane
        val[i] = odd ? make_uint4(recv[i][0], recv[i][1], pk[dt][0], pk[dt][1]) : make_uint4(pk[dt][0], pk[dt][1], recv[i][0], recv[i][1]);
    }
}
__device__ __forceinline__ void store_o_packed(const uint4 (&val)[2], bool valid, __bf16* orow, int g) {
    if (valid) {
        const bool odd = g & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(orow + (odd ? 2 + i : i) * 16 + (g >> 1) * 8) = val[i];
    }
}
__device__ __forceinline__ void store_o_tile(const f32x4 (&o)[4], float inv, bool valid, __bf16* orow, int g) {
    uint4 val[2];
    pack_o_tile(o, inv, g, val);
    store_o_packed(val, valid, orow, g);
}

// ------------------------------------------------------------------------------------------------------------
// bf16 MFMA kernel.  One 256-thread workgroup per (head, frame).
//   LDS:  K tile  [NP keys][64] bf16, 128-byte rows, 16-byte chunks XOR-swizzled with (row>>1)&7
//         V^T tile [64 d][NP keys] bf16, row stride == 16 (mod 256) bytes -> conflict-free ds_read_b64
//   Per wave: 16 query rows at a time.  S^T = K . Q^T with v_mfma_f32_16x16x32_bf16 (K rows feed MFMA "A", Q^T feeds
//   "B"), so lane (q = lane&15, g = lane>>4) holds, for every 16-key tile j, the scores of keys 16j+4g+{0..3} for
//   ITS query: the softmax reduction over keys is in-lane + two shuffles (xor 16, 32), and the exponentiated
//   scores are already in the "B" fragment layout of the PV MFMA (O^T = V^T . P^T) because the MFMA k index is
//   only a summation index: k-slot (g, j<4) <-> key 32kb+4g+j, (g, j>=4) <-> key 32kb+16+4g+(j-4); V^T fragments are
//   read from LDS with the same key permutation.  No P round trip through LDS, no cross-lane data movement.
// ------------------------------------------------------------------------------------------------------------
// NKB = number of 32-key blocks (7 -> up to 224 keys, 9 -> up to 288 keys); NTV = number of 16-key tiles that hold at
// least one real key when known at compile time (13 for 197 tokens, 17 for 257), 0 = generic (all tiles, all masked).
// ATT_THREADS = 512: 8 waves share one staged (frame, head); 2 workgroups per CU (LDS 61 KB each) = 4 waves/SIMD.
// COMPACT (needs NTV > 0): K holds only the NTV*16 rows that are read and V^T rows are 16*NTV + 8 keys long -> 54 336 B for
// 197 tokens, THREE workgroups per CU; used with 256-thread workgroups (4 waves, 3-4 query tiles each).
template <int NTV>
constexpr int att_vt_stride(int NP, bool compact) { return compact ? (NTV * 16 + 8) * 2 : ((NP * 2 + 255) / 256) * 256 + 16; }
template <int NKB, int NTV, int ATT_THREADS = 512, bool COMPACT = false>
__global__ __launch_bounds__(ATT_THREADS, ATT_THREADS == 512 ? 4 : 3) void vit_attn_bf16_kernel(const __bf16* __restrict__ qkv, __bf16* __restrict__ out,
                                                            int ntok, int D, float scale_log2e) {
    constexpr int NP = NKB * 32;
    constexpr int NT = NKB * 2;                                    // 16-key tiles
    constexpr int VT_STRIDE = att_vt_stride<NTV>(NP, COMPACT);     // bytes
    constexpr int KROWS = COMPACT ? NTV * 16 : NP;                 // K rows kept in LDS
    constexpr int VKEYS = COMPACT ? VT_STRIDE / 2 : NP;            // keys per V^T row that are written (pairs: VKEYS / 2)
    static_assert(!COMPACT || NTV > 0, "the compact LDS image needs the number of valid key tiles at compile time");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sVt = smem + KROWS * 128;

    const int h = blockIdx.x, f = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t ld = (size_t)3 * D;
    const __bf16* base = qkv + (size_t)f * ntok * ld + h * 64;

    // ---- stage K (swizzled rows, NP*8 16-byte chunks) and V^T: ALL global loads of a thread are issued before the first LDS
    // write (cycle stamps: the load -> write -> load -> write form spent 11.5 K of a workgroup's 28.8 K cycles here, three to
    // four dependent memory round trips)
    constexpr int KIT = (KROWS * 8 + ATT_THREADS - 1) / ATT_THREADS;
    constexpr int VIT = ((VKEYS / 2) * 8 + ATT_THREADS - 1) / ATT_THREADS;
    att_u32x4 kreg[KIT], vreg0[VIT], vreg1[VIT];     // native vectors: plain SSA values (a HIP uint4 copy is a memcpy -> scratch)
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
        const int idx = tid + it * ATT_THREADS;
        const int r = idx >> 3, c = (idx & 7) ^ swz(r);
        kreg[it] = att_u32x4{0, 0, 0, 0};
        if (idx < KROWS * 8 && r < ntok) kreg[it] = *reinterpret_cast<const att_u32x4*>(base + (size_t)r * ld + D + c * 8);
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
        const int idx = tid + it * ATT_THREADS;
        const int kp = idx % (VKEYS / 2), dc = idx / (VKEYS / 2);
        vreg0[it] = vreg1[it] = att_u32x4{0, 0, 0, 0};
        if (idx < (VKEYS / 2) * 8) {
            if (2 * kp < ntok) vreg0[it] = *reinterpret_cast<const att_u32x4*>(base + (size_t)(2 * kp) * ld + 2 * D + dc * 8);
            if (2 * kp + 1 < ntok) vreg1[it] = *reinterpret_cast<const att_u32x4*>(base + (size_t)(2 * kp + 1) * ld + 2 * D + dc * 8);
        }
    }
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
        const int idx = tid + it * ATT_THREADS;
        if (idx < KROWS * 8) *reinterpret_cast<att_u32x4*>(sK + (idx >> 3) * 128 + (idx & 7) * 16) = kreg[it];
    }
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
        const int idx = tid + it * ATT_THREADS;
        if (idx < (VKEYS / 2) * 8) {
            const int kp = idx % (VKEYS / 2), dc = idx / (VKEYS / 2);
            const unsigned a[4] = {vreg0[it][0], vreg0[it][1], vreg0[it][2], vreg0[it][3]};
            const unsigned b[4] = {vreg1[it][0], vreg1[it][1], vreg1[it][2], vreg1[it][3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<unsigned*>(sVt + (dc * 8 + 2 * j) * VT_STRIDE + kp * 4) = (a[j] & 0xFFFFu) | (b[j] << 16);          // d = dc*8 + 2j
                *reinterpret_cast<unsigned*>(sVt + (dc * 8 + 2 * j + 1) * VT_STRIDE + kp * 4) = (a[j] >> 16) | (b[j] & 0xFFFF0000u);  // d + 1
            }
        }
    }
    if constexpr (COMPACT) {                 // the last V^T row is over-read by 16 bytes (keys 216-223 of d = 63): keep them finite
        if (tid < 4) *reinterpret_cast<att_u32x4*>(sVt + 64 * VT_STRIDE + tid * 16) = att_u32x4{0, 0, 0, 0};
    }
    __syncthreads();

    const int q16 = lane & 15, g = lane >> 4;
    const int nqt = (ntok + 15) >> 4;
    constexpr int nt_valid = NTV > 0 ? NTV : NKB * 2;   // 16-key tiles that are computed
    for (int qt = wave; qt < nqt; qt += ATT_THREADS / 64) {
        int qrow = qt * 16 + q16;
        const bool qvalid = qrow < ntok;
        if (!qvalid) qrow = ntok - 1;
        // Q fragment ("B" operand): Q[qrow][32ks + 8g .. +8]
        bf16x8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[ks] = *reinterpret_cast<const bf16x8*>(base + (size_t)qrow * ld + ks * 32 + g * 8);

        // S^T tiles (tiles that hold only padded keys are skipped; their probabilities are 0)
        f32x4 s[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (j < nt_valid) {
                const int kr = j * 16 + q16;   // key row this lane reads for the "A" operand
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + kr * 128 + (((ks * 4 + g) ^ swz(kr)) << 4));
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], acc, 0, 0, 0);
                }
            }
            s[j] = acc;
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // bound the K-fragment live ranges (no spills)
        }
        // mask the padded keys of the last (partial) tile, row max over keys (in-lane, then across the 4 lane groups)
        float mx = -1e30f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (NTV == 0 || j == nt_valid - 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (j * 16 + g * 4 + r >= ntok) s[j][r] = -1e30f;
            }
            if (j < nt_valid) mx = fmaxf(fmaxf(mx, fmaxf(s[j][0], s[j][1])), fmaxf(s[j][2], s[j][3]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mxs = mx * scale_log2e;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j < nt_valid) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(s[j][r], scale_log2e, -mxs));   // raw v_exp_f32
                    s[j][r] = pv;
                    sum += pv;
                }
            }
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(sum);
        // O^T = V^T . P^T
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            bf16x8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (__bf16)s[2 * kb][r];
                pf[4 + r] = (__bf16)s[2 * kb + 1][r];
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const char* vr = sVt + (dt * 16 + q16) * VT_STRIDE + (kb * 32 + g * 4) * 2;
                const uint2 lo = *reinterpret_cast<const uint2*>(vr);
                const uint2 hi = *reinterpret_cast<const uint2*>(vr + 32);
                const uint4 packed = make_uint4(lo.x, lo.y, hi.x, hi.y);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, packed), pf, o[dt], 0, 0, 0);
            }
            if (kb & 1) __builtin_amdgcn_sched_barrier(0);
        }
        // O^T[d][q]: lane owns query q16, d = 16dt + 4g + r
        store_o_tile(o, inv, qvalid, out + ((size_t)f * ntok + qrow) * D + h * 64, g);
    }
}

// ------------------------------------------------------------------------------------------------------------
// fp32 VALU kernel (validation mode).  One workgroup per (head, frame); K and V rows in LDS (fp32); one thread per
// query row with an online softmax (running max / sum, fp32) -- all lanes read the same K/V row (LDS broadcast).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vit_attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           int ntok, int D, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sK = reinterpret_cast<float*>(smem);
    float* sV = sK + (size_t)ntok * 64;
    const int h = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const size_t ld = (size_t)3 * D;
    const float* base = qkv + (size_t)f * ntok * ld + h * 64;
    for (int idx = tid; idx < ntok * 16; idx += 256) {
        const int r = idx >> 4, c = idx & 15;
        *reinterpret_cast<float4*>(sK + r * 64 + c * 4) = *reinterpret_cast<const float4*>(base + (size_t)r * ld + D + c * 4);
        *reinterpret_cast<float4*>(sV + r * 64 + c * 4) =
            *reinterpret_cast<const float4*>(base + (size_t)r * ld + 2 * D + c * 4);
    }
    __syncthreads();
    for (int qrow = tid; qrow < ntok; qrow += 256) {
        float q[64], o[64];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)qrow * ld + c * 4);
            q[4 * c] = v.x * scale; q[4 * c + 1] = v.y * scale; q[4 * c + 2] = v.z * scale; q[4 * c + 3] = v.w * scale;
        }
#pragma unroll
        for (int d = 0; d < 64; ++d) o[d] = 0.f;
        float m = -1e30f, l = 0.f;
        for (int key = 0; key < ntok; ++key) {
            const float* kr = sK + key * 64;
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < 64; ++d) sc = fmaf(q[d], kr[d], sc);
            const float mn = fmaxf(m, sc);
            const float a = expf(m - mn), pv = expf(sc - mn);
            l = l * a + pv;
            const float* vr = sV + key * 64;
#pragma unroll
            for (int d = 0; d < 64; ++d) o[d] = fmaf(o[d], a, pv * vr[d]);
            m = mn;
        }
        const float inv = 1.0f / l;
        float* orow = out + ((size_t)f * ntok + qrow) * D + h * 64;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            *reinterpret_cast<float4*>(orow + 4 * c) =
                make_float4(o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv);
    }
}

template <int NKB, int NTV, int NTHR = 512, bool COMPACT = false>
int launch_bf16(const void* qkv, void* out, int F, int ntok, int D, int heads, hipStream_t s) {
    constexpr int NP = NKB * 32;
    constexpr int VT_STRIDE = att_vt_stride<NTV>(NP, COMPACT);
    constexpr int LDS = (COMPACT ? NTV * 16 : NP) * 128 + 64 * VT_STRIDE + (COMPACT ? 64 : 0);   // + tail pad: the last V^T row is over-read
    if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&vit_attn_bf16_kernel<NKB, NTV, NTHR, COMPACT>), LDS, "cfsar_vit_attention")) return rc;
    const float scale_log2e = 0.125f * 1.4426950408889634f;
    hipLaunchKernelGGL((vit_attn_bf16_kernel<NKB, NTV, NTHR, COMPACT>), dim3(heads, F), dim3(NTHR), LDS, s,
                       static_cast<const __bf16*>(qkv), static_cast<__bf16*>(out), ntok, D, scale_log2e);
    return cfsar_check_launch("cfsar_vit_attention(bf16)");
}

}  // namespace

extern "C" int cfsar_vit_attention(const void* qkv, void* out, int dtype, int F, int ntok, int D, int heads,
                                   cfsar_stream_t stream) {
    CFSAR_REQUIRE(qkv && out, "cfsar_vit_attention: null pointer");
    CFSAR_REQUIRE(F > 0 && ntok > 0 && heads > 0 && D == heads * 64, "cfsar_vit_attention: need D == heads*64 (D=%d heads=%d)",
                  D, heads);
    CFSAR_REQUIRE(F <= 65535, "cfsar_vit_attention: too many frames per call (%d)", F);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == CFSAR_BF16) {
        CFSAR_REQUIRE(ntok <= 288, "cfsar_vit_attention: ntok=%d > 288", ntok);
        // (A compact-LDS three-workgroups-per-CU form and a pipelined multi-head form were measured slower / equal in round 1 and
        // removed; numbers and per-phase cycle stamps: profiles/r01_attention_ablation.md.)
        if (ntok == 197) return launch_bf16<7, 13>(qkv, out, F, ntok, D, heads, s);
        if (ntok == 257) return launch_bf16<9, 17>(qkv, out, F, ntok, D, heads, s);     // ViT-L/14 @224
        if (ntok <= 224) return launch_bf16<7, 0>(qkv, out, F, ntok, D, heads, s);
        return launch_bf16<9, 0>(qkv, out, F, ntok, D, heads, s);
    }
    if (dtype == CFSAR_F32) {
        const int lds = ntok * 64 * 4 * 2;
        CFSAR_REQUIRE(lds <= 160 * 1024, "cfsar_vit_attention: ntok=%d too large for the fp32 kernel", ntok);
        if (int rc = cfsar_ensure_lds(reinterpret_cast<const void*>(&vit_attn_f32_kernel), lds, "cfsar_vit_attention")) return rc;
        hipLaunchKernelGGL(vit_attn_f32_kernel, dim3(heads, F), dim3(256), lds, s, static_cast<const float*>(qkv),
                           static_cast<float*>(out), ntok, D, 0.125f);
        return cfsar_check_launch("cfsar_vit_attention(f32)");
    }
    return cfsar_fail("cfsar_vit_attention: bad dtype %d", dtype);
}
