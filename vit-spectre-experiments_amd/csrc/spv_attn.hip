// spv_attn.hip -- softmax attention core of the baseline ViT (nn.MultiheadAttention inside the stock
// nn.TransformerEncoderLayer, reference spectre_vit/models/vit/vit.py:30-38): ctx = softmax(Q K^T / sqrt(hd)) V per
// (sequence, head), forward and backward, with the attention-probability dropout of the stock layer.
//
// Layout: qkv [seqs, len, 3E] (q | k | v, head h at columns h*hd of each third), ctx / dctx [seqs, len, E],
// probs / dscores [seqs, heads, len, len] (softmax output WITHOUT the dropout mask; the mask is regenerated from the
// counter hash wherever it is needed).  Which tensor axis is `len` is the caller's business: the reference feeds
// (B, N, E) with batch_first = False, so the sequence axis is the image batch.
//
// One wave per query row (forward, backward-1) or key row (backward-2): lanes over keys for the score row and the
// softmax reductions, lanes over the head dimension for the P.V / dS.K products with the probabilities broadcast from
// a wave-private LDS row.  First correct version of SURVEY row 8a-8; not yet an MFMA flash kernel.
#include "spv_common.h"

namespace {

constexpr int AW = 4;          // waves per workgroup
constexpr int MAXC = 16;       // len <= 64 * MAXC
constexpr int MAXHD = 128;

__device__ __forceinline__ float ldq(const void* p, size_t off, int bf) {
    return bf ? bf2f(static_cast<const bf16_t*>(p)[off]) : static_cast<const float*>(p)[off];
}
__device__ __forceinline__ void stq(void* p, size_t off, int bf, float v) {
    if (bf) static_cast<bf16_t*>(p)[off] = f2bf(v);
    else static_cast<float*>(p)[off] = v;
}
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// row id for the dropout hash: unique per (seq, head, query)
__device__ __forceinline__ uint64_t attn_row(int sh, int len, int i) { return (uint64_t)sh * len + i; }

__global__ __launch_bounds__(256) void attn_fwd_kernel(const void* __restrict__ qkv, void* __restrict__ ctx, void* __restrict__ probs,
                                                       int len, int heads, int hd, int bf, float p_drop, uint64_t seed) {
    extern __shared__ __attribute__((aligned(16))) float lds_f32[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* prow = lds_f32 + wave * (len + MAXHD);  // probabilities (masked) of this wave's query row
    float* qrow = prow + len;                        // its query vector
    const int sh = blockIdx.x;  // (sequence, head)
    const int s = sh / heads, h = sh % heads;
    const int E = heads * hd;
    const int i = blockIdx.y * AW + wave;
    if (i >= len) return;  // whole wave leaves: no workgroup barrier below
    const size_t rowq = ((size_t)s * len + i) * 3 * E + h * hd;
    const float scale = rsqrtf((float)hd);
    for (int d = lane; d < hd; d += 64) qrow[d] = ldq(qkv, rowq + d, bf) * scale;
    lds_sync();
    float sc[MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int k = c * 64 + lane;
        sc[c] = -INFINITY;
        if (c * 64 < len && k < len) {
            const size_t rk = ((size_t)s * len + k) * 3 * E + E + h * hd;
            float a = 0.0f;
            for (int d = 0; d < hd; ++d) a = fmaf(qrow[d], ldq(qkv, rk + d, bf), a);
            sc[c] = a;
            mx = fmaxf(mx, a);
        }
    }
    mx = wave_max(mx);
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c * 64 < len) {
            sc[c] = (c * 64 + lane < len) ? __expf(sc[c] - mx) : 0.0f;
            sum += sc[c];
        }
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum, inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const unsigned rkey = p_drop > 0.0f ? dropout_row_key(seed, attn_row(sh, len, i)) : 0u;
    const size_t prow_g = ((size_t)sh * len + i) * len;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int k = c * 64 + lane;
        if (c * 64 < len && k < len) {
            const float p = sc[c] * inv;
            stq(probs, prow_g + k, bf, p);
            prow[k] = p_drop > 0.0f ? p * dropout_scale(rkey, (unsigned)k, p_drop, inv_keep) : p;
        }
    }
    lds_sync();
    // ctx[i][d] = sum_k p[k] V[k][d]: lanes over d, probabilities broadcast from LDS
    for (int d = lane; d < hd; d += 64) {
        float a = 0.0f;
        const size_t vb = (size_t)s * len * 3 * E + 2 * E + h * hd + d;
        for (int k = 0; k < len; ++k) a = fmaf(prow[k], ldq(qkv, vb + (size_t)k * 3 * E, bf), a);
        stq(ctx, ((size_t)s * len + i) * E + h * hd + d, bf, a);
    }
}

// backward 1 (wave per query row): dS[i][k] = P (dP - sum_k P dP) / sqrt(hd) with dP = mask * (dctx . V^T); dQ[i] = dS[i] . K
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(const void* __restrict__ dctx, const void* __restrict__ qkv,
                                                         const void* __restrict__ probs, void* __restrict__ ds,
                                                         void* __restrict__ dqkv, int len, int heads, int hd, int bf, float p_drop,
                                                         uint64_t seed) {
    extern __shared__ __attribute__((aligned(16))) float lds_f32[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* srow = lds_f32 + wave * (len + MAXHD);
    float* grow = srow + len;  // dctx row
    const int sh = blockIdx.x;
    const int s = sh / heads, h = sh % heads;
    const int E = heads * hd;
    const int i = blockIdx.y * AW + wave;
    if (i >= len) return;
    const float scale = rsqrtf((float)hd);
    const size_t gi = ((size_t)s * len + i) * E + h * hd;
    for (int d = lane; d < hd; d += 64) grow[d] = ldq(dctx, gi + d, bf);
    lds_sync();
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const unsigned rkey = p_drop > 0.0f ? dropout_row_key(seed, attn_row(sh, len, i)) : 0u;
    const size_t prow_g = ((size_t)sh * len + i) * len;
    float dp[MAXC], pr[MAXC];
    float dot = 0.0f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int k = c * 64 + lane;
        dp[c] = 0.0f;
        pr[c] = 0.0f;
        if (c * 64 < len && k < len) {
            const size_t rv = ((size_t)s * len + k) * 3 * E + 2 * E + h * hd;
            float a = 0.0f;
            for (int d = 0; d < hd; ++d) a = fmaf(grow[d], ldq(qkv, rv + d, bf), a);
            if (p_drop > 0.0f) a *= dropout_scale(rkey, (unsigned)k, p_drop, inv_keep);
            dp[c] = a;
            pr[c] = ldq(probs, prow_g + k, bf);
            dot += pr[c] * a;
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int k = c * 64 + lane;
        if (c * 64 < len && k < len) {
            const float v = pr[c] * (dp[c] - dot) * scale;
            stq(ds, prow_g + k, bf, v);
            srow[k] = v;
        }
    }
    lds_sync();
    for (int d = lane; d < hd; d += 64) {
        float a = 0.0f;
        const size_t kb = (size_t)s * len * 3 * E + E + h * hd + d;
        for (int k = 0; k < len; ++k) a = fmaf(srow[k], ldq(qkv, kb + (size_t)k * 3 * E, bf), a);
        stq(dqkv, ((size_t)s * len + i) * 3 * E + h * hd + d, bf, a);
    }
}

// backward 2 (wave per key row): dK[k] = sum_i dS[i][k] Q[i], dV[k] = sum_i (mask P)[i][k] dctx[i]
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(const void* __restrict__ dctx, const void* __restrict__ qkv,
                                                          const void* __restrict__ probs, const void* __restrict__ ds,
                                                          void* __restrict__ dqkv, int len, int heads, int hd, int bf, float p_drop,
                                                          uint64_t seed) {
    extern __shared__ __attribute__((aligned(16))) float lds_f32[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* scol = lds_f32 + wave * 2 * len;  // dS[:, k]
    float* pcol = scol + len;                // (mask P)[:, k]
    const int sh = blockIdx.x;
    const int s = sh / heads, h = sh % heads;
    const int E = heads * hd;
    const int k = blockIdx.y * AW + wave;
    if (k >= len) return;
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int i = lane; i < len; i += 64) {
        const size_t e = ((size_t)sh * len + i) * len + k;
        float p = ldq(probs, e, bf);
        if (p_drop > 0.0f) p *= dropout_scale(dropout_row_key(seed, attn_row(sh, len, i)), (unsigned)k, p_drop, inv_keep);
        scol[i] = ldq(ds, e, bf);
        pcol[i] = p;
    }
    lds_sync();
    for (int d = lane; d < hd; d += 64) {
        float ak = 0.0f, av = 0.0f;
        const size_t qb = (size_t)s * len * 3 * E + h * hd + d;
        const size_t gb = (size_t)s * len * E + h * hd + d;
        for (int i = 0; i < len; ++i) {
            ak = fmaf(scol[i], ldq(qkv, qb + (size_t)i * 3 * E, bf), ak);
            av = fmaf(pcol[i], ldq(dctx, gb + (size_t)i * E, bf), av);
        }
        const size_t o = ((size_t)s * len + k) * 3 * E + h * hd + d;
        stq(dqkv, o + E, bf, ak);
        stq(dqkv, o + 2 * E, bf, av);
    }
}

int check(const char* name, int seqs, int len, int heads, int hd, int dtype) {
    SPV_CHECK(seqs > 0 && len > 0 && heads > 0 && hd > 0, "%s: empty", name);
    SPV_CHECK(len <= 64 * MAXC, "%s: len=%d > %d", name, len, 64 * MAXC);
    SPV_CHECK(hd <= MAXHD, "%s: head_dim=%d > %d", name, hd, MAXHD);
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "%s: bad dtype %d", name, dtype);
    return 0;
}

}  // namespace

extern "C" int spv_attention_fwd(const void* qkv, void* ctx, void* probs, int seqs, int len, int heads, int head_dim, int dtype,
                                 float p_drop, uint64_t seed, void* stream) {
    if (check("spv_attention_fwd", seqs, len, heads, head_dim, dtype)) return 1;
    dim3 grid(seqs * heads, cdiv(len, AW));
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(64 * AW), (size_t)AW * (len + MAXHD) * sizeof(float), static_cast<hipStream_t>(stream),
                       qkv, ctx, probs, len, heads, head_dim, dtype == SPV_BF16, p_drop, seed);
    SPV_LAUNCH_CHECK("spv_attention_fwd");
    return 0;
}

extern "C" int spv_attention_bwd(const void* dctx, const void* qkv, const void* probs, void* dscores, void* dqkv, int seqs, int len,
                                 int heads, int head_dim, int dtype, float p_drop, uint64_t seed, void* stream) {
    if (check("spv_attention_bwd", seqs, len, heads, head_dim, dtype)) return 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(seqs * heads, cdiv(len, AW));
    hipLaunchKernelGGL(attn_bwd_q_kernel, grid, dim3(64 * AW), (size_t)AW * (len + MAXHD) * sizeof(float), st, dctx, qkv, probs, dscores,
                       dqkv, len, heads, head_dim, dtype == SPV_BF16, p_drop, seed);
    SPV_LAUNCH_CHECK("spv_attention_bwd(q)");
    hipLaunchKernelGGL(attn_bwd_kv_kernel, grid, dim3(64 * AW), (size_t)AW * 2 * len * sizeof(float), st, dctx, qkv, probs, dscores, dqkv,
                       len, heads, head_dim, dtype == SPV_BF16, p_drop, seed);
    SPV_LAUNCH_CHECK("spv_attention_bwd(kv)");
    return 0;
}
