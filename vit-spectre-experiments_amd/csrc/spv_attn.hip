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
// a wave-private LDS row.  v1 (below) is the general fallback; v2 (further down) stages the operands in LDS.
#define SPV_USES_SEED
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
    const unsigned rkey = p_drop > 0.0f ? dropout_row_key(live_seed(seed), attn_row(sh, len, i)) : 0u;
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
    const unsigned rkey = p_drop > 0.0f ? dropout_row_key(live_seed(seed), attn_row(sh, len, i)) : 0u;
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
        if (p_drop > 0.0f) p *= dropout_scale(dropout_row_key(live_seed(seed), attn_row(sh, len, i)), (unsigned)k, p_drop, inv_keep);
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

// ------------------------------------------------------------------------------------------------------------------
// v2: the same arithmetic with the operands where the lanes can reach them.  K and V (forward, backward-1) or Q and dctx
// (backward-2) of one (sequence, head) are staged ONCE per workgroup into LDS with 16-byte global loads, rows padded by
// 8 bytes so that 8-byte LDS reads of consecutive rows are bank-conflict free; every inner product then reads LDS
// instead of issuing one 2-byte global load per element (v1: 30 ms per layer-pass at len 512).  Used when head_dim is
// 16 / 32 / 64 and the staged rows fit in LDS; v1 stays as the general fallback.
template <typename T> __device__ __forceinline__ void lds_ld4(const unsigned char* p, float (&v)[4]);
template <> __device__ __forceinline__ void lds_ld4<bf16_t>(const unsigned char* p, float (&v)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void lds_ld4<float>(const unsigned char* p, float (&v)[4]) {
    const float2 a = *reinterpret_cast<const float2*>(p), b = *reinterpret_cast<const float2*>(p + 8);
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}

// stage `len` rows of HD elements (global row stride gstride elements) into LDS rows of RS bytes
template <typename T, int HD>
__device__ __forceinline__ void stage_rows(const T* __restrict__ g, size_t gstride, unsigned char* lds, int len) {
    constexpr int RS = HD * (int)sizeof(T) + 8;
    constexpr int CPR = HD * (int)sizeof(T) / 8;  // 8-byte chunks per row
    for (int e = threadIdx.x; e < len * CPR; e += blockDim.x) {
        const int r = e / CPR, c = e % CPR;
        *reinterpret_cast<uint2*>(lds + (size_t)r * RS + c * 8) =
            *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(g + (size_t)r * gstride) + c * 8);
    }
}

constexpr int A2W = 8;  // waves per workgroup (forward, backward-1)

template <typename T, int HD> struct A2 {
    static constexpr int RS = HD * (int)sizeof(T) + 8;   // staged row stride in bytes
    static constexpr int NG = HD / 4, NS = 64 / NG;       // P.V phase: NG lane groups over d (4 each) x NS key slices
    static size_t lds_fwd(int len) { return (size_t)2 * len * RS + (size_t)A2W * (len + HD + NS * HD) * sizeof(float); }
    static size_t lds_kv(int len) { return (size_t)2 * len * RS + (size_t)64 * (2 * HD + 1) * sizeof(float); }
};

// ctx row = softmax(q K^T) V for the rows of one chunk; `second` selects what the rows are contracted with afterwards
template <typename T, int HD>
__global__ __launch_bounds__(64 * A2W) void attn2_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ ctx, T* __restrict__ probs,
                                                             int len, int heads, int rows_per_wave, float p_drop, uint64_t seed) {
    using P = A2<T, HD>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
    unsigned char* Ks = lds2;
    unsigned char* Vs = lds2 + (size_t)len * P::RS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* prow = reinterpret_cast<float*>(lds2 + (size_t)2 * len * P::RS) + (size_t)wave * (len + HD + P::NS * HD);
    float* qrow = prow + len;
    float* red = qrow + HD;
    const int sh = blockIdx.x, s = sh / heads, h = sh % heads, E = heads * HD;
    const T* base = qkv + (size_t)s * len * 3 * E + h * HD;
    stage_rows<T, HD>(base + E, (size_t)3 * E, Ks, len);
    stage_rows<T, HD>(base + 2 * E, (size_t)3 * E, Vs, len);
    __syncthreads();
    const float scale = rsqrtf((float)HD);
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const int i0 = (blockIdx.y * A2W + wave) * rows_per_wave;
    for (int i = i0; i < min(i0 + rows_per_wave, len); ++i) {
        if (lane < HD) qrow[lane] = (float)io<T>::ld(base + (size_t)i * 3 * E + lane) * scale;
        lds_sync();
        float qv[HD];
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const float4 t4 = *reinterpret_cast<const float4*>(qrow + d);
            qv[d] = t4.x; qv[d + 1] = t4.y; qv[d + 2] = t4.z; qv[d + 3] = t4.w;
        }
        float sc[MAXC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int k = c * 64 + lane;
            sc[c] = -INFINITY;
            if (c * 64 < len && k < len) {
                const unsigned char* kr = Ks + (size_t)k * P::RS;
                float a = 0.0f;
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    float kv[4];
                    lds_ld4<T>(kr + d * sizeof(T), kv);
                    a = fmaf(qv[d], kv[0], a); a = fmaf(qv[d + 1], kv[1], a); a = fmaf(qv[d + 2], kv[2], a); a = fmaf(qv[d + 3], kv[3], a);
                }
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
        const float inv = 1.0f / sum;
        const unsigned rkey = p_drop > 0.0f ? dropout_row_key(live_seed(seed), attn_row(sh, len, i)) : 0u;
        T* pg = probs + ((size_t)sh * len + i) * len;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int k = c * 64 + lane;
            if (c * 64 < len && k < len) {
                const float pv = sc[c] * inv;
                io<T>::st(pg + k, pv);
                prow[k] = p_drop > 0.0f ? pv * dropout_scale(rkey, (unsigned)k, p_drop, inv_keep) : pv;
            }
        }
        lds_sync();
        // ctx[i][:] = sum_k p[k] V[k][:]: lane = (d group of 4, key slice); slices meet in LDS
        const int dg = lane % P::NG, ks = lane / P::NG;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int k = ks; k < len; k += P::NS) {
            float vv[4];
            lds_ld4<T>(Vs + (size_t)k * P::RS + dg * 4 * sizeof(T), vv);
            const float pk = prow[k];
            acc[0] = fmaf(pk, vv[0], acc[0]); acc[1] = fmaf(pk, vv[1], acc[1]); acc[2] = fmaf(pk, vv[2], acc[2]); acc[3] = fmaf(pk, vv[3], acc[3]);
        }
        *reinterpret_cast<float4*>(red + ks * HD + dg * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        lds_sync();
        if (lane < HD) {
            float o = 0.0f;
#pragma unroll
            for (int q = 0; q < P::NS; ++q) o += red[q * HD + lane];
            io<T>::st(ctx + ((size_t)s * len + i) * E + h * HD + lane, o);
        }
        lds_sync();  // qrow / prow / red are rewritten by the next row
    }
}

// backward 1: dS[i][:] = P (dP - <P, dP>) / sqrt(hd), dP = mask * (dctx[i] . V^T); dQ[i] = dS[i] . K
template <typename T, int HD>
__global__ __launch_bounds__(64 * A2W) void attn2_bwd_q_kernel(const T* __restrict__ dctx, const T* __restrict__ qkv,
                                                               const T* __restrict__ probs, T* __restrict__ ds, T* __restrict__ dqkv,
                                                               int len, int heads, int rows_per_wave, float p_drop, uint64_t seed) {
    using P = A2<T, HD>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
    unsigned char* Ks = lds2;
    unsigned char* Vs = lds2 + (size_t)len * P::RS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* srow = reinterpret_cast<float*>(lds2 + (size_t)2 * len * P::RS) + (size_t)wave * (len + HD + P::NS * HD);
    float* grow = srow + len;
    float* red = grow + HD;
    const int sh = blockIdx.x, s = sh / heads, h = sh % heads, E = heads * HD;
    const T* base = qkv + (size_t)s * len * 3 * E + h * HD;
    stage_rows<T, HD>(base + E, (size_t)3 * E, Ks, len);
    stage_rows<T, HD>(base + 2 * E, (size_t)3 * E, Vs, len);
    __syncthreads();
    const float scale = rsqrtf((float)HD);
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const int i0 = (blockIdx.y * A2W + wave) * rows_per_wave;
    for (int i = i0; i < min(i0 + rows_per_wave, len); ++i) {
        if (lane < HD) grow[lane] = (float)io<T>::ld(dctx + ((size_t)s * len + i) * E + h * HD + lane);
        lds_sync();
        float gv[HD];
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            const float4 t4 = *reinterpret_cast<const float4*>(grow + d);
            gv[d] = t4.x; gv[d + 1] = t4.y; gv[d + 2] = t4.z; gv[d + 3] = t4.w;
        }
        const unsigned rkey = p_drop > 0.0f ? dropout_row_key(live_seed(seed), attn_row(sh, len, i)) : 0u;
        const size_t pg = ((size_t)sh * len + i) * len;
        float dp[MAXC], pr[MAXC];
        float dot = 0.0f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int k = c * 64 + lane;
            dp[c] = 0.0f;
            pr[c] = 0.0f;
            if (c * 64 < len && k < len) {
                const unsigned char* vr = Vs + (size_t)k * P::RS;
                float a = 0.0f;
#pragma unroll
                for (int d = 0; d < HD; d += 4) {
                    float vv[4];
                    lds_ld4<T>(vr + d * sizeof(T), vv);
                    a = fmaf(gv[d], vv[0], a); a = fmaf(gv[d + 1], vv[1], a); a = fmaf(gv[d + 2], vv[2], a); a = fmaf(gv[d + 3], vv[3], a);
                }
                if (p_drop > 0.0f) a *= dropout_scale(rkey, (unsigned)k, p_drop, inv_keep);
                dp[c] = a;
                pr[c] = (float)io<T>::ld(probs + pg + k);
                dot += pr[c] * a;
            }
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int k = c * 64 + lane;
            if (c * 64 < len && k < len) {
                const float v = pr[c] * (dp[c] - dot) * scale;
                io<T>::st(ds + pg + k, v);
                srow[k] = v;
            }
        }
        lds_sync();
        const int dg = lane % P::NG, ks = lane / P::NG;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int k = ks; k < len; k += P::NS) {
            float kv[4];
            lds_ld4<T>(Ks + (size_t)k * P::RS + dg * 4 * sizeof(T), kv);
            const float sk = srow[k];
            acc[0] = fmaf(sk, kv[0], acc[0]); acc[1] = fmaf(sk, kv[1], acc[1]); acc[2] = fmaf(sk, kv[2], acc[2]); acc[3] = fmaf(sk, kv[3], acc[3]);
        }
        *reinterpret_cast<float4*>(red + ks * HD + dg * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        lds_sync();
        if (lane < HD) {
            float o = 0.0f;
#pragma unroll
            for (int q = 0; q < P::NS; ++q) o += red[q * HD + lane];
            io<T>::st(dqkv + ((size_t)s * len + i) * 3 * E + h * HD + lane, o);
        }
        lds_sync();
    }
}

// backward 2: one workgroup (4 waves) per 64 key rows, lane = key; the waves split the query rows, read P[i][keys] and
// dS[i][keys] coalesced, and accumulate dV[key][:] += (mask P)[i][key] dctx[i][:], dK[key][:] += dS[i][key] Q[i][:] in
// registers (Q and dctx rows broadcast from LDS); the four partial sums meet in LDS in wave order.
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn2_bwd_kv_kernel(const T* __restrict__ dctx, const T* __restrict__ qkv,
                                                           const T* __restrict__ probs, const T* __restrict__ ds, T* __restrict__ dqkv,
                                                           int len, int heads, float p_drop, uint64_t seed) {
    using P = A2<T, HD>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
    unsigned char* Qs = lds2;
    unsigned char* Gs = lds2 + (size_t)len * P::RS;
    float* red = reinterpret_cast<float*>(lds2 + (size_t)2 * len * P::RS);  // [64 keys][2 HD + 1]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sh = blockIdx.x, s = sh / heads, h = sh % heads, E = heads * HD;
    stage_rows<T, HD>(qkv + (size_t)s * len * 3 * E + h * HD, (size_t)3 * E, Qs, len);
    stage_rows<T, HD>(dctx + (size_t)s * len * E + h * HD, (size_t)E, Gs, len);
    __syncthreads();
    const int key = blockIdx.y * 64 + lane;
    const bool live = key < len;
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    float ak[HD], av[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { ak[d] = 0.0f; av[d] = 0.0f; }
    for (int i = wave; i < len; i += 4) {
        const size_t e = ((size_t)sh * len + i) * len + key;
        float pv = 0.0f, sv = 0.0f;
        if (live) {
            pv = (float)io<T>::ld(probs + e);
            sv = (float)io<T>::ld(ds + e);
            if (p_drop > 0.0f) pv *= dropout_scale(dropout_row_key(live_seed(seed), attn_row(sh, len, i)), (unsigned)key, p_drop, inv_keep);
        }
        const unsigned char* qr = Qs + (size_t)i * P::RS;
        const unsigned char* gr = Gs + (size_t)i * P::RS;
#pragma unroll
        for (int d = 0; d < HD; d += 4) {
            float qv[4], gv[4];
            lds_ld4<T>(qr + d * sizeof(T), qv);  // same address in every lane: LDS broadcast
            lds_ld4<T>(gr + d * sizeof(T), gv);
#pragma unroll
            for (int u = 0; u < 4; ++u) { ak[d + u] = fmaf(sv, qv[u], ak[d + u]); av[d + u] = fmaf(pv, gv[u], av[d + u]); }
        }
    }
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
            float* r = red + (size_t)lane * (2 * HD + 1);
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                r[d] = (w == 0 ? 0.0f : r[d]) + ak[d];
                r[HD + d] = (w == 0 ? 0.0f : r[HD + d]) + av[d];
            }
        }
        __syncthreads();
    }
    // write dK | dV rows: thread t -> (key t / 4, quarter of the 2 HD columns)
    for (int e = threadIdx.x; e < 64 * 2 * HD; e += 256) {
        const int kk = e / (2 * HD), c = e % (2 * HD);
        const int kg = blockIdx.y * 64 + kk;
        if (kg < len) io<T>::st(dqkv + ((size_t)s * len + kg) * 3 * E + (c < HD ? E : 2 * E - HD) + h * HD + c, red[(size_t)kk * (2 * HD + 1) + c]);
    }
}

constexpr size_t A2_LDS_MAX = 150 * 1024;

template <typename T, int HD>
int launch_attn2_fwd(const void* qkv, void* ctx, void* probs, int seqs, int len, int heads, float p_drop, uint64_t seed, hipStream_t st) {
    const size_t lds = A2<T, HD>::lds_fwd(len);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_fwd_kernel<T, HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int rpw = std::max(1, std::min(16, cdiv(len, A2W)));
    dim3 grid(seqs * heads, cdiv(len, A2W * rpw));
    hipLaunchKernelGGL((attn2_fwd_kernel<T, HD>), grid, dim3(64 * A2W), lds, st, (const T*)qkv, (T*)ctx, (T*)probs, len, heads, rpw, p_drop, seed);
    SPV_LAUNCH_CHECK("spv_attention_fwd(v2)");
    return 0;
}
template <typename T, int HD>
int launch_attn2_bwd(const void* dctx, const void* qkv, const void* probs, void* ds, void* dqkv, int seqs, int len, int heads,
                     float p_drop, uint64_t seed, hipStream_t st) {
    const size_t lds = A2<T, HD>::lds_fwd(len), lds_kv = A2<T, HD>::lds_kv(len);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_bwd_q_kernel<T, HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_bwd_kv_kernel<T, HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
    const int rpw = std::max(1, std::min(16, cdiv(len, A2W)));
    dim3 grid(seqs * heads, cdiv(len, A2W * rpw));
    hipLaunchKernelGGL((attn2_bwd_q_kernel<T, HD>), grid, dim3(64 * A2W), lds, st, (const T*)dctx, (const T*)qkv, (const T*)probs, (T*)ds,
                       (T*)dqkv, len, heads, rpw, p_drop, seed);
    SPV_LAUNCH_CHECK("spv_attention_bwd(v2 q)");
    hipLaunchKernelGGL((attn2_bwd_kv_kernel<T, HD>), dim3(seqs * heads, cdiv(len, 64)), dim3(256), lds_kv, st, (const T*)dctx,
                       (const T*)qkv, (const T*)probs, (const T*)ds, (T*)dqkv, len, heads, p_drop, seed);
    SPV_LAUNCH_CHECK("spv_attention_bwd(v2 kv)");
    return 0;
}

// v2 applies when head_dim is 16 / 32 / 64, rows are 8-byte aligned and the staged operands fit in LDS
template <typename T> inline bool attn2_ok(int len, int heads, int hd, const void* a, const void* b) {
    if (!(hd == 16 || hd == 32 || hd == 64)) return false;
    if (((uintptr_t)a & 7) || ((uintptr_t)b & 7) || ((size_t)heads * hd * sizeof(T)) % 8) return false;
    const size_t rs = hd * sizeof(T) + 8;
    const size_t f = (size_t)2 * len * rs + (size_t)A2W * (len + hd + 64 / (hd / 4) * hd) * sizeof(float);
    const size_t kv = (size_t)2 * len * rs + (size_t)64 * (2 * hd + 1) * sizeof(float);
    return f <= A2_LDS_MAX && kv <= A2_LDS_MAX;
}

#define A2_DISPATCH(T, hd, FN, ...)                          \
    do {                                                     \
        if (hd == 16) return FN<T, 16>(__VA_ARGS__);         \
        if (hd == 32) return FN<T, 32>(__VA_ARGS__);         \
        return FN<T, 64>(__VA_ARGS__);                       \
    } while (0)

// ------------------------------------------------------------------------------------------------------------------
// v3 (bf16, head_dim 32 or 64): both contractions on MFMA, nothing of size len x len stored.
//
// A wave owns 32 queries (forward, backward-1) or 32 keys (backward-2) and walks the other axis in blocks of 32.  The
// score block is computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x16_bf16, 2 per block), so a lane owns one query
// (forward / backward-1) resp. one key (backward-2) and its 16 accumulator registers are 16 entries of that row:
// softmax statistics are in-register maxima / sums plus one exchange between the two 32-lane halves.  The register
// layout of the accumulator (row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) is, 8 registers at a time, exactly an MFMA
// operand whose k slot (g, j) holds entry 16 t + 4 g + (j & 3) + 8 (j >> 2): P (or dS) goes into the second product
// straight from registers, and the other operand is read from a TRANSPOSED LDS copy (V^T, K^T, Q^T, dctx^T: rows = head
// dim, columns = sequence, two 8-byte reads per k16 step at sequence offsets 16 t + 4 g and + 8).
// Forward: pass 1 row max / sum, pass 2 P = exp(s - m) / l -> O^T += V^T P^T; the row log-sum-exp is kept instead of P.
// Backward recomputes P from it: backward-1 (per query) delta = sum_k P dP, dS = P (dP - delta) / sqrt(hd), dQ^T += K^T dS^T;
// backward-2 (per key) dV^T.. as dV = (mask P)^T dctx, dK = dS^T Q with P / dS as the register operand.
// The `probs` / `dscores` arguments of the C-ABI are used as scratch for lse / delta (seqs*heads*len floats each).
typedef __attribute__((ext_vector_type(4))) short a3s4;
typedef __attribute__((ext_vector_type(8))) short a3s8;

__device__ __forceinline__ bf16x8 a3_ld16(const bf16_t* p) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
}
// two 8-byte LDS reads -> one k16 operand (sequence entries +0..3 and +8..11)
__device__ __forceinline__ bf16x8 a3_ldt(const bf16_t* row) {
    const a3s4 lo = *reinterpret_cast<const a3s4*>(row), hi = *reinterpret_cast<const a3s4*>(row + 8);
    const a3s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 a3_pack(const float (&v)[16], int t) {
    a3s8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (short)f2bf(v[8 * t + j]);
    return __builtin_bit_cast(bf16x8, o);
}
// stage src rows [len][HD] (row stride gstride elements) transposed into dst[HD][LP], zero beyond len (LR = LP - 4 columns)
template <int HD>
__device__ __forceinline__ void a3_stage_t(const bf16_t* __restrict__ src, size_t gstride, bf16_t* dst, int len, int LP) {
    constexpr int CPR = HD / 8;  // 16-byte chunks per row
    const int LR = LP - 4;
    for (int e = threadIdx.x; e < LR * CPR; e += blockDim.x) {
        const int key = e / CPR, c = e % CPR;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (key < len) v = *reinterpret_cast<const uint4*>(src + (size_t)key * gstride + c * 8);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            dst[(size_t)(c * 8 + 2 * u) * LP + key] = (bf16_t)(w[u] & 0xffffu);
            dst[(size_t)(c * 8 + 2 * u + 1) * LP + key] = (bf16_t)(w[u] >> 16);
        }
    }
}
// entry index inside a 32-block of accumulator register r for lane half g
__device__ __forceinline__ int a3_idx(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

__device__ __forceinline__ f32x16 a3_zero() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
    return z;
}
// write one lane's 16 accumulator entries (4 runs of 4 consecutive columns) of a 32-wide block as 8-byte stores
__device__ __forceinline__ void a3_store_block(bf16_t* o, const f32x16& v) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        uint2 w;
        w.x = pack_bf16x2(v[4 * qd], v[4 * qd + 1]);
        w.y = pack_bf16x2(v[4 * qd + 2], v[4 * qd + 3]);
        *reinterpret_cast<uint2*>(o + 8 * qd) = w;
    }
}

// HD = head dim (32 or 64): KS = HD / 16 MFMA k steps for the score products, DB = HD / 32 output blocks of the second products
template <int HD>
__global__ __launch_bounds__(256, (HD == 32 ? 4 : 3)) void attn3_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx, float* __restrict__ lse,
                                                        int len, int heads, float p_drop, uint64_t seed) {
    constexpr int KS = HD / 16, DB = HD / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    bf16_t* Vt = reinterpret_cast<bf16_t*>(lds3);
    const int LR = (len + 31) & ~31, LP = LR + 4;
    const int sh = blockIdx.x, s = sh / heads, h = sh % heads, E = heads * HD;
    const bf16_t* base = qkv + (size_t)s * len * 3 * E + h * HD;
    a3_stage_t<HD>(base + 2 * E, (size_t)3 * E, Vt, len, LP);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, c = lane & 31;
    const int q0 = (blockIdx.y * 4 + wave) * 32;
    if (q0 >= len) return;
    const int q = q0 + c, qi = min(q, len - 1);
    bf16x8 qb[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t) qb[t] = a3_ld16(base + (size_t)qi * 3 * E + 16 * t + 8 * g);
    const float scale = rsqrtf((float)HD);
    const int nb = LR >> 5;
    float m = -INFINITY, l = 0.0f;
    // the K rows of block kb + 1 are requested before block kb is used: the loop is otherwise one global round trip per block
    auto krow = [&](int kb) { return base + E + (size_t)min(kb * 32 + c, len - 1) * 3 * E + 8 * g; };
    bf16x8 nk[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t) nk[t] = a3_ld16(krow(0) + 16 * t);
    for (int kb = 0; kb < nb; ++kb) {
        bf16x8 ka[KS];
#pragma unroll
        for (int t = 0; t < KS; ++t) ka[t] = nk[t];
        if (kb + 1 < nb) {
#pragma unroll
            for (int t = 0; t < KS; ++t) nk[t] = a3_ld16(krow(kb + 1) + 16 * t);
        }
        f32x16 S = a3_zero();
#pragma unroll
        for (int t = 0; t < KS; ++t) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[t], qb[t], S, 0, 0, 0);
        float bm = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            S[r] = (kb * 32 + a3_idx(r, g) < len) ? S[r] * scale : -INFINITY;
            bm = fmaxf(bm, S[r]);
        }
        if (bm > -INFINITY) {
            const float mn = fmaxf(m, bm);
            float a = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) a += __expf(S[r] - mn);
            l = l * __expf(m - mn) + a;
            m = mn;
        }
    }
    {
        const float mo = __shfl_xor(m, 32), lo = __shfl_xor(l, 32);
        const float M = fmaxf(m, mo);
        l = l * __expf(m - M) + lo * __expf(mo - M);
        m = M;
    }
    const float inv = 1.0f / l;
    if (g == 0 && q < len) lse[(size_t)sh * len + q] = m + __logf(l);
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const unsigned rkey = p_drop > 0.0f ? dropout_row_key(live_seed(seed), attn_row(sh, len, qi)) : 0u;
    f32x16 O[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) O[d] = a3_zero();
#pragma unroll
    for (int t = 0; t < KS; ++t) nk[t] = a3_ld16(krow(0) + 16 * t);
    for (int kb = 0; kb < nb; ++kb) {
        bf16x8 ka[KS];
#pragma unroll
        for (int t = 0; t < KS; ++t) ka[t] = nk[t];
        if (kb + 1 < nb) {
#pragma unroll
            for (int t = 0; t < KS; ++t) nk[t] = a3_ld16(krow(kb + 1) + 16 * t);
        }
        f32x16 S = a3_zero();
#pragma unroll
        for (int t = 0; t < KS; ++t) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[t], qb[t], S, 0, 0, 0);
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + a3_idx(r, g);
            pv[r] = key < len ? __expf(S[r] * scale - m) * inv : 0.0f;
        }
        if (p_drop > 0.0f) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] *= dropout_scale(rkey, (unsigned)(kb * 32 + a3_idx(r, g)), p_drop, inv_keep);
        }
        const bf16x8 p0 = a3_pack(pv, 0), p1 = a3_pack(pv, 1);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            const bf16_t* vr = Vt + (size_t)(d * 32 + c) * LP + kb * 32 + 4 * g;
            O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3_ldt(vr), p0, O[d], 0, 0, 0);
            O[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3_ldt(vr + 16), p1, O[d], 0, 0, 0);
        }
    }
    if (q < len) {
#pragma unroll
        for (int d = 0; d < DB; ++d) a3_store_block(ctx + ((size_t)s * len + q) * E + h * HD + d * 32 + 4 * g, O[d]);
    }
}

// backward 1: per wave 32 queries.  delta[q] = sum_k P dP', dS = P (dP' - delta) / sqrt(hd), dQ^T += K^T dS^T  (dP' = mask dP)
template <int HD>
__global__ __launch_bounds__(256, (HD == 32 ? 3 : 2)) void attn3_bwd_q_kernel(const bf16_t* __restrict__ dctx, const bf16_t* __restrict__ qkv,
                                                          const float* __restrict__ lse, float* __restrict__ delta,
                                                          bf16_t* __restrict__ dqkv, int len, int heads, float p_drop, uint64_t seed) {
    constexpr int KS = HD / 16, DB = HD / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    bf16_t* Kt = reinterpret_cast<bf16_t*>(lds3);
    const int LR = (len + 31) & ~31, LP = LR + 4;
    const int sh = blockIdx.x, s = sh / heads, h = sh % heads, E = heads * HD;
    const bf16_t* base = qkv + (size_t)s * len * 3 * E + h * HD;
    a3_stage_t<HD>(base + E, (size_t)3 * E, Kt, len, LP);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, c = lane & 31;
    const int q0 = (blockIdx.y * 4 + wave) * 32;
    if (q0 >= len) return;
    const int q = q0 + c, qi = min(q, len - 1);
    const bf16_t* gr = dctx + ((size_t)s * len + qi) * E + h * HD + 8 * g;
    bf16x8 qb[KS], gb[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t) {
        qb[t] = a3_ld16(base + (size_t)qi * 3 * E + 16 * t + 8 * g);
        gb[t] = a3_ld16(gr + 16 * t);
    }
    const float scale = rsqrtf((float)HD);
    const float lq = lse[(size_t)sh * len + qi];
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const unsigned rkey = p_drop > 0.0f ? dropout_row_key(live_seed(seed), attn_row(sh, len, qi)) : 0u;
    const int nb = LR >> 5;
    float dot = 0.0f;
    f32x16 dQ[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) dQ[d] = a3_zero();
    auto krow = [&](int kb) { return base + E + (size_t)min(kb * 32 + c, len - 1) * 3 * E + 8 * g; };
    for (int pass = 0; pass < 2; ++pass) {
        bf16x8 nk[KS], nv[KS];
#pragma unroll
        for (int t = 0; t < KS; ++t) { nk[t] = a3_ld16(krow(0) + 16 * t); nv[t] = a3_ld16(krow(0) + E + 16 * t); }  // V row = K row + E
        for (int kb = 0; kb < nb; ++kb) {
            bf16x8 ka[KS], va[KS];
#pragma unroll
            for (int t = 0; t < KS; ++t) { ka[t] = nk[t]; va[t] = nv[t]; }
            if (kb + 1 < nb) {  // next block's rows in flight under this block's arithmetic
                const bf16_t* nr = krow(kb + 1);
#pragma unroll
                for (int t = 0; t < KS; ++t) { nk[t] = a3_ld16(nr + 16 * t); nv[t] = a3_ld16(nr + E + 16 * t); }
            }
            f32x16 S = a3_zero(), dP = a3_zero();
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[t], qb[t], S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[t], gb[t], dP, 0, 0, 0);
            }
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + a3_idx(r, g);
                const float pv = key < len ? __expf(S[r] * scale - lq) : 0.0f;
                float dp = dP[r];
                if (p_drop > 0.0f) dp *= dropout_scale(rkey, (unsigned)key, p_drop, inv_keep);
                if (pass == 0) dot += pv * dp;
                else ds[r] = pv * (dp - dot) * scale;
            }
            if (pass == 1) {
                const bf16x8 s0 = a3_pack(ds, 0), s1 = a3_pack(ds, 1);
#pragma unroll
                for (int d = 0; d < DB; ++d) {
                    const bf16_t* ktr = Kt + (size_t)(d * 32 + c) * LP + kb * 32 + 4 * g;
                    dQ[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3_ldt(ktr), s0, dQ[d], 0, 0, 0);
                    dQ[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3_ldt(ktr + 16), s1, dQ[d], 0, 0, 0);
                }
            }
        }
        if (pass == 0) {
            dot += __shfl_xor(dot, 32);
            if (g == 0 && q < len) delta[(size_t)sh * len + q] = dot;
        }
    }
    if (q < len) {
#pragma unroll
        for (int d = 0; d < DB; ++d) a3_store_block(dqkv + ((size_t)s * len + q) * 3 * E + h * HD + d * 32 + 4 * g, dQ[d]);
    }
}

// backward 2: per wave 32 keys.  S = Q K^T with lane = key; dV += (mask P)^T dctx, dK += dS^T Q
template <int HD>
__global__ __launch_bounds__(256, (HD == 32 ? 3 : 2)) void attn3_bwd_kv_kernel(const bf16_t* __restrict__ dctx, const bf16_t* __restrict__ qkv,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           bf16_t* __restrict__ dqkv, int len, int heads, float p_drop, uint64_t seed) {
    constexpr int KS = HD / 16, DB = HD / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
    const int LR = (len + 31) & ~31, LP = LR + 4;
    bf16_t* Qt = reinterpret_cast<bf16_t*>(lds3);
    bf16_t* Gt = Qt + (size_t)HD * LP;
    float* lse_s = reinterpret_cast<float*>(Gt + (size_t)HD * LP);
    float* del_s = lse_s + LR;
    unsigned* rk_s = reinterpret_cast<unsigned*>(del_s + LR);
    const int sh = blockIdx.x, s = sh / heads, h = sh % heads, E = heads * HD;
    const bf16_t* base = qkv + (size_t)s * len * 3 * E + h * HD;
    const bf16_t* gbase = dctx + (size_t)s * len * E + h * HD;
    a3_stage_t<HD>(base, (size_t)3 * E, Qt, len, LP);
    a3_stage_t<HD>(gbase, (size_t)E, Gt, len, LP);
    for (int i = threadIdx.x; i < LR; i += blockDim.x) {
        const bool ok = i < len;
        lse_s[i] = ok ? lse[(size_t)sh * len + i] : 0.0f;
        del_s[i] = ok ? delta[(size_t)sh * len + i] : 0.0f;
        rk_s[i] = (ok && p_drop > 0.0f) ? dropout_row_key(live_seed(seed), attn_row(sh, len, i)) : 0u;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, c = lane & 31;
    const int k0 = (blockIdx.y * 4 + wave) * 32;
    if (k0 >= len) return;
    const int key = k0 + c, ki = min(key, len - 1);
    const bf16_t* kr = base + E + (size_t)ki * 3 * E + 8 * g;
    bf16x8 kb_[KS], vb_[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t) { kb_[t] = a3_ld16(kr + 16 * t); vb_[t] = a3_ld16(kr + E + 16 * t); }
    const float scale = rsqrtf((float)HD);
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    f32x16 dK[DB], dV[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) { dK[d] = a3_zero(); dV[d] = a3_zero(); }
    const int nb = LR >> 5;
    auto qrow = [&](int qb) { return base + (size_t)min(qb * 32 + c, len - 1) * 3 * E + 8 * g; };
    auto grow = [&](int qb) { return gbase + (size_t)min(qb * 32 + c, len - 1) * E + 8 * g; };
    bf16x8 nq[KS], ng[KS];
#pragma unroll
    for (int t = 0; t < KS; ++t) { nq[t] = a3_ld16(qrow(0) + 16 * t); ng[t] = a3_ld16(grow(0) + 16 * t); }
    for (int qb = 0; qb < nb; ++qb) {
        bf16x8 qa[KS], ga[KS];
#pragma unroll
        for (int t = 0; t < KS; ++t) { qa[t] = nq[t]; ga[t] = ng[t]; }
        if (qb + 1 < nb) {  // next block's Q / dctx rows in flight under this block's arithmetic
#pragma unroll
            for (int t = 0; t < KS; ++t) { nq[t] = a3_ld16(qrow(qb + 1) + 16 * t); ng[t] = a3_ld16(grow(qb + 1) + 16 * t); }
        }
        f32x16 S = a3_zero(), dP = a3_zero();
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[t], kb_[t], S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[t], vb_[t], dP, 0, 0, 0);
        }
        float pm[16], ds[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = qb * 32 + a3_idx(r, g);  // < LR: the staged arrays are padded
            const bool ok = qq < len && key < len;
            const float pv = ok ? __expf(S[r] * scale - lse_s[qq]) : 0.0f;
            const float msk = p_drop > 0.0f ? dropout_scale(rk_s[qq], (unsigned)key, p_drop, inv_keep) : 1.0f;
            pm[r] = pv * msk;
            ds[r] = pv * (dP[r] * msk - del_s[qq]) * scale;
        }
        const bf16x8 p0 = a3_pack(pm, 0), p1 = a3_pack(pm, 1), s0 = a3_pack(ds, 0), s1 = a3_pack(ds, 1);
#pragma unroll
        for (int d = 0; d < DB; ++d) {
            const bf16_t* gt = Gt + (size_t)(d * 32 + c) * LP + qb * 32 + 4 * g;
            const bf16_t* qt = Qt + (size_t)(d * 32 + c) * LP + qb * 32 + 4 * g;
            dV[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p0, a3_ldt(gt), dV[d], 0, 0, 0);
            dV[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, a3_ldt(gt + 16), dV[d], 0, 0, 0);
            dK[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s0, a3_ldt(qt), dK[d], 0, 0, 0);
            dK[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s1, a3_ldt(qt + 16), dK[d], 0, 0, 0);
        }
    }
    // accumulators: lane = head-dim column d * 32 + c, register r <-> key k0 + idx(r, g)
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = k0 + a3_idx(r, g);
            if (kk < len) {
                bf16_t* o = dqkv + ((size_t)s * len + kk) * 3 * E + E + h * HD + d * 32 + c;
                o[0] = f2bf(dK[d][r]);
                o[E] = f2bf(dV[d][r]);
            }
        }
}

template <int HD>
int launch_attn3_fwd(const void* qkv, void* ctx, void* probs, int seqs, int len, int heads, float p_drop, uint64_t seed, hipStream_t st) {
    const int LP = ((len + 31) & ~31) + 4;
    const size_t lds = (size_t)HD * LP * 2;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_fwd_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attn3_fwd_kernel<HD>), dim3(seqs * heads, cdiv(len, 128)), dim3(256), lds, st, (const bf16_t*)qkv, (bf16_t*)ctx,
                       (float*)probs, len, heads, p_drop, seed);
    SPV_LAUNCH_CHECK("spv_attention_fwd(v3)");
    return 0;
}
template <int HD>
int launch_attn3_bwd(const void* dctx, const void* qkv, const void* probs, void* dscores, void* dqkv, int seqs, int len, int heads,
                     float p_drop, uint64_t seed, hipStream_t st) {
    const int LR = (len + 31) & ~31, LP = LR + 4;
    const size_t lds_q = (size_t)HD * LP * 2, lds_kv = (size_t)2 * HD * LP * 2 + (size_t)3 * LR * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_bwd_q_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn3_bwd_kv_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
    dim3 grid(seqs * heads, cdiv(len, 128));
    hipLaunchKernelGGL((attn3_bwd_q_kernel<HD>), grid, dim3(256), lds_q, st, (const bf16_t*)dctx, (const bf16_t*)qkv, (const float*)probs,
                       (float*)dscores, (bf16_t*)dqkv, len, heads, p_drop, seed);
    SPV_LAUNCH_CHECK("spv_attention_bwd(v3 q)");
    hipLaunchKernelGGL((attn3_bwd_kv_kernel<HD>), grid, dim3(256), lds_kv, st, (const bf16_t*)dctx, (const bf16_t*)qkv,
                       (const float*)probs, (const float*)dscores, (bf16_t*)dqkv, len, heads, p_drop, seed);
    SPV_LAUNCH_CHECK("spv_attention_bwd(v3 kv)");
    return 0;
}

inline bool attn3_ok(int len, int heads, int hd, int dtype, const void* a, const void* b, const void* c) {
    if (dtype != SPV_BF16 || !(hd == 32 || hd == 64) || len < 2 || (heads * hd) % 8) return false;
    if (((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)c & 15)) return false;
    const size_t LP = ((len + 31) & ~31) + 4;
    return 2 * (size_t)hd * LP * 2 + 3 * (LP - 4) * 4 <= 150 * 1024;
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
    if (attn3_ok(len, heads, head_dim, dtype, qkv, ctx, probs)) {
        hipStream_t st3 = static_cast<hipStream_t>(stream);
        return head_dim == 32 ? launch_attn3_fwd<32>(qkv, ctx, probs, seqs, len, heads, p_drop, seed, st3)
                              : launch_attn3_fwd<64>(qkv, ctx, probs, seqs, len, heads, p_drop, seed, st3);
    }
    if (dtype == SPV_BF16 && attn2_ok<bf16_t>(len, heads, head_dim, qkv, ctx))
        A2_DISPATCH(bf16_t, head_dim, launch_attn2_fwd, qkv, ctx, probs, seqs, len, heads, p_drop, seed, static_cast<hipStream_t>(stream));
    if (dtype == SPV_F32 && attn2_ok<float>(len, heads, head_dim, qkv, ctx))
        A2_DISPATCH(float, head_dim, launch_attn2_fwd, qkv, ctx, probs, seqs, len, heads, p_drop, seed, static_cast<hipStream_t>(stream));
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
    if (attn3_ok(len, heads, head_dim, dtype, qkv, dctx, probs) && ((uintptr_t)dscores & 15) == 0)
        return head_dim == 32 ? launch_attn3_bwd<32>(dctx, qkv, probs, dscores, dqkv, seqs, len, heads, p_drop, seed, st)
                              : launch_attn3_bwd<64>(dctx, qkv, probs, dscores, dqkv, seqs, len, heads, p_drop, seed, st);
    if (dtype == SPV_BF16 && attn2_ok<bf16_t>(len, heads, head_dim, qkv, dctx))
        A2_DISPATCH(bf16_t, head_dim, launch_attn2_bwd, dctx, qkv, probs, dscores, dqkv, seqs, len, heads, p_drop, seed, st);
    if (dtype == SPV_F32 && attn2_ok<float>(len, heads, head_dim, qkv, dctx))
        A2_DISPATCH(float, head_dim, launch_attn2_bwd, dctx, qkv, probs, dscores, dqkv, seqs, len, heads, p_drop, seed, st);
    dim3 grid(seqs * heads, cdiv(len, AW));
    hipLaunchKernelGGL(attn_bwd_q_kernel, grid, dim3(64 * AW), (size_t)AW * (len + MAXHD) * sizeof(float), st, dctx, qkv, probs, dscores,
                       dqkv, len, heads, head_dim, dtype == SPV_BF16, p_drop, seed);
    SPV_LAUNCH_CHECK("spv_attention_bwd(q)");
    hipLaunchKernelGGL(attn_bwd_kv_kernel, grid, dim3(64 * AW), (size_t)AW * 2 * len * sizeof(float), st, dctx, qkv, probs, dscores, dqkv,
                       len, heads, head_dim, dtype == SPV_BF16, p_drop, seed);
    SPV_LAUNCH_CHECK("spv_attention_bwd(kv)");
    return 0;
}

int spv_seed_ptr_set_attn(const unsigned long long* p) { return spv_seed_symbol_set(p); }
