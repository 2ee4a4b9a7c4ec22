// spv_head.hip -- the classifier end of the training step: a SpectreLinear over a FEW rows (the class head: 512 CLS rows,
// 512 -> 100) as one forward and two backward launches, and mean cross-entropy as one forward and one backward launch.
//
// Why: on the generic path this 52-MFLOP problem was 21 launches per step -- CLS add, cast, fp32 weight shadow, split-K GEMM +
// reduce, tail, log-softmax, NLL (+2 fills), their backwards, two transposes, two more GEMMs -- 150 us of a 2.2 ms step, every
// launch at its floor of 4-6 us.  Nothing here is bandwidth or MFMA work: the kernels below are plain fp32 FMA loops over the
// master weights (no bf16 shadow, fp32 logits as under stock autocast), sized so that one launch covers the batch.
//
//   out = GELU(LayerNorm(x W^T + b)) + AdaptiveAvgPool1d(n)(x)          reference spectre_vit/models/spectre/layers.py:95-101
//   x   = xa[row * lda] (+ xb[row * ldb])                                the CLS row of the encoder output + the global residual's
//                                                                        CLS row (reference spectre.py:199-201: (x + src)[:, 0])
//   loss = mean_r( logsumexp(z_r) - z_r[label_r] )                       nn.CrossEntropyLoss() -- reference repl/train.py:196,226
#include "spv_common.h"

namespace {

constexpr int HR = 4;        // rows per workgroup (one wave each in the row phases)
constexpr int HT = 1024;     // threads per workgroup: 16 waves, so that ~100 KB of W is in flight per CU (a workgroup streams all of W; with
                             // 4 waves the 512 x 100 x 512 head forward took 40 us, its rows backward 21 us -- L2 latency, nothing else)
constexpr int HW = HT / 64;  // waves per workgroup
constexpr int MAXN = 128;    // outputs per row: two per lane
constexpr int MAXK = 1024;   // inputs per row (LDS: HR * MAXK floats = 16 KiB)

template <typename T> __device__ __forceinline__ float ldx(const void* p, size_t i) { return io<T>::ld(static_cast<const T*>(p) + i); }

__device__ __forceinline__ int win_start(int i, int n, int k) { return (int)(((int64_t)i * k) / n); }
__device__ __forceinline__ int win_end(int i, int n, int k) { return (int)((((int64_t)(i + 1)) * k + n - 1) / n); }

// ---- forward: grid = ceil(rows / HR).  KJ = ceil(k / 64): lane l holds columns l, l + 64, ... of its rows in registers.
// Every load a wave needs for FOUR outputs is in flight before the first is used: one output per trip is a chain of 25 L2 round
// trips per wave (measured 90 us for the 512 x 100 x 512 head).
template <typename T, int KJ>
__global__ __launch_bounds__(HT) void small_sl_fwd_kernel(const void* __restrict__ xa, int64_t lda, const void* __restrict__ xb, int64_t ldb,
                                                          const float* __restrict__ W, const float* __restrict__ bias,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ out, float* __restrict__ h, float* __restrict__ xs_out,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int rows, int n, int k) {
    __shared__ float xs[HR][MAXK];
    __shared__ float hs[HR][MAXN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * HR;
    // every load below is unconditional on a clamped address: a load inside its own `if` gets its own s_waitcnt vmcnt(0), and 64 of
    // those in a row were 30 of this kernel's first 33 us
    float xr[HR][KJ];
    const void* xb2 = xb != nullptr ? xb : xa;
    const int64_t ldb2 = xb != nullptr ? ldb : lda;
    const float bsc = xb != nullptr ? 1.0f : 0.0f;
#pragma unroll
    for (int r = 0; r < HR; ++r) {
        const int row = min(r0 + r, rows - 1);
        float va[KJ], vb[KJ];
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const int c = min(lane + 64 * j, k - 1);
            va[j] = ldx<T>(xa, (size_t)row * lda + c);
            vb[j] = ldx<T>(xb2, (size_t)row * ldb2 + c);
        }
#pragma unroll
        for (int j = 0; j < KJ; ++j) xr[r][j] = fmaf(bsc, vb[j], va[j]) * (lane + 64 * j < k ? 1.0f : 0.0f);   // a mask, not a select: a select sinks the loads into a branch
        if (KJ > 8) __builtin_amdgcn_sched_barrier(0);   // wide rows: one row's loads at a time, or the raw halves of all four spill
    }
    // wave w < HR also publishes row w: the pooled skip reads windows of it (LDS); the weight gradient reads the summed input in fp32
    // (global, stored at the end of the kernel: a store here would sit in front of the W loads in the memory counter)
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
        float v = xr[0][j];
#pragma unroll
        for (int r = 1; r < HR; ++r) v = wave == r ? xr[r][j] : v;
        if (wave < HR) xs[wave][min(lane + 64 * j, MAXK - 1)] = v;
    }
    // h = x W^T + b: wave w takes outputs w, w + 16, ...; lanes split K
    constexpr int OB = KJ <= 8 ? 7 : (KJ <= 12 ? 3 : 2);  // outputs per batch of loads (56 / 32 registers of W in flight: 128 VGPRs per lane at 16 waves; 7 x 16 waves covers n <= 112 in one trip)
    for (int o0 = wave; o0 < n; o0 += OB * HW) {
        float w[OB][KJ], bv[OB];
#pragma unroll
        for (int i = 0; i < OB; ++i) {
            const int o = min(o0 + i * HW, n - 1);
            bv[i] = bias[o];
#pragma unroll
            for (int j = 0; j < KJ; ++j) w[i][j] = W[(size_t)o * k + min(lane + 64 * j, k - 1)];
        }
#pragma unroll
        for (int i = 0; i < OB; ++i) {
            const int o = o0 + i * HW;
            float a[HR];
#pragma unroll
            for (int r = 0; r < HR; ++r) {
                a[r] = 0.0f;
#pragma unroll
                for (int j = 0; j < KJ; ++j) a[r] = fmaf(xr[r][j], w[i][j], a[r]);   // columns >= k: xr is zero there
                a[r] = wave_sum(a[r]);
            }
            if (lane == 0 && o < n) {
#pragma unroll
                for (int r = 0; r < HR; ++r) hs[r][o] = a[r] + bv[i];
            }
        }
    }
    __syncthreads();
    // LayerNorm + GELU + pooled skip: wave r owns row r, lane owns outputs lane and lane + 64
    const int row = r0 + wave;
    if (wave >= HR || row >= rows) return;
    float hv[2], s = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int o = lane + 64 * j;
        hv[j] = o < n ? hs[wave][o] : 0.0f;
        s += hv[j];
    }
    const float mu = wave_sum(s) / (float)n;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int o = lane + 64 * j;
        const float d = o < n ? hv[j] - mu : 0.0f;
        q += d * d;
    }
    const float rs = rsqrtf(wave_sum(q) / (float)n + 1e-5f);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int o = lane + 64 * j;
        if (o < n) {
            const float ln = (hv[j] - mu) * rs * gamma[o] + beta[o];
            float skip;
            if (n == k) skip = xs[wave][o];
            else {
                const int s0 = win_start(o, n, k), e0 = win_end(o, n, k);
                float t = 0.0f;
                for (int c = s0; c < e0; ++c) t += xs[wave][c];
                skip = t / (float)(e0 - s0);
            }
            out[(size_t)row * n + o] = gelu_erf(ln) + skip;
            h[(size_t)row * n + o] = hv[j];
        }
    }
#pragma unroll
    for (int j = 0; j < KJ; ++j)
        if (lane + 64 * j < k) xs_out[(size_t)row * k + lane + 64 * j] = xs[wave][lane + 64 * j];
}

// ---- backward, rows: dh (for the weight gradient), dx, per-workgroup partial sums of dgamma / dbeta / dbias
template <typename T>
__global__ __launch_bounds__(HT) void small_sl_bwd_rows_kernel(const float* __restrict__ dout, const float* __restrict__ h,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ W, float* __restrict__ dh, void* __restrict__ dx,
                                                               float* __restrict__ partials, int rows, int n, int k) {
    __shared__ float dhs[HR][MAXN];
    __shared__ float dos[HR][MAXN];
    __shared__ float part[HR][3][MAXN];
    __shared__ float joinbuf[HT * HR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * HR;
    const int row = r0 + wave;
    if (wave < HR) {
        const bool live = row < rows;
        const float mu = live ? mean[row] : 0.0f, rs = live ? rstd[row] : 0.0f;
        float t[2], xh[2], dl[2], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = lane + 64 * j;
            t[j] = xh[j] = dl[j] = 0.0f;
            if (live && o < n) {
                const float g = gamma[o];
                const float d = dout[(size_t)row * n + o];
                xh[j] = (h[(size_t)row * n + o] - mu) * rs;
                dl[j] = d * gelu_erf_grad(xh[j] * g + beta[o]);
                t[j] = dl[j] * g;
                s1 += t[j];
                s2 += t[j] * xh[j];
                dos[wave][o] = d;
            }
        }
        const float m1 = wave_sum(s1) / (float)n, m2 = wave_sum(s2) / (float)n;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = lane + 64 * j;
            if (o < n) {
                const float v = live ? rs * (t[j] - m1 - xh[j] * m2) : 0.0f;
                dhs[wave][o] = v;
                if (!live) dos[wave][o] = 0.0f;
                part[wave][0][o] = dl[j] * xh[j];
                part[wave][1][o] = dl[j];
                part[wave][2][o] = v;
                if (live) dh[(size_t)row * n + o] = v;
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < 3 * n; e += HT) {
        const int w = e / n, o = e - w * n;
        float a = 0.0f;
#pragma unroll
        for (int r = 0; r < HR; ++r) a += part[r][w][o];
        partials[((size_t)blockIdx.x * 3 + w) * n + o] = a;
    }
    // dx = dh W + pool^T(dout).  Thread = (column c, output slice q): the 1024 threads cover min(k, 1024) columns x (1024 / columns)
    // slices of the outputs, each slice's W column fetched in one batch; the slices are joined through LDS in a fixed order.
    const int cols = min(k, HT);
    const int nq = HT / cols;                 // output slices (k = 512: 2)
    const int c = tid % cols, qs = tid / cols;
    constexpr int LB = 32;                    // loads in flight per thread
    float a[HR];
#pragma unroll
    for (int r = 0; r < HR; ++r) a[r] = 0.0f;
    if (qs < nq) {
        for (int o0 = qs; o0 < n; o0 += LB * nq) {
            float w[LB];
#pragma unroll
            for (int j = 0; j < LB; ++j) {
                w[j] = W[(size_t)min(o0 + j * nq, n - 1) * k + c];   // clamped, unconditional (outputs >= n are masked below)
            }
#pragma unroll
            for (int j = 0; j < LB; ++j) {
                const int o = o0 + j * nq;
                const float wv = o < n ? w[j] : 0.0f;
                const int oc = min(o, n - 1);
#pragma unroll
                for (int r = 0; r < HR; ++r) a[r] = fmaf(dhs[r][oc], wv, a[r]);
            }
        }
    }
    float* join = reinterpret_cast<float*>(joinbuf);   // [nq][HR][cols] <= 1024 * HR floats
    if (qs < nq) {
#pragma unroll
        for (int r = 0; r < HR; ++r) join[((size_t)qs * HR + r) * cols + c] = a[r];
    }
    __syncthreads();
    if (tid < cols) {
        float o4[HR];
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            float t = 0.0f;
            for (int q2 = 0; q2 < nq; ++q2) t += join[((size_t)q2 * HR + r) * cols + c];
            o4[r] = t;
        }
        if (n == k) {
#pragma unroll
            for (int r = 0; r < HR; ++r) o4[r] += dos[r][c];
        } else {
            // windows [floor(i k / n), ceil((i + 1) k / n)) that hold column c: i0 = floor(c n / k) and its neighbours (n <= k)
            const int i0 = (int)(((int64_t)c * n) / k);
            for (int i = max(i0 - 1, 0); i <= min(i0 + 1, n - 1); ++i) {
                const int s0 = win_start(i, n, k), e0 = win_end(i, n, k);
                if (c >= s0 && c < e0) {
                    const float inv = 1.0f / (float)(e0 - s0);
#pragma unroll
                    for (int r = 0; r < HR; ++r) o4[r] = fmaf(dos[r][i], inv, o4[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < HR; ++r)
            if (r0 + r < rows) io<T>::st(static_cast<T*>(dx) + (size_t)(r0 + r) * k + c, o4[r]);
    }
}

// ---- backward, weights: workgroup (output pair, 128-column block): 4 row groups x 128 columns, dW[o][c] = sum_r dh[r][o] x[r][c] with the
// row groups' sums joined through LDS in a fixed order; the last workgroup folds dgamma / dbeta / dbias from the partials
constexpr int WT = 1024, WCOLS = 128, WGRP = WT / WCOLS, WOUT = 2;
__global__ __launch_bounds__(WT) void small_sl_bwd_w_kernel(const float* __restrict__ dh, const float* __restrict__ xs,
                                                            const float* __restrict__ partials, float* __restrict__ dW,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias,
                                                            int rows, int n, int k, int nparts, int cblocks) {
    __shared__ float red[WGRP][WOUT][WCOLS];
    const int tid = threadIdx.x;
    const int njobs = ((n + WOUT - 1) / WOUT) * cblocks;
    if ((int)blockIdx.x == njobs) {
        // fold: thread (column e, part slice q) -- 16 slices of the partials per column, joined in a fixed order
        constexpr int SL = 16;
        float* fold = &red[0][0][0];   // [SL][64] floats per pass
        for (int e0 = 0; e0 < 3 * n; e0 += 64) {
            const int e = e0 + (tid & 63), q = tid >> 6;
            float a = 0.0f;
            if (e < 3 * n) {
                const int w = e / n, c = e - w * n;
                for (int p0 = q; p0 < nparts; p0 += SL * 8) {   // eight independent loads in flight
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int pp = p0 + j * SL;
                        v[j] = partials[((size_t)min(pp, nparts - 1) * 3 + w) * n + c] * (pp < nparts ? 1.0f : 0.0f);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) a += v[j];
                }
            }
            fold[q * 64 + (tid & 63)] = a;
            __syncthreads();
            if (tid < 64 && e < 3 * n) {
                float t = 0.0f;
#pragma unroll
                for (int j = 0; j < SL; ++j) t += fold[j * 64 + tid];
                const int w = e / n, c = e - w * n;
                (w == 0 ? dgamma : (w == 1 ? dbeta : dbias))[c] = t;
            }
            __syncthreads();
        }
        return;
    }
    const int o0 = ((int)blockIdx.x / cblocks) * WOUT, c = ((int)blockIdx.x % cblocks) * WCOLS + (tid & (WCOLS - 1));
    const int g = __builtin_amdgcn_readfirstlane(tid / WCOLS);   // wave-uniform: the dh reads below are scalar loads
    const int o1 = min(o0 + 1, n - 1);
    const int cc = min(c, k - 1);
    float a0 = 0.0f, a1 = 0.0f;
    constexpr int RB = 32;   // rows in flight
    for (int rb = g; rb < rows; rb += WGRP * RB) {
        float x[RB], d0[RB], d1[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int r = rb + j * WGRP;
            const int rr = min(r, rows - 1);
            x[j] = xs[(size_t)rr * k + cc] * (r < rows ? 1.0f : 0.0f);
            d0[j] = dh[(size_t)rr * n + o0];
            d1[j] = dh[(size_t)rr * n + o1];
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            a0 = fmaf(d0[j], x[j], a0);
            a1 = fmaf(d1[j], x[j], a1);
        }
    }
    red[g][0][tid & (WCOLS - 1)] = a0;
    red[g][1][tid & (WCOLS - 1)] = a1;
    __syncthreads();
    if (tid < WOUT * WCOLS) {
        const int w = tid / WCOLS, cl = tid & (WCOLS - 1);
        const int o = o0 + w, col = ((int)blockIdx.x % cblocks) * WCOLS + cl;
        float t = 0.0f;
#pragma unroll
        for (int j = 0; j < WGRP; ++j) t += red[j][w][cl];
        if (o < n && col < k) dW[(size_t)o * k + col] = t;
    }
}

// ---- cross-entropy, mean over rows.  forward: wave per row; per-workgroup partial sums; the last workgroup to finish adds them in
// index order (deterministic) and re-arms the counter.
__global__ __launch_bounds__(HT) void ce_fwd_kernel(const float* __restrict__ z, const int64_t* __restrict__ labels, float* __restrict__ lse,
                                                    float* __restrict__ loss, float* __restrict__ partial, unsigned* __restrict__ counter,
                                                    int rows, int C) {
    __shared__ float ws[HT / 64];
    __shared__ bool last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc = 0.0f;
    for (int r = blockIdx.x * (HT / 64) + wave; r < rows; r += gridDim.x * (HT / 64)) {
        const float* zr = z + (size_t)r * C;
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, zr[c]);
        m = wave_max(m);
        float s = 0.0f;
        for (int c = lane; c < C; c += 64) s += __expf(zr[c] - m);
        const float l = m + __logf(wave_sum(s));
        const int64_t y = labels[r];
        if (lane == 0) {
            lse[r] = l;
            acc += (y >= 0 && y < C) ? l - zr[y] : __builtin_nanf("");   // a label outside [0, C) poisons the loss instead of reading wild
        }
    }
    if (lane == 0) ws[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        float t = 0.0f;
        for (int w = 0; w < HT / 64; ++w) t += ws[w];
        partial[blockIdx.x] = t;
        __threadfence();
        last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {
        __threadfence();
        // the partial sums, one per lane of wave 0 (gridDim.x <= 64), joined by the same tree every time
        if (wave == 0) {
            const float t = wave_sum(lane < (int)gridDim.x ? __builtin_nontemporal_load(partial + lane) : 0.0f);
            if (lane == 0) {
                *loss = t / (float)rows;
                *counter = 0u;
            }
        }
    }
}

// dz = (softmax(z) - onehot) * go / rows
__global__ __launch_bounds__(HT) void ce_bwd_kernel(const float* __restrict__ z, const int64_t* __restrict__ labels, const float* __restrict__ lse,
                                                    const float* __restrict__ go, float* __restrict__ dz, int rows, int C) {
    const float scale = go[0] / (float)rows;
    const int64_t total = (int64_t)rows * C;
    for (int64_t e = (int64_t)blockIdx.x * HT + threadIdx.x; e < total; e += (int64_t)gridDim.x * HT) {
        const int r = (int)(e / C), c = (int)(e - (int64_t)r * C);
        const float p = __expf(z[e] - lse[r]);
        dz[e] = (p - (labels[r] == c ? 1.0f : 0.0f)) * scale;
    }
}

}  // namespace

extern "C" int spv_small_sl_supported(int rows, int n, int k) {
    static const bool off = SPV_LAB_SET("SPV_NO_SMALL_HEAD");  // A/B switch: the generic SpectreLinear path
    return (!off && rows > 0 && rows <= 4096 && n > 0 && n <= MAXN && k > 0 && k <= MAXK && n <= k) ? 1 : 0;
}
extern "C" int64_t spv_small_sl_partial_floats(int rows, int n) { return (int64_t)cdiv(rows, HR) * 3 * n; }

extern "C" int spv_small_sl_fwd(const void* xa, int64_t lda, const void* xb, int64_t ldb, const float* W, const float* bias, const float* gamma,
                                const float* beta, float* out, float* h, float* xs, float* mean, float* rstd, int rows, int n, int k,
                                int dtype, void* stream) {
    SPV_CHECK(spv_small_sl_supported(rows, n, k), "spv_small_sl_fwd: rows=%d n=%d k=%d outside the small-rows kernel", rows, n, k);
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_small_sl_fwd: bad dtype %d", dtype);
    SPV_CHECK(lda >= k && (xb == nullptr || ldb >= k), "spv_small_sl_fwd: row strides below k");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(cdiv(rows, HR));
#define SPV_HEAD_FWD(TY, KJV) hipLaunchKernelGGL((small_sl_fwd_kernel<TY, KJV>), grid, dim3(HT), 0, st, xa, lda, xb, ldb, W, bias, gamma, beta, out, h, xs, mean, rstd, rows, n, k)
    if (dtype == SPV_BF16) {
        if (k <= 512) SPV_HEAD_FWD(bf16_t, 8);
        else if (k <= 768) SPV_HEAD_FWD(bf16_t, 12);
        else SPV_HEAD_FWD(bf16_t, 16);
    } else {
        if (k <= 512) SPV_HEAD_FWD(float, 8);
        else if (k <= 768) SPV_HEAD_FWD(float, 12);
        else SPV_HEAD_FWD(float, 16);
    }
#undef SPV_HEAD_FWD
    SPV_LAUNCH_CHECK("spv_small_sl_fwd");
    return 0;
}

extern "C" int spv_small_sl_bwd(const float* dout, const float* h, const float* xs, const float* mean, const float* rstd, const float* W,
                                const float* gamma, const float* beta, float* dh, void* dx, float* dW, float* dgamma, float* dbeta,
                                float* dbias, float* partials, int rows, int n, int k, int dx_dtype, void* stream) {
    SPV_CHECK(spv_small_sl_supported(rows, n, k), "spv_small_sl_bwd: rows=%d n=%d k=%d outside the small-rows kernel", rows, n, k);
    SPV_CHECK(dx_dtype == SPV_F32 || dx_dtype == SPV_BF16, "spv_small_sl_bwd: bad dtype %d", dx_dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nparts = cdiv(rows, HR);
    if (dx_dtype == SPV_BF16)
        hipLaunchKernelGGL((small_sl_bwd_rows_kernel<bf16_t>), dim3(nparts), dim3(HT), 0, st, dout, h, mean, rstd, gamma, beta, W, dh, dx, partials, rows, n, k);
    else
        hipLaunchKernelGGL((small_sl_bwd_rows_kernel<float>), dim3(nparts), dim3(HT), 0, st, dout, h, mean, rstd, gamma, beta, W, dh, dx, partials, rows, n, k);
    SPV_LAUNCH_CHECK("spv_small_sl_bwd(rows)");
    const int cblocks = cdiv(k, WCOLS);
    hipLaunchKernelGGL(small_sl_bwd_w_kernel, dim3(cdiv(n, WOUT) * cblocks + 1), dim3(WT), 0, st, dh, xs, partials, dW, dgamma, dbeta, dbias, rows,
                       n, k, nparts, cblocks);
    SPV_LAUNCH_CHECK("spv_small_sl_bwd(weights)");
    return 0;
}

constexpr int CE_MAX_WG = 64;
extern "C" int64_t spv_cross_entropy_workspace_floats() { return CE_MAX_WG + 1; }  // partial sums + the arrival counter (zeroed once by the caller)

extern "C" int spv_cross_entropy_fwd(const float* logits, const int64_t* labels, float* lse, float* loss, float* workspace, int rows, int classes,
                                     void* stream) {
    SPV_CHECK(rows > 0 && classes > 0, "spv_cross_entropy_fwd: empty");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int wgs = std::min(cdiv(rows, HT / 64), CE_MAX_WG);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(wgs), dim3(HT), 0, st, logits, labels, lse, loss, workspace, reinterpret_cast<unsigned*>(workspace + CE_MAX_WG),
                       rows, classes);
    SPV_LAUNCH_CHECK("spv_cross_entropy_fwd");
    return 0;
}

extern "C" int spv_cross_entropy_bwd(const float* logits, const int64_t* labels, const float* lse, const float* grad_out, float* dlogits, int rows,
                                     int classes, void* stream) {
    SPV_CHECK(rows > 0 && classes > 0, "spv_cross_entropy_bwd: empty");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)rows * classes;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((unsigned)std::min<int64_t>((total + HT - 1) / HT, 1024)), dim3(HT), 0, st, logits, labels, lse, grad_out,
                       dlogits, rows, classes);
    SPV_LAUNCH_CHECK("spv_cross_entropy_bwd");
    return 0;
}
