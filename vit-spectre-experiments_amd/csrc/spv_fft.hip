// spv_fft.hip -- spectral token mixers: FNet Re(fft2) over (tokens, dim), rfft(x).real, Haar DWT.
// All HBM-bandwidth bound by design (algorithmic traffic = read x once + write y once).
//
// FNet fast path (dim a power of two >= 8, (tokens+1)*dim*4 + 8*dim <= 160 KiB, tokens <= 79):
// one 512-thread workgroup per sample keeps the whole (tokens x dim) tile in LDS as fp32:
//   A  coalesced 16/8-byte loads -> LDS
//   B  ceil(tokens/2) in-place complex Stockham FFTs of length dim (two real rows per FFT), dim/8 threads each
//   C  tokens-point real-part DFT per frequency on the VALU with wave-uniform twiddles from the scalar
//      cache; only m <= tokens/2 and k <= dim/2 are computed, the other three quadrants follow from
//      y[m,k] = y[N-m,D-k] = P-Q, y[N-m,k] = y[m,D-k] = P+Q; results go straight to HBM (coalesced).
// The math lives in spv_fft_core.h and is unit-tested on the host (tests/cpu_harness).
#include "spv_common.h"
#include "spv_fft_core.h"

namespace {

constexpr int FT = 512;  // threads per workgroup of the fast path

__device__ __forceinline__ float ld1(const void* base, size_t off, int bf) {
    return bf ? bf2f(static_cast<const bf16_t*>(base)[off]) : static_cast<const float*>(base)[off];
}
__device__ __forceinline__ void st1(void* base, size_t off, int bf, float v) {
    if (bf) static_cast<bf16_t*>(base)[off] = f2bf(v);
    else static_cast<float*>(base)[off] = v;
}

__global__ __launch_bounds__(256) void fnet_twiddle_kernel(float* __restrict__ tw, int N) {
    const int total = (N + 1) * 2 * FNET_TWS;
    const int nh1 = N / 2 + 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int m = i % FNET_TWS, cs = (i / FNET_TWS) & 1, n = i / (2 * FNET_TWS);
        float v = 0.0f;
        if (n < N && m < nh1) {
            const int r = (int)(((long long)m * n) % N);
            float s, c;
            sincospif(2.0f * (float)r / (float)N, &s, &c);
            v = cs ? s : c;
        }
        tw[i] = v;
    }
}

template <int R>
__device__ __forceinline__ void pass_radix(float* re, float* im, const float* twr, const float* twi, int D, int Ns, int t, int tpf,
                                           bool active) {
    constexpr int NB = 8 / R;  // butterflies per thread
    cpx v[NB][R];
    if (active) {
#pragma unroll
        for (int u = 0; u < NB; ++u) stockham_load<R>(re, im, twr, twi, D, Ns, t + u * tpf, v[u]);
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int u = 0; u < NB; ++u) stockham_store<R>(re, im, Ns, t + u * tpf, v[u]);
    }
    __syncthreads();
}

template <int MH>
__global__ __launch_bounds__(FT) void fnet_lds_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                      const float* __restrict__ tw2, int N, int D, int log2tpf, FftPlan plan,
                                                      int bf) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* twr = lds + (size_t)(N + 1) * D;
    float* twi = twr + D;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const size_t base = (size_t)b * N * D;

    // ---- phase A
    const int nv = (N * D) >> 2;
    for (int v = tid; v < nv; v += FT) {
        float f[4];
        if (bf) io<bf16_t>::ld4(static_cast<const bf16_t*>(x) + base + 4 * v, f);
        else io<float>::ld4(static_cast<const float*>(x) + base + 4 * v, f);
        *reinterpret_cast<float4*>(lds + 4 * v) = make_float4(f[0], f[1], f[2], f[3]);
    }
    for (int d = tid; d < D; d += FT) {
        lds[(size_t)N * D + d] = 0.0f;
        float s, c;
        sincospif(2.0f * (float)d / (float)D, &s, &c);
        twr[d] = c;
        twi[d] = -s;
    }
    __syncthreads();

    // ---- phase B: row-pair FFTs
    const int tpf = 1 << log2tpf;
    const int per_round = FT >> log2tpf;
    const int NF = (N + 1) >> 1;
    const int t = tid & (tpf - 1);
    for (int f0 = 0; f0 < NF; f0 += per_round) {
        const int f = f0 + (tid >> log2tpf);
        const bool active = f < NF;
        float* re = lds + (size_t)(2 * (active ? f : 0)) * D;
        float* im = re + D;
        for (int p = 0; p < plan.npass; ++p) {
            const int R = plan.radix[p], Ns = plan.ns[p];
            if (R == 8) pass_radix<8>(re, im, twr, twi, D, Ns, t, tpf, active);
            else if (R == 4) pass_radix<4>(re, im, twr, twi, D, Ns, t, tpf, active);
            else pass_radix<2>(re, im, twr, twi, D, Ns, t, tpf, active);
        }
    }

    // ---- phase C: token-axis DFT, m split in two halves over the two 256-thread halves
    const int nh1 = N / 2 + 1;
    const int mh = (nh1 + 1) >> 1;
    const int half = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int m_off = half * mh;
    const int m_cnt = half ? nh1 - mh : mh;
    const int Dh = D >> 1;
    for (int k = tid & 255; k < Dh; k += 256) {
        const int mk = k == 0 ? 0 : D - k;
        float P[MH], Q[MH];
#pragma unroll
        for (int m = 0; m < MH; ++m) { P[m] = 0.0f; Q[m] = 0.0f; }
        for (int f = 0; f < NF; ++f) {
            const float* zr = lds + (size_t)(2 * f) * D;
            const float* zi = zr + D;
            float a1, b1, a2, b2;
            unpack_pair(zr[k], zr[mk], zi[k], zi[mk], a1, b1, a2, b2);
            const float* t1 = tw2 + (size_t)(2 * f) * 2 * FNET_TWS + m_off;  // wave-uniform -> scalar loads
            const float* t2 = t1 + 2 * FNET_TWS;
#pragma unroll
            for (int m = 0; m < MH; ++m) {
                P[m] = fmaf(t1[m], a1, fmaf(t2[m], a2, P[m]));
                Q[m] = fmaf(t1[FNET_TWS + m], b1, fmaf(t2[FNET_TWS + m], b2, Q[m]));
            }
        }
#pragma unroll
        for (int m = 0; m < MH; ++m) {
            if (m < m_cnt) {
                const int M = m_off + m;
                const int mm = M == 0 ? 0 : N - M;
                const float d = P[m] - Q[m], s = P[m] + Q[m];
                st1(y, base + (size_t)M * D + k, bf, d);
                if (mm != M) st1(y, base + (size_t)mm * D + k, bf, s);
                if (k != 0) {
                    st1(y, base + (size_t)M * D + mk, bf, s);
                    if (mm != M) st1(y, base + (size_t)mm * D + mk, bf, d);
                }
            }
        }
    }
    // k = D/2: B = 0, one thread per m
    if (tid < nh1) {
        const int M = tid;
        float P = 0.0f;
        for (int f = 0; f < NF; ++f) {
            const float a1 = lds[(size_t)(2 * f) * D + Dh], a2 = lds[(size_t)(2 * f + 1) * D + Dh];
            P += tw2[(size_t)(2 * f) * 2 * FNET_TWS + M] * a1 + tw2[(size_t)(2 * f + 1) * 2 * FNET_TWS + M] * a2;
        }
        const int mm = M == 0 ? 0 : N - M;
        st1(y, base + (size_t)M * D + Dh, bf, P);
        if (mm != M) st1(y, base + (size_t)mm * D + Dh, bf, P);
    }
}

// ---------------- generic fallback (any tokens, dim): two direct-DFT kernels through an fp32 workspace
// stage 1: ws[row][k] = (A, B) = (sum_d x cos(2 pi k d/D), sum_d x sin(..)), one workgroup per row
__global__ __launch_bounds__(256) void fnet_generic_stage1(const void* __restrict__ x, float2* __restrict__ ws, int D, int bf) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xr = lds;
    float* ct = lds + D;
    float* st = ct + D;
    const size_t row = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        xr[d] = ld1(x, row * D + d, bf);
        float s, c;
        sincospif(2.0f * (float)d / (float)D, &s, &c);
        ct[d] = c;
        st[d] = s;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
        float a = 0.0f, bsum = 0.0f;
        int idx = 0;
        for (int d = 0; d < D; ++d) {
            a = fmaf(xr[d], ct[idx], a);
            bsum = fmaf(xr[d], st[idx], bsum);
            idx += k;
            if (idx >= D) idx -= D;
        }
        ws[row * D + k] = make_float2(a, bsum);
    }
}
// stage 2: y[b][m][k] = sum_n cos(2 pi m n/N) A[n][k] - sin(..) B[n][k]; grid = (ceil(D/256), B)
__global__ __launch_bounds__(256) void fnet_generic_stage2(const float2* __restrict__ ws, void* __restrict__ y, int N, int D, int bf) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ct = lds;
    float* st = lds + N;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s, c;
        sincospif(2.0f * (float)n / (float)N, &s, &c);
        ct[n] = c;
        st[n] = s;
    }
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= D) return;
    const size_t base = (size_t)blockIdx.y * N * D;
    for (int m = 0; m < N; ++m) {
        float acc = 0.0f;
        int idx = 0;
        for (int n = 0; n < N; ++n) {
            const float2 ab = ws[base + (size_t)n * D + k];
            acc = fmaf(ct[idx], ab.x, acc);
            acc = fmaf(-st[idx], ab.y, acc);
            idx += m;
            if (idx >= N) idx -= N;
        }
        st1(y, base + (size_t)m * D + k, bf, acc);
    }
}

// rfft(x).real and its adjoint; one workgroup per row
__global__ __launch_bounds__(256) void rfft_real_kernel(const void* __restrict__ x, void* __restrict__ y, int D, int transpose, int bf) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = D / 2 + 1;
    const int n_in = transpose ? K : D, n_out = transpose ? D : K;
    float* xr = lds;
    float* ct = lds + D;  // D >= K
    const size_t row = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float s, c;
        sincospif(2.0f * (float)d / (float)D, &s, &c);
        ct[d] = c;
        if (d < n_in) xr[d] = ld1(x, row * n_in + d, bf);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < n_out; o += blockDim.x) {
        float a = 0.0f;
        int idx = 0;
        for (int i = 0; i < n_in; ++i) {  // cos(2 pi o i / D): symmetric in (o, i), so the same loop serves the adjoint
            a = fmaf(xr[i], ct[idx], a);
            idx += o;
            if (idx >= D) idx -= D;
        }
        st1(y, row * n_out + o, bf, a);
    }
}

// one Haar level over the leading `len` positions of the chosen axis; positions >= len are copied
__global__ __launch_bounds__(256) void haar_level_kernel(const void* __restrict__ src, void* __restrict__ dst, int64_t total, int N,
                                                         int D, int axis_tokens, int len, int inverse, int bf) {
    const float r = 0.70710678118654752440f;
    const int lh = len >> 1, la = len - lh;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(e % D);
        const int n = (int)((e / D) % N);
        const int64_t bbase = (e / ((int64_t)D * N)) * (int64_t)D * N;
        const int p = axis_tokens ? n : d;
        auto at = [&](int q) -> float {
            return axis_tokens ? ld1(src, bbase + (int64_t)q * D + d, bf) : ld1(src, bbase + (int64_t)n * D + q, bf);
        };
        float v;
        if (p >= len) v = at(p);
        else if (!inverse) {
            if (p < lh) v = (at(2 * p) + at(2 * p + 1)) * r;
            else if (p < la) v = at(len - 1);
            else { const int q = p - la; v = (at(2 * q) - at(2 * q + 1)) * r; }
        } else {
            if ((len & 1) && p == len - 1) v = at(la - 1);
            else {
                const int q = p >> 1;
                const float a = at(q), dd = at(la + q);
                v = ((p & 1) ? a - dd : a + dd) * r;
            }
        }
        st1(dst, e, bf, v);
    }
}

}  // namespace

extern "C" int64_t spv_fnet_twiddle_floats(int tokens) { return (int64_t)(tokens + 1) * 2 * FNET_TWS; }

extern "C" int spv_fnet_make_twiddle(float* tw, int tokens, void* stream) {
    SPV_CHECK(tokens > 0, "spv_fnet_make_twiddle: tokens=%d", tokens);
    hipLaunchKernelGGL(fnet_twiddle_kernel, dim3(cdiv((tokens + 1) * 2 * FNET_TWS, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), tw, tokens);
    SPV_LAUNCH_CHECK("spv_fnet_make_twiddle");
    return 0;
}

static bool fnet_fast_ok(int tokens, int dim) {
    if (dim < 8 || (dim & (dim - 1)) || dim > 4096) return false;
    if (tokens / 2 + 1 > 2 * FNET_MH) return false;
    if (((int64_t)tokens * dim) % 4) return false;
    const size_t lds = ((size_t)(tokens + 1) * dim + 2 * (size_t)dim) * 4;
    return lds <= 160 * 1024;
}

extern "C" int64_t spv_fnet_workspace_floats(int batch, int tokens, int dim) {
    return fnet_fast_ok(tokens, dim) ? 0 : (int64_t)batch * tokens * dim * 2;
}

extern "C" int spv_fnet_mix(const void* x, void* y, const float* twiddle, int batch, int tokens, int dim, int dtype,
                            float* workspace, void* stream) {
    SPV_CHECK(batch > 0 && tokens > 0 && dim > 0, "spv_fnet_mix: empty");
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_fnet_mix: bad dtype %d", dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int bf = dtype == SPV_BF16;
    if (fnet_fast_ok(tokens, dim)) {
        SPV_CHECK(twiddle != nullptr, "spv_fnet_mix: twiddle table required");
        const size_t lds = ((size_t)(tokens + 1) * dim + 2 * (size_t)dim) * 4;
        FftPlan plan = make_fft_plan(dim);
        int log2tpf = 0;
        while ((8 << log2tpf) < dim) ++log2tpf;  // tpf = dim / 8
        const int nh1 = tokens / 2 + 1, mh = (nh1 + 1) / 2;
#define FNET_LAUNCH(MHV)                                                                                                   \
        do {                                                                                                               \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fnet_lds_kernel<MHV>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                160 * 1024);                                                                               \
            hipLaunchKernelGGL((fnet_lds_kernel<MHV>), dim3(batch), dim3(FT), lds, st, x, y, twiddle, tokens, dim, log2tpf, plan, bf); \
        } while (0)
        if (mh <= 4) FNET_LAUNCH(4);
        else if (mh <= 9) FNET_LAUNCH(9);
        else if (mh <= 13) FNET_LAUNCH(13);
        else if (mh <= 17) FNET_LAUNCH(17);
        else FNET_LAUNCH(FNET_MH);
#undef FNET_LAUNCH
        SPV_LAUNCH_CHECK("spv_fnet_mix");
        return 0;
    }
    SPV_CHECK(workspace != nullptr, "spv_fnet_mix: workspace required for tokens=%d dim=%d", tokens, dim);
    SPV_CHECK((size_t)dim * 12 <= 64 * 1024 && (size_t)tokens * 8 <= 64 * 1024, "spv_fnet_mix: shape too large");
    hipLaunchKernelGGL(fnet_generic_stage1, dim3(batch * tokens), dim3(256), (size_t)dim * 12, st, x,
                       reinterpret_cast<float2*>(workspace), dim, bf);
    SPV_LAUNCH_CHECK("spv_fnet_mix(stage1)");
    hipLaunchKernelGGL(fnet_generic_stage2, dim3(cdiv(dim, 256), batch), dim3(256), (size_t)tokens * 8, st,
                       reinterpret_cast<const float2*>(workspace), y, tokens, dim, bf);
    SPV_LAUNCH_CHECK("spv_fnet_mix(stage2)");
    return 0;
}

extern "C" int spv_rfft_real(const void* x, void* y, int rows, int dim, int transpose, int dtype, void* stream) {
    SPV_CHECK(rows > 0 && dim > 0, "spv_rfft_real: empty");
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_rfft_real: bad dtype %d", dtype);
    SPV_CHECK((size_t)dim * 8 <= 64 * 1024, "spv_rfft_real: dim too large");
    hipLaunchKernelGGL(rfft_real_kernel, dim3(rows), dim3(256), (size_t)dim * 8, static_cast<hipStream_t>(stream), x, y, dim,
                       transpose, dtype == SPV_BF16);
    SPV_LAUNCH_CHECK("spv_rfft_real");
    return 0;
}

extern "C" int spv_haar_dwt(const void* x, void* y, int batch, int tokens, int dim, int axis, int levels, int inverse, int dtype,
                            void* scratch, void* stream) {
    SPV_CHECK(batch > 0 && tokens > 0 && dim > 0, "spv_haar_dwt: empty");
    SPV_CHECK(axis == 1 || axis == 2, "spv_haar_dwt: axis must be 1 (tokens) or 2 (dim)");
    SPV_CHECK(levels >= 1 && levels <= 16, "spv_haar_dwt: levels=%d", levels);
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_haar_dwt: bad dtype %d", dtype);
    SPV_CHECK(levels == 1 || scratch != nullptr, "spv_haar_dwt: scratch needed for levels > 1");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int lens[16];
    int cur = axis == 1 ? tokens : dim;
    for (int l = 0; l < levels; ++l) { lens[l] = cur; cur = cur - cur / 2; }
    const int64_t total = (int64_t)batch * tokens * dim;
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 4096);
    const void* src = x;
    for (int i = 0; i < levels; ++i) {
        const int l = inverse ? levels - 1 - i : i;
        void* dst = ((levels - 1 - i) % 2 == 0) ? y : scratch;
        hipLaunchKernelGGL(haar_level_kernel, dim3(grid), dim3(256), 0, st, src, dst, total, tokens, dim, axis == 1, lens[l], inverse,
                           dtype == SPV_BF16);
        SPV_LAUNCH_CHECK("spv_haar_dwt");
        src = dst;
    }
    return 0;
}
