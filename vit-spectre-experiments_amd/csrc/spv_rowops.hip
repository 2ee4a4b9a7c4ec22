// spv_rowops.hip -- per-row (channel axis) kernels: the SpectreLinear tail (LayerNorm -> GELU -> +avg-pool
// skip -> dropout) and residual + LayerNorm, forward and backward.  HBM-bandwidth bound.
//
// Mapping: one wave64 per row, the whole row cached in registers (VEC contiguous elements per lane per
// step, 16 B / 8 B vector accesses for VEC = 4), wave shuffles for the two LayerNorm reductions, fp32
// statistics.  Column reductions of the backward (dgamma, dbeta, dbias) are carried in registers across
// the rows a wave owns, folded across the 4 waves of a workgroup through LDS in a fixed order, written
// as one slab per workgroup and folded by a second tiny kernel: deterministic, no atomics.
#include "spv_common.h"

namespace {

constexpr float LN_EPS = 1e-5f;
constexpr int BWD_MAX_WG = 256;

template <int VEC> __device__ __forceinline__ void ldv(const void* base, size_t off, int bf, float (&v)[VEC]);
template <> __device__ __forceinline__ void ldv<4>(const void* base, size_t off, int bf, float (&v)[4]) {
    if (bf) io<bf16_t>::ld4(static_cast<const bf16_t*>(base) + off, v);
    else io<float>::ld4(static_cast<const float*>(base) + off, v);
}
template <> __device__ __forceinline__ void ldv<1>(const void* base, size_t off, int bf, float (&v)[1]) {
    v[0] = bf ? bf2f(static_cast<const bf16_t*>(base)[off]) : static_cast<const float*>(base)[off];
}
template <int VEC> __device__ __forceinline__ void stv(void* base, size_t off, int bf, const float (&v)[VEC]);
template <> __device__ __forceinline__ void stv<4>(void* base, size_t off, int bf, const float (&v)[4]) {
    if (bf) io<bf16_t>::st4(static_cast<bf16_t*>(base) + off, v);
    else io<float>::st4(static_cast<float*>(base) + off, v);
}
template <> __device__ __forceinline__ void stv<1>(void* base, size_t off, int bf, const float (&v)[1]) {
    if (bf) static_cast<bf16_t*>(base)[off] = f2bf(v[0]);
    else static_cast<float*>(base)[off] = v[0];
}
__device__ __forceinline__ float ld1(const void* base, size_t off, int bf) {
    return bf ? bf2f(static_cast<const bf16_t*>(base)[off]) : static_cast<const float*>(base)[off];
}
__device__ __forceinline__ void st1(void* base, size_t off, int bf, float v) {
    if (bf) static_cast<bf16_t*>(base)[off] = f2bf(v);
    else static_cast<float*>(base)[off] = v;
}

// window of nn.AdaptiveAvgPool1d(n_out) output i over n_in inputs: [floor(i in/out), ceil((i+1) in/out))
__device__ __forceinline__ void pool_window(int i, int n_in, int n_out, int& s, int& e) {
    s = (int)(((long long)i * n_in) / n_out);
    e = (int)((((long long)(i + 1)) * n_in + n_out - 1) / n_out);
}

// mean / rstd of a register-cached row (two-pass, fp32)
template <int VEC, int MAXI>
__device__ __forceinline__ void row_stats(const float (&v)[MAXI][VEC], int n, int lane, float& mean, float& rstd) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
#pragma unroll
        for (int k = 0; k < VEC; ++k) s += v[i][k];  // out-of-range slots hold 0
    mean = wave_sum(s) / (float)n;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int e0 = (i * 64 + lane) * VEC;
        if (e0 < n) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) { float d = v[i][k] - mean; q += d * d; }
        }
    }
    rstd = rsqrtf(wave_sum(q) / (float)n + LN_EPS);
}

// fold per-wave column partials (NP arrays) of the 4 waves through LDS in wave order, write the slab.
template <int VEC, int MAXI, int NP>
__device__ __forceinline__ void write_partials(float (&acc)[NP][MAXI][VEC], float* lds, float* slab, int n, int lane, int wave) {
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int i = 0; i < MAXI; ++i) {
                    const int e0 = (i * 64 + lane) * VEC;
                    if (e0 < n) {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) {
                            float* q = lds + p * n + e0 + k;
                            *q = (w == 0 ? 0.0f : *q) + acc[p][i][k];
                        }
                    }
                }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < NP * n; c += blockDim.x) slab[c] = lds[c];
}

// ------------------------------------------------------------------------------------------------
template <int VEC, int MAXI>
__global__ __launch_bounds__(256) void tail_fwd_kernel(const void* __restrict__ h, const void* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       void* __restrict__ out, float* __restrict__ mean_o,
                                                       float* __restrict__ rstd_o, int rows, int n, int k_in, int bf,
                                                       int out_bf, float p_drop, uint64_t seed) {
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int row = wave_g; row < rows; row += nwaves) {
        float hv[MAXI][VEC];
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int e0 = (i * 64 + lane) * VEC;
            if (e0 < n) ldv<VEC>(h, (size_t)row * n + e0, bf, hv[i]);
            else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) hv[i][k] = 0.0f;
            }
        }
        float mean, rstd;
        row_stats<VEC, MAXI>(hv, n, lane, mean, rstd);
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int e0 = (i * 64 + lane) * VEC;
            if (e0 >= n) continue;
            float g[VEC], b[VEC], pv[VEC], o[VEC];
            ldv<VEC>(gamma, e0, 0, g);
            ldv<VEC>(beta, e0, 0, b);
            if (k_in == n) {
                ldv<VEC>(x, (size_t)row * k_in + e0, bf, pv);
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    int s, e;
                    pool_window(e0 + k, k_in, n, s, e);
                    float acc = 0.0f;
                    for (int j = s; j < e; ++j) acc += ld1(x, (size_t)row * k_in + j, bf);
                    pv[k] = acc / (float)(e - s);
                }
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float ln = (hv[i][k] - mean) * rstd * g[k] + b[k];
                o[k] = gelu_erf(ln) + pv[k];
                if (p_drop > 0.0f) o[k] *= dropout_scale(seed, (uint64_t)row * n + e0 + k, p_drop, inv_keep);
            }
            stv<VEC>(out, (size_t)row * n + e0, out_bf, o);
        }
        if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
    }
}

template <int VEC, int MAXI>
__global__ __launch_bounds__(256) void tail_bwd_kernel(const void* __restrict__ dout, const void* __restrict__ h,
                                                       const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       void* __restrict__ dh, void* __restrict__ dxp,
                                                       float* __restrict__ partials, int rows, int n, int k_in, int bf,
                                                       int dout_bf, float p_drop, uint64_t seed) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wave_g = blockIdx.x * 4 + wave;
    const int nwaves = gridDim.x * 4;
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    float acc[3][MAXI][VEC];  // 0: dgamma, 1: dbeta, 2: dbias
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[p][i][k] = 0.0f;

    for (int row = wave_g; row < rows; row += nwaves) {
        const float mean = mean_i[row], rstd = rstd_i[row];
        float xh[MAXI][VEC], dxh[MAXI][VEC];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int e0 = (i * 64 + lane) * VEC;
            if (e0 < n) {
                float hv[VEC], dv[VEC], g[VEC], b[VEC];
                ldv<VEC>(h, (size_t)row * n + e0, bf, hv);
                ldv<VEC>(dout, (size_t)row * n + e0, dout_bf, dv);
                ldv<VEC>(gamma, e0, 0, g);
                ldv<VEC>(beta, e0, 0, b);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float d = dv[k];
                    if (p_drop > 0.0f) d *= dropout_scale(seed, (uint64_t)row * n + e0 + k, p_drop, inv_keep);
                    dv[k] = d;
                    float xhat = (hv[k] - mean) * rstd;
                    float dln = d * gelu_erf_grad(xhat * g[k] + b[k]);
                    acc[0][i][k] += dln * xhat;
                    acc[1][i][k] += dln;
                    float t = dln * g[k];
                    xh[i][k] = xhat;
                    dxh[i][k] = t;
                    s1 += t;
                    s2 += t * xhat;
                }
                if (k_in == n) stv<VEC>(dxp, (size_t)row * k_in + e0, bf, dv);  // identity skip
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) { xh[i][k] = 0.0f; dxh[i][k] = 0.0f; }
            }
        }
        const float m1 = wave_sum(s1) / (float)n, m2 = wave_sum(s2) / (float)n;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int e0 = (i * 64 + lane) * VEC;
            if (e0 >= n) continue;
            float o[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                o[k] = rstd * (dxh[i][k] - m1 - xh[i][k] * m2);
                acc[2][i][k] += o[k];
            }
            stv<VEC>(dh, (size_t)row * n + e0, bf, o);
        }
        if (k_in != n) {
            // transposed pooling: input j receives dout[c] / width(c) from every window c that covers j
            for (int j = lane; j < k_in; j += 64) {
                const int c_lo = (int)(((long long)j * n) / k_in);
                const int c_hi = (int)((((long long)(j + 1)) * n + k_in - 1) / k_in) - 1;
                float a = 0.0f;
                for (int c = c_lo; c <= c_hi; ++c) {
                    int s, e;
                    pool_window(c, k_in, n, s, e);
                    float d = ld1(dout, (size_t)row * n + c, dout_bf);
                    if (p_drop > 0.0f) d *= dropout_scale(seed, (uint64_t)row * n + c, p_drop, inv_keep);
                    a += d / (float)(e - s);
                }
                st1(dxp, (size_t)row * k_in + j, bf, a);
            }
        }
    }
    write_partials<VEC, MAXI, 3>(acc, lds, partials + (size_t)blockIdx.x * 3 * n, n, lane, wave);
}

template <int VEC, int MAXI>
__global__ __launch_bounds__(256) void addln_fwd_kernel(const void* __restrict__ a, const void* __restrict__ b,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        void* __restrict__ out, float* __restrict__ mean_o,
                                                        float* __restrict__ rstd_o, int rows, int n, int mode, int bf) {
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    for (int row = wave_g; row < rows; row += nwaves) {
        float v[MAXI][VEC], res[MAXI][VEC];
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int e0 = (i * 64 + lane) * VEC;
            if (e0 < n) {
                ldv<VEC>(a, (size_t)row * n + e0, bf, v[i]);
                ldv<VEC>(b, (size_t)row * n + e0, bf, res[i]);
                if (mode == 1) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) v[i][k] += res[i][k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) { v[i][k] = 0.0f; res[i][k] = 0.0f; }
            }
        }
        float mean, rstd;
        row_stats<VEC, MAXI>(v, n, lane, mean, rstd);
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int e0 = (i * 64 + lane) * VEC;
            if (e0 >= n) continue;
            float g[VEC], be[VEC], o[VEC];
            ldv<VEC>(gamma, e0, 0, g);
            ldv<VEC>(beta, e0, 0, be);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                o[k] = (v[i][k] - mean) * rstd * g[k] + be[k];
                if (mode == 0) o[k] += res[i][k];
            }
            stv<VEC>(out, (size_t)row * n + e0, bf, o);
        }
        if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
    }
}

template <int VEC, int MAXI>
__global__ __launch_bounds__(256) void addln_bwd_kernel(const void* __restrict__ dout, const void* __restrict__ a,
                                                        const void* __restrict__ b, const float* __restrict__ mean_i,
                                                        const float* __restrict__ rstd_i, const float* __restrict__ gamma,
                                                        void* __restrict__ din, float* __restrict__ partials, int rows, int n,
                                                        int mode, int bf) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wave_g = blockIdx.x * 4 + wave;
    const int nwaves = gridDim.x * 4;
    float acc[2][MAXI][VEC];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[p][i][k] = 0.0f;
    for (int row = wave_g; row < rows; row += nwaves) {
        const float mean = mean_i[row], rstd = rstd_i[row];
        float xh[MAXI][VEC], dxh[MAXI][VEC];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int e0 = (i * 64 + lane) * VEC;
            if (e0 < n) {
                float av[VEC], dv[VEC], g[VEC];
                ldv<VEC>(a, (size_t)row * n + e0, bf, av);
                if (mode == 1) {
                    float bv[VEC];
                    ldv<VEC>(b, (size_t)row * n + e0, bf, bv);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) av[k] += bv[k];
                }
                ldv<VEC>(dout, (size_t)row * n + e0, bf, dv);
                ldv<VEC>(gamma, e0, 0, g);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float xhat = (av[k] - mean) * rstd;
                    acc[0][i][k] += dv[k] * xhat;
                    acc[1][i][k] += dv[k];
                    float t = dv[k] * g[k];
                    xh[i][k] = xhat;
                    dxh[i][k] = t;
                    s1 += t;
                    s2 += t * xhat;
                }
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) { xh[i][k] = 0.0f; dxh[i][k] = 0.0f; }
            }
        }
        const float m1 = wave_sum(s1) / (float)n, m2 = wave_sum(s2) / (float)n;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int e0 = (i * 64 + lane) * VEC;
            if (e0 >= n) continue;
            float o[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) o[k] = rstd * (dxh[i][k] - m1 - xh[i][k] * m2);
            stv<VEC>(din, (size_t)row * n + e0, bf, o);
        }
    }
    write_partials<VEC, MAXI, 2>(acc, lds, partials + (size_t)blockIdx.x * 2 * n, n, lane, wave);
}

// out[p][c] = sum_w partials[w][p][c]
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ partials, float* __restrict__ o0,
                                                            float* __restrict__ o1, float* __restrict__ o2, int parts, int np,
                                                            int n) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= np * n) return;
    float s = 0.0f;
    for (int w = 0; w < parts; ++w) s += partials[(size_t)w * np * n + c];
    const int p = c / n, cc = c % n;
    float* o = p == 0 ? o0 : (p == 1 ? o1 : o2);
    if (o) o[cc] = s;
}

struct RowCfg { int vec, maxi; };
inline bool pick_cfg(int n, RowCfg& c) {
    if (n % 4 == 0) {
        const int ni = cdiv(n, 256);
        c.vec = 4;
        if (ni <= 2) c.maxi = 2; else if (ni <= 3) c.maxi = 3; else if (ni <= 4) c.maxi = 4;
        else if (ni <= 12) c.maxi = 12; else if (ni <= 16) c.maxi = 16; else return false;
        return true;
    }
    if (n <= 1024) { c.vec = 1; c.maxi = 16; return true; }
    return false;
}

#define ROW_DISPATCH(cfg, KERNEL, grid, lds_bytes, st, ...)                                                    \
    do {                                                                                                       \
        if (cfg.vec == 4 && cfg.maxi == 2) hipLaunchKernelGGL((KERNEL<4, 2>), grid, dim3(256), lds_bytes, st, __VA_ARGS__);        \
        else if (cfg.vec == 4 && cfg.maxi == 3) hipLaunchKernelGGL((KERNEL<4, 3>), grid, dim3(256), lds_bytes, st, __VA_ARGS__);   \
        else if (cfg.vec == 4 && cfg.maxi == 4) hipLaunchKernelGGL((KERNEL<4, 4>), grid, dim3(256), lds_bytes, st, __VA_ARGS__);   \
        else if (cfg.vec == 4 && cfg.maxi == 12) hipLaunchKernelGGL((KERNEL<4, 12>), grid, dim3(256), lds_bytes, st, __VA_ARGS__); \
        else if (cfg.vec == 4 && cfg.maxi == 16) hipLaunchKernelGGL((KERNEL<4, 16>), grid, dim3(256), lds_bytes, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<1, 16>), grid, dim3(256), lds_bytes, st, __VA_ARGS__);                  \
    } while (0)

inline int check_dtype(int d) { return d == SPV_F32 || d == SPV_BF16; }

}  // namespace

extern "C" int64_t spv_rowop_partial_floats(int n) { return (int64_t)BWD_MAX_WG * 3 * n; }

extern "C" int spv_spectre_tail_fwd(const void* h, const void* x, const float* gamma, const float* beta, void* out,
                                    float* mean, float* rstd, int rows, int n, int k_in, int dtype, int out_dtype,
                                    float p_drop, uint64_t seed, void* stream) {
    SPV_CHECK(rows > 0 && n > 0 && k_in > 0, "spv_spectre_tail_fwd: empty");
    SPV_CHECK(check_dtype(dtype) && check_dtype(out_dtype), "spv_spectre_tail_fwd: bad dtype");
    SPV_CHECK(p_drop >= 0.0f && p_drop < 1.0f, "spv_spectre_tail_fwd: p_drop=%f", p_drop);
    RowCfg cfg;
    SPV_CHECK(pick_cfg(n, cfg), "spv_spectre_tail_fwd: unsupported row length %d", n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(std::min(cdiv(rows, 4), 2048));
    ROW_DISPATCH(cfg, tail_fwd_kernel, grid, 0, st, h, x, gamma, beta, out, mean, rstd, rows, n, k_in,
                 dtype == SPV_BF16, out_dtype == SPV_BF16, p_drop, seed);
    SPV_LAUNCH_CHECK("spv_spectre_tail_fwd");
    return 0;
}

extern "C" int spv_spectre_tail_bwd(const void* dout, const void* h, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, void* dh, void* dx_pool, float* dgamma,
                                    float* dbeta, float* dbias, float* partials, int rows, int n, int k_in, int dtype,
                                    int dout_dtype, float p_drop, uint64_t seed, void* stream) {
    SPV_CHECK(rows > 0 && n > 0 && k_in > 0, "spv_spectre_tail_bwd: empty");
    SPV_CHECK(check_dtype(dtype) && check_dtype(dout_dtype), "spv_spectre_tail_bwd: bad dtype");
    RowCfg cfg;
    SPV_CHECK(pick_cfg(n, cfg), "spv_spectre_tail_bwd: unsupported row length %d", n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int wgs = std::min(cdiv(rows, 4), BWD_MAX_WG);
    ROW_DISPATCH(cfg, tail_bwd_kernel, dim3(wgs), (size_t)3 * n * sizeof(float), st, dout, h, mean, rstd, gamma, beta, dh,
                 dx_pool, partials, rows, n, k_in, dtype == SPV_BF16, dout_dtype == SPV_BF16, p_drop, seed);
    SPV_LAUNCH_CHECK("spv_spectre_tail_bwd");
    hipLaunchKernelGGL(fold_partials_kernel, dim3(cdiv(3 * n, 256)), dim3(256), 0, st, partials, dgamma, dbeta, dbias, wgs, 3, n);
    SPV_LAUNCH_CHECK("spv_spectre_tail_bwd(fold)");
    return 0;
}

extern "C" int spv_add_layernorm_fwd(const void* a, const void* b, const float* gamma, const float* beta, void* out,
                                     float* mean, float* rstd, int rows, int n, int mode, int dtype, void* stream) {
    SPV_CHECK(rows > 0 && n > 0, "spv_add_layernorm_fwd: empty");
    SPV_CHECK(check_dtype(dtype), "spv_add_layernorm_fwd: bad dtype");
    SPV_CHECK(mode == 0 || mode == 1, "spv_add_layernorm_fwd: mode=%d", mode);
    RowCfg cfg;
    SPV_CHECK(pick_cfg(n, cfg), "spv_add_layernorm_fwd: unsupported row length %d", n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(std::min(cdiv(rows, 4), 2048));
    ROW_DISPATCH(cfg, addln_fwd_kernel, grid, 0, st, a, b, gamma, beta, out, mean, rstd, rows, n, mode, dtype == SPV_BF16);
    SPV_LAUNCH_CHECK("spv_add_layernorm_fwd");
    return 0;
}

extern "C" int spv_add_layernorm_bwd(const void* dout, const void* a, const void* b, const float* mean, const float* rstd,
                                     const float* gamma, void* din, float* dgamma, float* dbeta, float* partials, int rows,
                                     int n, int mode, int dtype, void* stream) {
    SPV_CHECK(rows > 0 && n > 0, "spv_add_layernorm_bwd: empty");
    SPV_CHECK(check_dtype(dtype), "spv_add_layernorm_bwd: bad dtype");
    SPV_CHECK(mode == 0 || mode == 1, "spv_add_layernorm_bwd: mode=%d", mode);
    RowCfg cfg;
    SPV_CHECK(pick_cfg(n, cfg), "spv_add_layernorm_bwd: unsupported row length %d", n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int wgs = std::min(cdiv(rows, 4), BWD_MAX_WG);
    ROW_DISPATCH(cfg, addln_bwd_kernel, dim3(wgs), (size_t)2 * n * sizeof(float), st, dout, a, b, mean, rstd, gamma, din,
                 partials, rows, n, mode, dtype == SPV_BF16);
    SPV_LAUNCH_CHECK("spv_add_layernorm_bwd");
    hipLaunchKernelGGL(fold_partials_kernel, dim3(cdiv(2 * n, 256)), dim3(256), 0, st, partials, dgamma, dbeta, (float*)nullptr, wgs, 2, n);
    SPV_LAUNCH_CHECK("spv_add_layernorm_bwd(fold)");
    return 0;
}
