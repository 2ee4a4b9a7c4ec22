// spv_rowops.hip -- per-row (channel axis) kernels: the SpectreLinear tail (LayerNorm -> GELU -> +avg-pool
// skip -> dropout) and residual + LayerNorm, forward and backward.  HBM-bandwidth bound.
//
// Mapping: one wave64 per row, the whole row cached in registers (VEC contiguous elements per lane per
// step, 16 B / 8 B vector accesses for VEC = 4), DPP reductions for the two LayerNorm sums, fp32
// statistics.  Column reductions of the backward (dgamma, dbeta, dbias) are carried across the rows a wave
// owns (registers in the generic kernels, per-wave LDS slots in the lane-contiguous ones), folded across
// the 4 waves of a workgroup through LDS in a fixed order, written as one slab per workgroup and folded by
// a second small kernel: deterministic, no atomics.
#define SPV_USES_SEED
#include "spv_common.h"

namespace {

constexpr float LN_EPS = 1e-5f;
constexpr int RT = 256;           // threads per workgroup (4 waves; 3-4 workgroups per CU keep 12-16 rows in flight)
constexpr int RW = RT / 64;       // waves (= rows in flight) per workgroup
constexpr int BWD_MAX_WG = 1024;  // backward grids are capped here: one partial slab per workgroup

// how the avg-pool skip maps channels: identity, exact windows of k_in/n inputs, or the general
// overlapping/up-sampling windows (tables built once per workgroup in LDS, no per-element division)
enum PoolMode { POOL_IDENT = 0, POOL_EXACT = 1, POOL_TABLE = 2, POOL_REPEAT = 3 };  // REPEAT: n = r * k_in, out[c] = x[c / r]

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
// Same-wave LDS hand-off (one lane writes, another lane of the SAME wave reads): the LDS pipe is in order per wave,
// so only the compiler must be kept from reordering; waiting on lgkmcnt alone leaves global loads/stores in flight
// (a workgroup-scope fence would also drain vmcnt, i.e. wait for the previous row's stores every row).
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ float ld1(const void* base, size_t off, int bf) {
    return bf ? bf2f(static_cast<const bf16_t*>(base)[off]) : static_cast<const float*>(base)[off];
}
__device__ __forceinline__ void st1(void* base, size_t off, int bf, float v) {
    if (bf) static_cast<bf16_t*>(base)[off] = f2bf(v);
    else static_cast<float*>(base)[off] = v;
}

// window of nn.AdaptiveAvgPool1d(n_out) output i over n_in inputs: [floor(i in/out), ceil((i+1) in/out))
// (32-bit arithmetic: the host checks n_in * n_out < 2^31)
__device__ __forceinline__ void pool_window(int i, int n_in, int n_out, int& s, int& e) {
    s = (int)(((unsigned)i * (unsigned)n_in) / (unsigned)n_out);
    e = (int)((((unsigned)(i + 1)) * (unsigned)n_in + (unsigned)n_out - 1u) / (unsigned)n_out);
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

// fold per-wave column partials (NP arrays) of the workgroup's waves through LDS in wave order, write the slab.
template <int VEC, int MAXI, int NP>
__device__ __forceinline__ void write_partials(float (&acc)[NP][MAXI][VEC], float* lds, float* slab, int n, int lane, int wave) {
    for (int w = 0; w < RW; ++w) {
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
template <int VEC, int MAXI, bool FASTG>
__global__ __launch_bounds__(RT, (MAXI <= 4 ? 4 : (MAXI <= 12 ? 2 : 1))) void tail_fwd_kernel(const void* __restrict__ h, const void* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       void* __restrict__ out, float* __restrict__ mean_o,
                                                       float* __restrict__ rstd_o, int rows, int n, int k_in, int bf,
                                                       int out_bf, float p_drop, uint64_t seed, int pool_mode) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int pw = k_in / n;  // POOL_EXACT window
    const float inv_pw = 1.0f / (float)(pw > 0 ? pw : 1);
    const int lane = threadIdx.x & 63;
    // POOL_TABLE: per output channel, window start | count << 16 and 1 / count (LDS tables: register copies cost occupancy)
    int* win_s = reinterpret_cast<int*>(lds);
    float* win_inv = lds + n;
    if (pool_mode == POOL_TABLE) {
        for (int c = threadIdx.x; c < n; c += RT) {
            int s, e;
            pool_window(c, k_in, n, s, e);
            win_s[c] = s | ((e - s) << 16);
            win_inv[c] = 1.0f / (float)(e - s);
        }
        __syncthreads();
    }
    float* xs = lds + 2 * n + (threadIdx.x >> 6) * k_in;  // POOL_TABLE: this wave's fp32 copy of the input row
    const bool exact4 = pool_mode == POOL_EXACT && (pw & 3) == 0 && VEC == 4;
    float* xq = lds + (threadIdx.x >> 6) * (k_in >> 2);   // POOL_EXACT (pw % 4 == 0): per-wave sums of 4 inputs
    const int wave_g = blockIdx.x * RW + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * RW;
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint64_t seed_l = live_seed(seed);  // hoisted: the graph-mode seed word is read once, not once per row
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
        const unsigned rkey = p_drop > 0.0f ? dropout_row_key(seed_l, (uint64_t)row) : 0u;
        if (exact4) {
            // coalesced pass over the input row: lane sums 4 consecutive inputs; the owner of an output adds pw/4 of them
            lds_fence();
            for (int qd = lane; qd < (k_in >> 2); qd += 64) {
                float t4[4];
                ldv<4>(x, (size_t)row * k_in + 4 * qd, bf, t4);
                xq[qd] = (t4[0] + t4[1]) + (t4[2] + t4[3]);
            }
            lds_fence();
        }
        if (pool_mode == POOL_TABLE) {
            lds_fence();  // previous row's pooled reads are done before the stage is overwritten
            if ((k_in & 3) == 0) {
                for (int j = lane * 4; j < k_in; j += 256) {
                    float t4[4];
                    ldv<4>(x, (size_t)row * k_in + j, bf, t4);
                    *reinterpret_cast<float4*>(xs + j) = make_float4(t4[0], t4[1], t4[2], t4[3]);
                }
            } else {
                for (int j = lane; j < k_in; j += 64) xs[j] = ld1(x, (size_t)row * k_in + j, bf);
            }
            lds_fence();  // same-wave LDS traffic is in order; the fence keeps the compiler from reordering
        }
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int e0 = (i * 64 + lane) * VEC;
            if (e0 >= n) continue;
            float g[VEC], b[VEC], pv[VEC], o[VEC];
            ldv<VEC>(gamma, e0, 0, g);
            ldv<VEC>(beta, e0, 0, b);
            if (pool_mode == POOL_IDENT) {
                ldv<VEC>(x, (size_t)row * k_in + e0, bf, pv);
            } else if (pool_mode == POOL_REPEAT) {
                const int rr = n / k_in;
#pragma unroll
                for (int k = 0; k < VEC; ++k) pv[k] = ld1(x, (size_t)row * k_in + (e0 + k) / rr, bf);
            } else if (pool_mode == POOL_EXACT) {
                // this lane's VEC outputs average VEC*pw contiguous inputs
                const size_t xb = (size_t)row * k_in + (size_t)e0 * pw;
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float acc = 0.0f;
                    if (exact4) {
                        const int q0 = (e0 + k) * (pw >> 2);
                        for (int j = 0; j < (pw >> 2); ++j) acc += xq[q0 + j];
                    } else {
                        for (int j = 0; j < pw; ++j) acc += ld1(x, xb + (size_t)k * pw + j, bf);
                    }
                    pv[k] = acc * inv_pw;
                }
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const int sc = win_s[e0 + k];
                    const int s = sc & 0xffff, cnt = sc >> 16;
                    float acc = xs[s];
                    if (cnt > 1) acc += xs[s + 1];
                    if (cnt > 2) acc += xs[s + 2];
                    for (int j = 3; j < cnt; ++j) acc += xs[s + j];
                    pv[k] = acc * win_inv[e0 + k];
                }
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float ln = (hv[i][k] - mean) * rstd * g[k] + b[k];
                float act, dact_unused;
                if (FASTG) gelu_fast(ln, act, dact_unused);
                else act = gelu_erf(ln);
                o[k] = act + pv[k];
                if (p_drop > 0.0f) o[k] *= dropout_scale(rkey, (unsigned)(e0 + k), p_drop, inv_keep);
            }
            stv<VEC>(out, (size_t)row * n + e0, out_bf, o);
        }
        if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
    }
}

template <int VEC, int MAXI, bool FASTG>
__global__ __launch_bounds__(RT, (MAXI <= 4 ? 4 : (MAXI <= 12 ? 2 : 1))) void tail_bwd_kernel(const void* __restrict__ dout, const void* __restrict__ h,
                                                       const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       void* __restrict__ dh, void* __restrict__ dxp,
                                                       float* __restrict__ partials, int rows, int n, int k_in, int bf,
                                                       int dout_bf, float p_drop, uint64_t seed, int pool_mode,
                                                       const void* __restrict__ dx_add) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // LDS: [3n] partial fold | POOL_TABLE: c_lo[k_in], c_hi[k_in], inv_w[n]
    int* c_lo_t = reinterpret_cast<int*>(lds + 3 * n);
    int* c_hi_t = c_lo_t + k_in;
    float* inv_w_t = reinterpret_cast<float*>(c_hi_t + k_in);
    if (pool_mode == POOL_TABLE) {
        for (int j = threadIdx.x; j < k_in; j += RT) {
            c_lo_t[j] = (int)(((unsigned)j * (unsigned)n) / (unsigned)k_in);
            c_hi_t[j] = (int)((((unsigned)(j + 1)) * (unsigned)n + (unsigned)k_in - 1u) / (unsigned)k_in) - 1;
        }
        for (int c = threadIdx.x; c < n; c += RT) {
            int s, e;
            pool_window(c, k_in, n, s, e);
            inv_w_t[c] = 1.0f / (float)(e - s);
        }
        __syncthreads();
    }
    const int pw = k_in / n;
    const float inv_pw = 1.0f / (float)(pw > 0 ? pw : 1);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* ds = inv_w_t + n + wave * n;  // POOL_TABLE: this wave's fp32 copy of the masked dout row
    const int wave_g = blockIdx.x * RW + wave;
    const int nwaves = gridDim.x * RW;
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint64_t seed_l = live_seed(seed);  // hoisted: the graph-mode seed word is read once, not once per row
    float acc[3][MAXI][VEC];  // 0: dgamma, 1: dbeta, 2: dbias
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[p][i][k] = 0.0f;

    for (int row = wave_g; row < rows; row += nwaves) {
        const float mean = mean_i[row], rstd = rstd_i[row];
        const unsigned rkey = p_drop > 0.0f ? dropout_row_key(seed_l, (uint64_t)row) : 0u;
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
                    if (p_drop > 0.0f) d *= dropout_scale(rkey, (unsigned)(e0 + k), p_drop, inv_keep);
                    dv[k] = d;
                    float xhat = (hv[k] - mean) * rstd;
                    float dgel, act_unused;
                    if (FASTG) gelu_fast(xhat * g[k] + b[k], act_unused, dgel);
                    else dgel = gelu_erf_grad(xhat * g[k] + b[k]);
                    float dln = d * dgel;
                    acc[0][i][k] += dln * xhat;
                    acc[1][i][k] += dln;
                    float t = dln * g[k];
                    xh[i][k] = xhat;
                    dxh[i][k] = t;
                    s1 += t;
                    s2 += t * xhat;
                }
                if (pool_mode == POOL_IDENT) {  // identity skip
                    if (dx_add != nullptr) {
                        float ad[VEC];
                        ldv<VEC>(dx_add, (size_t)row * k_in + e0, bf, ad);
                        float t[VEC];
#pragma unroll
                        for (int k = 0; k < VEC; ++k) t[k] = dv[k] + ad[k];
                        stv<VEC>(dxp, (size_t)row * k_in + e0, bf, t);
                    } else {
                        stv<VEC>(dxp, (size_t)row * k_in + e0, bf, dv);
                    }
                }
                else if (pool_mode == POOL_TABLE) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) ds[e0 + k] = dv[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) { xh[i][k] = 0.0f; dxh[i][k] = 0.0f; }
            }
        }
        const float m1 = wave_sum(s1) / (float)n, m2 = wave_sum(s2) / (float)n;
        if (pool_mode == POOL_TABLE) lds_fence();
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
        if (pool_mode == POOL_EXACT && dxp != nullptr) {  // dxp == nullptr: the data-gradient GEMM adds this term in its epilogue
            // transposed pooling, exact windows: input j receives dout[j / pw] / pw
            if (VEC == 4 && (pw & 3) == 0) {
                for (int j0 = lane * 4; j0 < k_in; j0 += 256) {
                    const int c = j0 / pw;
                    float d = ld1(dout, (size_t)row * n + c, dout_bf);
                    if (p_drop > 0.0f) d *= dropout_scale(rkey, (unsigned)c, p_drop, inv_keep);
                    d *= inv_pw;
                    float o[VEC];
#pragma unroll
                    for (int u = 0; u < VEC; ++u) o[u] = d;
                    if (dx_add != nullptr) {
                        float ad[VEC];
                        ldv<VEC>(dx_add, (size_t)row * k_in + j0, bf, ad);
#pragma unroll
                        for (int u = 0; u < VEC; ++u) o[u] += ad[u];
                    }
                    stv<VEC>(dxp, (size_t)row * k_in + j0, bf, o);
                }
            } else {
                for (int j = lane; j < k_in; j += 64) {
                    const int c = j / pw;
                    float d = ld1(dout, (size_t)row * n + c, dout_bf);
                    if (p_drop > 0.0f) d *= dropout_scale(rkey, (unsigned)c, p_drop, inv_keep);
                    st1(dxp, (size_t)row * k_in + j, bf, d * inv_pw + (dx_add ? ld1(dx_add, (size_t)row * k_in + j, bf) : 0.0f));
                }
            }
        } else if (pool_mode == POOL_REPEAT) {
            // every input j fed n / k_in outputs with weight 1
            const int rr = n / k_in;
            for (int j = lane; j < k_in; j += 64) {
                float a = 0.0f;
                for (int c = j * rr; c < (j + 1) * rr; ++c) {
                    float d = ld1(dout, (size_t)row * n + c, dout_bf);
                    if (p_drop > 0.0f) d *= dropout_scale(rkey, (unsigned)c, p_drop, inv_keep);
                    a += d;
                }
                if (dx_add != nullptr) a += ld1(dx_add, (size_t)row * k_in + j, bf);
                st1(dxp, (size_t)row * k_in + j, bf, a);
            }
        } else if (pool_mode == POOL_TABLE) {
            // transposed pooling: input j receives dout[c] / width(c) from every window c that covers j
            for (int j = lane; j < k_in; j += 64) {
                const int c_lo = c_lo_t[j], c_hi = c_hi_t[j];
                float a = 0.0f;
                for (int c = c_lo; c <= c_hi; ++c) a += ds[c] * inv_w_t[c];
                if (dx_add != nullptr) a += ld1(dx_add, (size_t)row * k_in + j, bf);
                st1(dxp, (size_t)row * k_in + j, bf, a);
            }
            lds_fence();  // pooled reads done before the next row overwrites the stage
        }
    }
    write_partials<VEC, MAXI, 3>(acc, lds, partials + (size_t)blockIdx.x * 3 * n, n, lane, wave);
}

template <int VEC, int MAXI>
__global__ __launch_bounds__(RT) void addln_fwd_kernel(const void* __restrict__ a, const void* __restrict__ b,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        void* __restrict__ out, float* __restrict__ mean_o,
                                                        float* __restrict__ rstd_o, int rows, int n, int mode, int bf) {
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * RW + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * RW;
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
__global__ __launch_bounds__(RT) void addln_bwd_kernel(const void* __restrict__ dout, const void* __restrict__ a,
                                                        const void* __restrict__ b, const float* __restrict__ mean_i,
                                                        const float* __restrict__ rstd_i, const float* __restrict__ gamma,
                                                        void* __restrict__ din, float* __restrict__ partials, int rows, int n,
                                                        int mode, int bf) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wave_g = blockIdx.x * RW + wave;
    const int nwaves = gridDim.x * RW;
    float acc[2][MAXI][VEC];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[p][i][k] = 0.0f;
    // the rows a wave owns are a serial chain (load -> two wave reductions -> store) with loop-carried column sums, so
    // the next row's loads are issued before the current row is reduced: two rows in flight per wave
    float av[MAXI][VEC], dv[MAXI][VEC], mean = 0.0f, rstd = 0.0f;
    auto fetch = [&](int row, float (&fa)[MAXI][VEC], float (&fd)[MAXI][VEC], float& fm, float& fr) {
        fm = mean_i[row];
        fr = rstd_i[row];
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int e0 = (i * 64 + lane) * VEC;
            if (e0 < n) {
                ldv<VEC>(a, (size_t)row * n + e0, bf, fa[i]);
                if (mode == 1) {
                    float bv[VEC];
                    ldv<VEC>(b, (size_t)row * n + e0, bf, bv);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) fa[i][k] += bv[k];
                }
                ldv<VEC>(dout, (size_t)row * n + e0, bf, fd[i]);
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) { fa[i][k] = 0.0f; fd[i][k] = 0.0f; }
            }
        }
    };
    if (wave_g < rows) fetch(wave_g, av, dv, mean, rstd);
    for (int row = wave_g; row < rows; row += nwaves) {
        float nav[MAXI][VEC], ndv[MAXI][VEC], nmean = 0.0f, nrstd = 0.0f;
        const int nrow = row + nwaves;
        if (nrow < rows) fetch(nrow, nav, ndv, nmean, nrstd);
        float xh[MAXI][VEC], dxh[MAXI][VEC];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int e0 = (i * 64 + lane) * VEC;
            if (e0 < n) {
                float g[VEC];
                ldv<VEC>(gamma, e0, 0, g);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float xhat = (av[i][k] - mean) * rstd;
                    acc[0][i][k] += dv[i][k] * xhat;
                    acc[1][i][k] += dv[i][k];
                    float t = dv[i][k] * g[k];
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
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
#pragma unroll
            for (int k = 0; k < VEC; ++k) { av[i][k] = nav[i][k]; dv[i][k] = ndv[i][k]; }
        mean = nmean;
        rstd = nrstd;
    }
    write_partials<VEC, MAXI, 2>(acc, lds, partials + (size_t)blockIdx.x * 2 * n, n, lane, wave);
}

// The same for rows of exactly MAXI * 256 elements (the encoder widths 512 / 768 / 1024), dtype and mode compile-time facts, and the
// NEXT row's loads kept in flight RAW: the generic kernel converts inside the branch that loaded (run-time dtype flag), which puts an
// s_waitcnt behind every load -- 37 us for the 136 MB of a Small layer (0.35 of HBM peak) where the tails reach 0.6-0.7.
template <int MAXI, bool BF, int MODE>
__global__ __launch_bounds__(RT) void addln_bwd_fast_kernel(const void* __restrict__ dout, const void* __restrict__ a,
                                                             const void* __restrict__ b, const float* __restrict__ mean_i,
                                                             const float* __restrict__ rstd_i, const float* __restrict__ gamma,
                                                             void* __restrict__ din, float* __restrict__ partials, int rows) {
    constexpr int VEC = 4, n = MAXI * 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wave_g = blockIdx.x * RW + wave;
    const int nwaves = gridDim.x * RW;
    float acc[2][MAXI][VEC];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[p][i][k] = 0.0f;
    struct RawRow { uint4 a[MAXI], b[MAXI], d[MAXI]; float m, r; };
    auto ld1 = [&](const void* p, size_t idx) -> uint4 {
        if constexpr (BF) {
            const uint2 t = *reinterpret_cast<const uint2*>(static_cast<const bf16_t*>(p) + idx);
            return make_uint4(t.x, t.y, 0u, 0u);
        } else {
            return *reinterpret_cast<const uint4*>(static_cast<const float*>(p) + idx);
        }
    };
    auto un = [&](const uint4& r, float (&o)[4]) {
        if constexpr (BF) {
            o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
            o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
        } else {
            o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
        }
    };
    auto fetch = [&](int row, RawRow& f) {
        f.m = mean_i[row];
        f.r = rstd_i[row];
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const size_t idx = (size_t)row * n + (i * 64 + lane) * VEC;
            f.a[i] = ld1(a, idx);
            if constexpr (MODE == 1) f.b[i] = ld1(b, idx);
            f.d[i] = ld1(dout, idx);
        }
    };
    float g[MAXI][VEC];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const float4 t = *reinterpret_cast<const float4*>(gamma + (i * 64 + lane) * VEC);
        g[i][0] = t.x; g[i][1] = t.y; g[i][2] = t.z; g[i][3] = t.w;
    }
    RawRow cur;
    if (wave_g < rows) fetch(wave_g, cur);
    for (int row = wave_g; row < rows; row += nwaves) {
        RawRow nxt = cur;
        const int nrow = row + nwaves;
        if (nrow < rows) fetch(nrow, nxt);
        const float mean = cur.m, rstd = cur.r;
        float xh[MAXI][VEC], dxh[MAXI][VEC];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            float av[4], dv[4];
            un(cur.a[i], av);
            un(cur.d[i], dv);
            if constexpr (MODE == 1) {
                float bv[4];
                un(cur.b[i], bv);
#pragma unroll
                for (int k = 0; k < VEC; ++k) av[k] += bv[k];
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float xhat = (av[k] - mean) * rstd;
                acc[0][i][k] += dv[k] * xhat;
                acc[1][i][k] += dv[k];
                const float t = dv[k] * g[i][k];
                xh[i][k] = xhat;
                dxh[i][k] = t;
                s1 += t;
                s2 += t * xhat;
            }
        }
        const float m1 = wave_sum(s1) * (1.0f / (float)n), m2 = wave_sum(s2) * (1.0f / (float)n);
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            float o[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) o[k] = rstd * (dxh[i][k] - m1 - xh[i][k] * m2);
            stv<VEC>(din, (size_t)row * n + (i * 64 + lane) * VEC, BF, o);
        }
        cur = nxt;
    }
    write_partials<VEC, MAXI, 2>(acc, lds, partials + (size_t)blockIdx.x * 2 * n, n, lane, wave);
}

// ------------------------------------------------------------------------------------------------
// One-level Haar DWT along the embedding axis + LayerNorm-1 + residual as ONE row kernel each way (bf16, D = 512 KD):
//   out = LN(haar(x)) * gamma + beta + x      (SpectreEncoderLayer with the 'dwt_embed' mixer: norm1(mix(x)) + x, spectre.py:66)
// Lane l owns input elements [8c, 8c + 8) of chunk c = 64 j + l, i.e. four pairs, hence the band values a at 4c .. 4c + 3 and d at
// D/2 + 4c .. -- the transform is lane-local both ways, LayerNorm's statistics do not care about the order, and the column
// layout (i * 64 + lane) * 4 of write_partials is exactly this ownership (i = j for the averages, KD + j for the differences).
// Nothing of the mixer is stored: the backward recomputes haar(x) from x.  Replaces haar (12 us) + add+LayerNorm (16 us) forward and
// add+LayerNorm backward (32 us) + inverse haar (12 us) per layer.
template <int KD>
__global__ __launch_bounds__(RT) void haar_ln_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, bf16_t* __restrict__ out,
                                                          float* __restrict__ mean_o, float* __restrict__ rstd_o, int rows) {
    constexpr int D = 512 * KD, H = D / 2;
    const float r = 0.70710678118654752440f;
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * RW + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * RW;
    for (int row = wave_g; row < rows; row += nwaves) {
        const bf16_t* xr = x + (size_t)row * D;
        uint4 raw[KD];
        uint2 xa[KD], xd[KD];
#pragma unroll
        for (int j = 0; j < KD; ++j) {
            const int c = 64 * j + lane;
            raw[j] = *reinterpret_cast<const uint4*>(xr + 8 * c);
            xa[j] = *reinterpret_cast<const uint2*>(xr + 4 * c);          // the residual at the positions this lane writes
            xd[j] = *reinterpret_cast<const uint2*>(xr + H + 4 * c);
        }
        float av[KD][4], dv[KD][4], s = 0.0f;
#pragma unroll
        for (int j = 0; j < KD; ++j) {
            const unsigned w[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float x0 = __uint_as_float(w[u] << 16), x1 = __uint_as_float(w[u] & 0xffff0000u);
                // the unfused path stores the band values as bf16 before LayerNorm reads them: keep its rounding
                av[j][u] = bf2f(f2bf((x0 + x1) * r));
                dv[j][u] = bf2f(f2bf((x0 - x1) * r));
                s += av[j][u] + dv[j][u];
            }
        }
        const float mu = wave_sum(s) * (1.0f / (float)D);
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < KD; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float e0 = av[j][u] - mu, e1 = dv[j][u] - mu;
                q += e0 * e0 + e1 * e1;
            }
        const float rs = rsqrtf(wave_sum(q) * (1.0f / (float)D) + 1e-5f);
        if (lane == 0) { mean_o[row] = mu; rstd_o[row] = rs; }
        bf16_t* orow = out + (size_t)row * D;
#pragma unroll
        for (int j = 0; j < KD; ++j) {
            const int c = 64 * j + lane;
            const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * c), gd = *reinterpret_cast<const float4*>(gamma + H + 4 * c);
            const float4 ba = *reinterpret_cast<const float4*>(beta + 4 * c), bd = *reinterpret_cast<const float4*>(beta + H + 4 * c);
            const float g0[4] = {ga.x, ga.y, ga.z, ga.w}, g1[4] = {gd.x, gd.y, gd.z, gd.w};
            const float b0[4] = {ba.x, ba.y, ba.z, ba.w}, b1[4] = {bd.x, bd.y, bd.z, bd.w};
            const float ra[4] = {__uint_as_float(xa[j].x << 16), __uint_as_float(xa[j].x & 0xffff0000u), __uint_as_float(xa[j].y << 16),
                                 __uint_as_float(xa[j].y & 0xffff0000u)};
            const float rd[4] = {__uint_as_float(xd[j].x << 16), __uint_as_float(xd[j].x & 0xffff0000u), __uint_as_float(xd[j].y << 16),
                                 __uint_as_float(xd[j].y & 0xffff0000u)};
            float oa[4], od[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                oa[u] = (av[j][u] - mu) * rs * g0[u] + b0[u] + ra[u];
                od[u] = (dv[j][u] - mu) * rs * g1[u] + b1[u] + rd[u];
            }
            uint2 wa, wd;
            wa.x = pack_bf16x2(oa[0], oa[1]); wa.y = pack_bf16x2(oa[2], oa[3]);
            wd.x = pack_bf16x2(od[0], od[1]); wd.y = pack_bf16x2(od[2], od[3]);
            *reinterpret_cast<uint2*>(orow + 4 * c) = wa;
            *reinterpret_cast<uint2*>(orow + H + 4 * c) = wd;
        }
    }
}

template <int KD>
__global__ __launch_bounds__(RT) void haar_ln_bwd_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x,
                                                          const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                          const float* __restrict__ gamma, bf16_t* __restrict__ dx,
                                                          float* __restrict__ partials, int rows) {
    constexpr int D = 512 * KD, H = D / 2, MAXI = 2 * KD;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float r = 0.70710678118654752440f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_g = blockIdx.x * RW + wave;
    const int nwaves = gridDim.x * RW;
    float acc[2][MAXI][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[p][i][k] = 0.0f;
    float g0[KD][4], g1[KD][4];
#pragma unroll
    for (int j = 0; j < KD; ++j) {
        const int c = 64 * j + lane;
        const float4 ga = *reinterpret_cast<const float4*>(gamma + 4 * c), gd = *reinterpret_cast<const float4*>(gamma + H + 4 * c);
        g0[j][0] = ga.x; g0[j][1] = ga.y; g0[j][2] = ga.z; g0[j][3] = ga.w;
        g1[j][0] = gd.x; g1[j][1] = gd.y; g1[j][2] = gd.z; g1[j][3] = gd.w;
    }
    for (int row = wave_g; row < rows; row += nwaves) {
        const bf16_t* xr = x + (size_t)row * D;
        const bf16_t* dr = dout + (size_t)row * D;
        uint4 raw[KD], res[KD];
        uint2 da[KD], dd[KD];
#pragma unroll
        for (int j = 0; j < KD; ++j) {
            const int c = 64 * j + lane;
            raw[j] = *reinterpret_cast<const uint4*>(xr + 8 * c);
            res[j] = *reinterpret_cast<const uint4*>(dr + 8 * c);          // the residual path's gradient for the lane's 8 inputs
            da[j] = *reinterpret_cast<const uint2*>(dr + 4 * c);           // dout at the band positions this lane owns
            dd[j] = *reinterpret_cast<const uint2*>(dr + H + 4 * c);
        }
        const float mu = mean_i[row], rs = rstd_i[row];
        float xh0[KD][4], xh1[KD][4], t0[KD][4], t1[KD][4], s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < KD; ++j) {
            const unsigned w[4] = {raw[j].x, raw[j].y, raw[j].z, raw[j].w};
            const float d0[4] = {__uint_as_float(da[j].x << 16), __uint_as_float(da[j].x & 0xffff0000u), __uint_as_float(da[j].y << 16),
                                 __uint_as_float(da[j].y & 0xffff0000u)};
            const float d1[4] = {__uint_as_float(dd[j].x << 16), __uint_as_float(dd[j].x & 0xffff0000u), __uint_as_float(dd[j].y << 16),
                                 __uint_as_float(dd[j].y & 0xffff0000u)};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float x0 = __uint_as_float(w[u] << 16), x1 = __uint_as_float(w[u] & 0xffff0000u);
                const float a = bf2f(f2bf((x0 + x1) * r)), d = bf2f(f2bf((x0 - x1) * r));
                xh0[j][u] = (a - mu) * rs;
                xh1[j][u] = (d - mu) * rs;
                acc[0][j][u] += d0[u] * xh0[j][u];
                acc[1][j][u] += d0[u];
                acc[0][KD + j][u] += d1[u] * xh1[j][u];
                acc[1][KD + j][u] += d1[u];
                t0[j][u] = d0[u] * g0[j][u];
                t1[j][u] = d1[u] * g1[j][u];
                s1 += t0[j][u] + t1[j][u];
                s2 += t0[j][u] * xh0[j][u] + t1[j][u] * xh1[j][u];
            }
        }
        const float m1 = wave_sum(s1) * (1.0f / (float)D), m2 = wave_sum(s2) * (1.0f / (float)D);
        bf16_t* orow = dx + (size_t)row * D;
#pragma unroll
        for (int j = 0; j < KD; ++j) {
            const int c = 64 * j + lane;
            const unsigned wr[4] = {res[j].x, res[j].y, res[j].z, res[j].w};
            float o[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // the unfused path stores d(haar output) as bf16 before the inverse transform reads it: keep its rounding
                const float ga = bf2f(f2bf(rs * (t0[j][u] - m1 - xh0[j][u] * m2)));
                const float gd = bf2f(f2bf(rs * (t1[j][u] - m1 - xh1[j][u] * m2)));
                o[2 * u] = (ga + gd) * r + __uint_as_float(wr[u] << 16);
                o[2 * u + 1] = (ga - gd) * r + __uint_as_float(wr[u] & 0xffff0000u);
            }
            uint4 t;
            t.x = pack_bf16x2(o[0], o[1]); t.y = pack_bf16x2(o[2], o[3]); t.z = pack_bf16x2(o[4], o[5]); t.w = pack_bf16x2(o[6], o[7]);
            *reinterpret_cast<uint4*>(orow + 8 * c) = t;
        }
    }
    write_partials<4, MAXI, 2>(acc, lds, partials + (size_t)blockIdx.x * 2 * D, D, lane, wave);
}

// out[p][c] = sum_w partials[w][p][c]; a workgroup of 1024 threads owns 16 columns (64-byte segments), 64 thread rows
// split the slabs (<= 16 loads each at 1024 slabs: the kernel is latency bound, so the loads must be spread thin),
// fixed summation order
__global__ __launch_bounds__(FOLD_COLS * FOLD_ROWS) void fold_partials_kernel(const float* __restrict__ partials, float* __restrict__ o0,
                                                                             float* __restrict__ o1, float* __restrict__ o2,
                                                                             int parts, int np, int n, float* __restrict__ o3 = nullptr,
                                                                             float* __restrict__ o4 = nullptr) {
    const FoldJob j{partials, {o0, o1, o2, o3, o4}, parts, np, n};
    fold_partials_block(j, blockIdx.x, threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Lane-contiguous variants of the SpectreLinear tail for channel counts that are multiples of 64 with a small
// rational pooling ratio (Small: 512 -> 768 and 768 -> 512).  Lane t owns outputs [t*CO, (t+1)*CO) and, because the
// adaptive windows [floor(c*k/n), ceil((c+1)*k/n)) never cross a multiple of k/64, exactly the inputs
// [t*CI, (t+1)*CI): the avg-pool skip and its transpose are lane-local with COMPILE-TIME windows -- no LDS stage, no
// tables, no divergent window loops (the general path spends ~24 us per call on them).
constexpr int lc_ws(int c, int CO, int CI) { return (c * CI) / CO; }
constexpr int lc_we(int c, int CO, int CI) { return ((c + 1) * CI + CO - 1) / CO; }

template <int CNT> __device__ __forceinline__ void ld_span(const void* base, size_t off, int bf, float (&v)[CNT]) {
#pragma unroll
    for (int q = 0; q < CNT / 4; ++q) {
        float t[4];
        ldv<4>(base, off + 4 * q, bf, t);
#pragma unroll
        for (int u = 0; u < 4; ++u) v[4 * q + u] = t[u];
    }
}
template <int CNT> __device__ __forceinline__ void st_span(void* base, size_t off, int bf, const float (&v)[CNT]) {
#pragma unroll
    for (int q = 0; q < CNT / 4; ++q) {
        float t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = v[4 * q + u];
        stv<4>(base, off + 4 * q, bf, t);
    }
}

// A lane's span as it lies in memory (packed bf16 pairs, or floats): loaded at the TOP of a row iteration, unpacked where it is
// used -- so every load of the row is in flight at once, in half the registers floats would take.
template <int CNT, int BF> struct RawSpan;
template <int CNT> struct RawSpan<CNT, 1> { unsigned w[CNT / 2]; };
template <int CNT> struct RawSpan<CNT, 0> { float w[CNT]; };
template <int CNT, int BF> __device__ __forceinline__ void ld_raw(const void* base, size_t off, RawSpan<CNT, BF>& r) {
    if constexpr (BF) {
        const unsigned* p = reinterpret_cast<const unsigned*>(static_cast<const bf16_t*>(base) + off);
#pragma unroll
        for (int i = 0; i < CNT / 2; ++i) r.w[i] = p[i];
    } else {
        const float* p = static_cast<const float*>(base) + off;
#pragma unroll
        for (int i = 0; i < CNT; ++i) r.w[i] = p[i];
    }
}
template <int CNT, int BF> __device__ __forceinline__ void unpack(const RawSpan<CNT, BF>& r, float (&v)[CNT]) {
    if constexpr (BF) {
#pragma unroll
        for (int i = 0; i < CNT / 2; ++i) {
            v[2 * i] = __uint_as_float(r.w[i] << 16);
            v[2 * i + 1] = __uint_as_float(r.w[i] & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int i = 0; i < CNT; ++i) v[i] = r.w[i];
    }
}

// LayerNorm-2 of the encoder layer riding on the linear3 tail (reference spectre.py:67: norm2(x1 + ff(x1))): the tail's
// output row f3 is complete in one wave, so x2 = LN2(x1 + f3) follows in registers (forward), and in the backward the
// LayerNorm-2 gradient ds is formed in registers and used at once as the tail's incoming gradient.  Saves the
// add+LayerNorm kernels and one read of f3 (forward) / ds (backward).  f3 is still written: the backward recomputes x1 + f3
// from the same bf16 tensors the unfused kernels read, so both paths give the same numbers.
struct TailLn2 {
    const void* res;        // x1: the residual added before LayerNorm-2
    const float* gamma2;
    const float* beta2;
    void* out2;             // forward: x2
    float* mean2;           // forward writes, backward reads
    float* rstd2;
    const void* dout2;      // backward: gradient wrt x2
    const void* f3;         // backward: the tail's forward output
    void* ds;               // backward: gradient wrt (x1 + f3), also needed by the residual branch
};
__device__ __forceinline__ float round_store(float v, int bf) { return bf ? bf2f(f2bf(v)) : v; }

template <int CO, int CI, bool FASTG, bool LN2>
__global__ __launch_bounds__(RT, 4) void tail_fwd_lc_kernel(const void* __restrict__ h, const void* __restrict__ x,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             void* __restrict__ out, float* __restrict__ mean_o,
                                                             float* __restrict__ rstd_o, int rows, int bf_rt, int out_bf_rt, float p_drop,
                                                             uint64_t seed, TailLn2 ln) {
    // the storage dtype is a COMPILE-TIME fact here (FASTG <=> bf16, and the host takes this kernel only when the output dtype
    // matches): as run-time flags every 8-byte load sat in its own branch arm with its own s_waitcnt vmcnt(0) -- ten serialized
    // memory round trips per row in the backward kernel, where one batch of loads is issued now
    constexpr int bf = FASTG ? 1 : 0, out_bf = bf;
    (void)bf_rt; (void)out_bf_rt;
    constexpr int n = 64 * CO, k_in = 64 * CI;
    constexpr float inv_n = 1.0f / (float)n;   // a multiply: the division by 768 (not a power of two) was ten instructions, four times per row
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * RW + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * RW;
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint64_t seed_l = live_seed(seed);  // hoisted: the graph-mode seed word is read once, not once per row
    float g[CO], b[CO];
    ld_span<CO>(gamma, (size_t)lane * CO, 0, g);
    ld_span<CO>(beta, (size_t)lane * CO, 0, b);
    for (int row = wave_g; row < rows; row += nwaves) {
        RawSpan<CO, bf> r_h, r_res;
        RawSpan<CI, bf> r_x;
        ld_raw<CO, bf>(h, (size_t)row * n + lane * CO, r_h);
        ld_raw<CI, bf>(x, (size_t)row * k_in + lane * CI, r_x);
        if (LN2) ld_raw<CO, bf>(ln.res, (size_t)row * n + lane * CO, r_res);  // the residual of LayerNorm-2: asked for with the rest
        float hv[CO], xin[CI];
        unpack<CO, bf>(r_h, hv);
        unpack<CI, bf>(r_x, xin);
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < CO; ++c) s += hv[c];
        const float mean = wave_sum(s) * inv_n;
        float q = 0.0f;
#pragma unroll
        for (int c = 0; c < CO; ++c) { const float d = hv[c] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) * inv_n + LN_EPS);
        const unsigned rkey = p_drop > 0.0f ? dropout_row_key(seed_l, (uint64_t)row) : 0u;
        const unsigned lane_gold = (unsigned)(lane * (CO / 2)) * 0x9e3779b1u;  // this lane's first column pair in the mask hash
        float o[CO];
#pragma unroll
        for (int c = 0; c < CO; c += 2) {  // column pairs: packed fp32 math
            float pv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < CI; ++j)
                    if (j >= lc_ws(c + u, CO, CI) && j < lc_we(c + u, CO, CI)) acc += xin[j];
                pv[u] = acc * (1.0f / (float)(lc_we(c + u, CO, CI) - lc_ws(c + u, CO, CI)));
            }
            const f32x2 hv2 = {hv[c], hv[c + 1]}, g2 = {g[c], g[c + 1]}, b2 = {b[c], b[c + 1]};
            const f32x2 ln = (hv2 - mean) * (g2 * rstd) + b2;
            f32x2 act, unused;
            if (FASTG) gelu_fast2(ln, act, unused);
            else act = f32x2{gelu_erf(ln.x), gelu_erf(ln.y)};
            o[c] = act.x + pv[0];
            o[c + 1] = act.y + pv[1];
            if (p_drop > 0.0f) {
                o[c] *= dropout_scale_at(rkey + lane_gold, c, p_drop, inv_keep);
                o[c + 1] *= dropout_scale_at(rkey + lane_gold, c + 1, p_drop, inv_keep);
            }
        }
        st_span<CO>(out, (size_t)row * n + lane * CO, out_bf, o);
        if (lane == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
        if (LN2) {
            float rv[CO], g2[CO], b2[CO];
            unpack<CO, bf>(r_res, rv);
            float sm = 0.0f;
#pragma unroll
            for (int c = 0; c < CO; ++c) { rv[c] += round_store(o[c], out_bf); sm += rv[c]; }  // x1 + f3 as the stored f3 reads back
            const float mean2 = wave_sum(sm) * inv_n;
            float q2 = 0.0f;
#pragma unroll
            for (int c = 0; c < CO; ++c) { const float d = rv[c] - mean2; q2 += d * d; }
            const float rstd2 = rsqrtf(wave_sum(q2) * inv_n + LN_EPS);
            ld_span<CO>(ln.gamma2, (size_t)lane * CO, 0, g2);
            ld_span<CO>(ln.beta2, (size_t)lane * CO, 0, b2);
#pragma unroll
            for (int c = 0; c < CO; ++c) rv[c] = (rv[c] - mean2) * rstd2 * g2[c] + b2[c];
            st_span<CO>(ln.out2, (size_t)row * n + lane * CO, bf, rv);
            if (lane == 0) { ln.mean2[row] = mean2; ln.rstd2[row] = rstd2; }
        }
    }
}

// The skip gradient of the layer ABOVE, taken at its source (tail_bwd_lc<12, 8> only: the layer above is the <8, 12> one, whose
// 12 pooled inputs per lane are exactly this kernel's 12 outputs per lane).  That layer's backward then writes no dx_pool and
// its data-gradient GEMM stores instead of accumulating: one [rows, 768] tensor is neither written nor re-read (102 MB per
// encoder layer), and the GEMM's epilogue loses its read-modify-write.  src = that layer's incoming gradient before its
// dropout mask ([rows, 512]: `ds` of spv_spectre_tail_ln_bwd), p_drop / seed = that layer's dropout.
struct TailUp {
    const void* src;
    float p_drop;
    uint64_t seed;
};

// Column sums (dgamma, dbeta, dbias) live in LDS, not registers: 3*CO accumulators per lane pushed the register count
// past the 4-waves/SIMD budget (spills).  Each wave owns [3][CO/4][64] float4 slots, lane-linear (conflict-free b128).
template <int CO, int CI, bool FASTG, bool LN2>
__global__ __launch_bounds__(RT, 4) void tail_bwd_lc_kernel(const void* __restrict__ dout, const void* __restrict__ h,
                                                             const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             void* __restrict__ dh, void* __restrict__ dxp, float* __restrict__ partials,
                                                             int rows, int bf_rt, int dout_bf_rt, float p_drop, uint64_t seed,
                                                             const void* __restrict__ dx_add, TailLn2 ln, TailUp up) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int bf = FASTG ? 1 : 0;  // compile-time dtype (dout has it too): see tail_fwd_lc_kernel
    (void)bf_rt; (void)dout_bf_rt;
    constexpr int n = 64 * CO, k_in = 64 * CI, Q = CO / 4;
    constexpr float inv_n = 1.0f / (float)n;
    constexpr int NP = LN2 ? 5 : 3;  // column-sum arrays: dgamma, dbeta, dbias (+ LayerNorm-2's dgamma, dbeta)
    static_assert(CO % 4 == 0, "lane span must be whole float4s");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_g = blockIdx.x * RW + wave;
    const int nwaves = gridDim.x * RW;
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    float4* accw = reinterpret_cast<float4*>(lds) + (size_t)wave * NP * Q * 64 + lane;  // slot (p, q) at accw[(p*Q+q)*64]
#pragma unroll
    for (int i = 0; i < NP * Q; ++i) accw[i * 64] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint64_t seed_l = live_seed(seed), up_seed_l = live_seed(up.seed);  // (one read of the graph-mode seed word, not one per row)
    for (int row = wave_g; row < rows; row += nwaves) {
        // every load of the row goes out here, before the first use (they were issued -- and waited for -- one after the other)
        RawSpan<CO, bf> r_h, r_d, r_s, r_t;
        RawSpan<CI, bf> r_u, r_a;
        ld_raw<CO, bf>(LN2 ? ln.dout2 : dout, (size_t)row * n + lane * CO, r_d);
        ld_raw<CO, bf>(h, (size_t)row * n + lane * CO, r_h);
        if (LN2) {
            ld_raw<CO, bf>(ln.f3, (size_t)row * n + lane * CO, r_s);
            ld_raw<CO, bf>(ln.res, (size_t)row * n + lane * CO, r_t);
        }
        constexpr bool WIDE_IN = CI > 16;   // many inputs per lane (3072 -> 768 at the Base width): the skip gradient goes out in chunks
        if constexpr (!WIDE_IN) {
            if (!LN2 && up.src != nullptr) ld_raw<CI, bf>(up.src, (size_t)row * k_in + lane * CI, r_u);
            if (dxp != nullptr && dx_add != nullptr) ld_raw<CI, bf>(dx_add, (size_t)row * k_in + lane * CI, r_a);
        }
        const float mean = mean_i[row], rstd = rstd_i[row];
        const unsigned rkey = p_drop > 0.0f ? dropout_row_key(seed_l, (uint64_t)row) : 0u;
        const unsigned lane_gold = (unsigned)(lane * (CO / 2)) * 0x9e3779b1u;  // this lane's first column pair in the mask hash
        float hv[CO], dv[CO], dxh[CO];
        unpack<CO, bf>(r_d, dv);
        if (LN2) {
            // incoming gradient = LayerNorm-2 backward of dout2 at s = x1 + f3, formed here instead of by a separate kernel
            float sv[CO], tv[CO];
            unpack<CO, bf>(r_s, sv);
            unpack<CO, bf>(r_t, tv);
            const float mean2 = ln.mean2[row], rstd2 = ln.rstd2[row];
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                float g2[4];
                ld_span<4>(ln.gamma2, (size_t)lane * CO + 4 * q, 0, g2);
                float4 v3 = accw[(3 * Q + q) * 64], v4 = accw[(4 * Q + q) * 64];
                float a3[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = 4 * q + e;
                    const float xh = (sv[c] + tv[c] - mean2) * rstd2;
                    a3[e] = dv[c] * xh;
                    sv[c] = xh;               // xhat2
                    tv[c] = dv[c] * g2[e];    // dy g
                    t1 += tv[c];
                    t2 += tv[c] * xh;
                }
                v3.x += a3[0]; v3.y += a3[1]; v3.z += a3[2]; v3.w += a3[3];
                v4.x += dv[4 * q]; v4.y += dv[4 * q + 1]; v4.z += dv[4 * q + 2]; v4.w += dv[4 * q + 3];
                accw[(3 * Q + q) * 64] = v3;
                accw[(4 * Q + q) * 64] = v4;
            }
            const float n1 = wave_sum(t1) * inv_n, n2 = wave_sum(t2) * inv_n;
#pragma unroll
            for (int c = 0; c < CO; ++c) dv[c] = round_store(rstd2 * (tv[c] - n1 - sv[c] * n2), bf);
            st_span<CO>(ln.ds, (size_t)row * n + lane * CO, bf, dv);
        } else {
            if (up.src != nullptr) {
                // + the transposed pooling of the masked gradient of the layer above (its CI outputs per lane from its CO...
                // seen from that layer: CI of its values, pooled onto CO of its inputs = this lane's CO columns)
                float u[CI];
                unpack<CI, bf>(r_u, u);
                if (up.p_drop > 0.0f) {
                    const unsigned ukey = dropout_row_key(up_seed_l, (uint64_t)row);
                    const float uinv = 1.0f / (1.0f - up.p_drop);
#pragma unroll
                    for (int c = 0; c < CI; ++c) u[c] *= dropout_scale_at(ukey + (unsigned)(lane * (CI / 2)) * 0x9e3779b1u, c, up.p_drop, uinv);
                }
#pragma unroll
                for (int j = 0; j < CO; ++j) {
                    float a = 0.0f;
#pragma unroll
                    for (int c = 0; c < CI; ++c)
                        if (j >= lc_ws(c, CI, CO) && j < lc_we(c, CI, CO)) a += u[c] * (1.0f / (float)(lc_we(c, CI, CO) - lc_ws(c, CI, CO)));
                    dv[j] += a;
                }
            }
        }
        unpack<CO, bf>(r_h, hv);
        f32x2 s1v = {0.0f, 0.0f}, s2v = {0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float g[4], b[4], a0[4], a1[4];
            ld_span<4>(gamma, (size_t)lane * CO + 4 * q, 0, g);
            ld_span<4>(beta, (size_t)lane * CO + 4 * q, 0, b);
            // column PAIRS on packed fp32 math (v_pk_*_f32): PMC showed this kernel VALU bound -- 569 vector instructions per row, the
            // four waves of a SIMD asking for 120 % of its issue slots -- once its loads were batched
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const int c = 4 * q + e;
                if (p_drop > 0.0f) {
                    dv[c] *= dropout_scale_at(rkey + lane_gold, c, p_drop, inv_keep);
                    dv[c + 1] *= dropout_scale_at(rkey + lane_gold, c + 1, p_drop, inv_keep);
                }
                const f32x2 hv2 = {hv[c], hv[c + 1]}, g2 = {g[e], g[e + 1]}, b2 = {b[e], b[e + 1]}, dv2 = {dv[c], dv[c + 1]};
                const f32x2 xhat = (hv2 - mean) * rstd;
                f32x2 dgel, unused;
                if (FASTG) gelu_fast2(xhat * g2 + b2, unused, dgel);
                else {
                    const f32x2 pre = xhat * g2 + b2;
                    dgel = f32x2{gelu_erf_grad(pre.x), gelu_erf_grad(pre.y)};
                }
                const f32x2 dln = dv2 * dgel;
                const f32x2 ax = dln * xhat;
                const f32x2 t = dln * g2;
                s1v += t;
                s2v += t * xhat;
                a0[e] = ax.x; a0[e + 1] = ax.y;
                a1[e] = dln.x; a1[e + 1] = dln.y;
                dxh[c] = t.x; dxh[c + 1] = t.y;
            }
            float4 v0 = accw[(0 * Q + q) * 64], v1 = accw[(1 * Q + q) * 64];
            v0.x += a0[0]; v0.y += a0[1]; v0.z += a0[2]; v0.w += a0[3];
            v1.x += a1[0]; v1.y += a1[1]; v1.z += a1[2]; v1.w += a1[3];
            accw[(0 * Q + q) * 64] = v0;
            accw[(1 * Q + q) * 64] = v1;
            __builtin_amdgcn_sched_barrier(0);  // keep the chunks sequential: interleaving them only adds live registers
        }
        const float m1 = wave_sum(s1v.x + s1v.y) * inv_n, m2 = wave_sum(s2v.x + s2v.y) * inv_n;
        float o[CO];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float4 v2 = accw[(2 * Q + q) * 64];
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const int c = 4 * q + e;
                const f32x2 hv2 = {hv[c], hv[c + 1]}, t2 = {dxh[c], dxh[c + 1]};
                const f32x2 o2 = (t2 - m1 - (hv2 - mean) * (rstd * m2)) * rstd;
                o[c] = o2.x;
                o[c + 1] = o2.y;
            }
            v2.x += o[4 * q]; v2.y += o[4 * q + 1]; v2.z += o[4 * q + 2]; v2.w += o[4 * q + 3];
            accw[(2 * Q + q) * 64] = v2;
        }
        st_span<CO>(dh, (size_t)row * n + lane * CO, bf, o);
        if (dxp == nullptr) continue;  // the consumer takes the skip gradient from this layer's incoming gradient (TailUp)
        // transposed pooling, lane local: input j receives dout[c] / width(c) from every window that covers it
        if constexpr (WIDE_IN) {
            // eight inputs at a time, stored as they are formed (a 48-element array and its residual's raw copy spilled 130 VGPRs)
#pragma unroll
            for (int j0 = 0; j0 < CI; j0 += 8) {
                float dxc[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = j0 + jj;
                    float a = 0.0f;
#pragma unroll
                    for (int c = 0; c < CO; ++c)
                        if (j >= lc_ws(c, CO, CI) && j < lc_we(c, CO, CI)) a += dv[c] * (1.0f / (float)(lc_we(c, CO, CI) - lc_ws(c, CO, CI)));
                    dxc[jj] = a;
                }
                if (dx_add != nullptr) {
                    RawSpan<8, bf> ra;
                    ld_raw<8, bf>(dx_add, (size_t)row * k_in + lane * CI + j0, ra);
                    float ad[8];
                    unpack<8, bf>(ra, ad);
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) dxc[jj] += ad[jj];
                }
                st_span<8>(dxp, (size_t)row * k_in + lane * CI + j0, bf, dxc);
            }
            continue;
        }
        float dx[CI];
#pragma unroll
        for (int j = 0; j < CI; ++j) {
            float a = 0.0f;
#pragma unroll
            for (int c = 0; c < CO; ++c)
                if (j >= lc_ws(c, CO, CI) && j < lc_we(c, CO, CI)) a += dv[c] * (1.0f / (float)(lc_we(c, CO, CI) - lc_ws(c, CO, CI)));
            dx[j] = a;
        }
        if (dx_add != nullptr) {  // residual gradient folded in here instead of a separate elementwise add
            float ad[CI];
            unpack<CI, bf>(r_a, ad);
#pragma unroll
            for (int j = 0; j < CI; ++j) dx[j] += ad[j];
        }
        st_span<CI>(dxp, (size_t)row * k_in + lane * CI, bf, dx);
    }
    __syncthreads();
    // sum the 4 waves' slots in wave order, one slab per workgroup: slab[p*n + col], col = lane*CO + 4q + e
    float* slab = partials + (size_t)blockIdx.x * NP * n;
    for (int i = threadIdx.x; i < NP * n; i += blockDim.x) {
        const int p = i / n, col = i - p * n;
        const int l = col / CO, r = col - l * CO;
        const int off = (((p * Q + (r >> 2)) * 64 + l) << 2) + (r & 3);
        float v = lds[off];
#pragma unroll
        for (int w = 1; w < RW; ++w) v += lds[w * NP * n + off];
        slab[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// SpectreLinear tail for WIDE rows whose skip repeats every input WR = 4 times (768 -> 3072: linear1 at the Base width): the four waves
// of a workgroup share ONE row -- wave w owns columns [768 w, 768 w + 768) = 12 outputs and 3 inputs per lane, windows lane-local --
// and join their LayerNorm sums through LDS.  (The generic kernel holds 48 fp32 outputs per lane, branches per 2-byte skip load and
// divides per element: 543 us forward / 541 us backward per Base layer at bs 64, 15x its bytes' worth.)  Column sums need no LDS: a
// wave sees the same columns in every row, so they live in 36 registers per lane until the slab is written.
constexpr int WCO = 12, WR = 4, WCI = WCO / WR;
template <int BF, bool FASTG>
__global__ __launch_bounds__(RT) void tail_wide_fwd_kernel(const void* __restrict__ h, const void* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, void* __restrict__ out, float* __restrict__ mean_o,
                                                           float* __restrict__ rstd_o, int rows, float p_drop, uint64_t seed) {
    constexpr int n = RW * 64 * WCO, k_in = n / WR;
    constexpr float inv_n = 1.0f / (float)n;
    __shared__ float red[2][2][RW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col0 = (wave * 64 + lane) * WCO, in0 = (wave * 64 + lane) * WCI;
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint64_t seed_l = live_seed(seed);
    float g[WCO], b[WCO];
    ld_span<WCO>(gamma, col0, 0, g);
    ld_span<WCO>(beta, col0, 0, b);
    int par = 0;
    for (int row = blockIdx.x; row < rows; row += gridDim.x, par ^= 1) {
        RawSpan<WCO, BF> r_h;
        ld_raw<WCO, BF>(h, (size_t)row * n + col0, r_h);
        float xv[WCI];
#pragma unroll
        for (int j = 0; j < WCI; ++j) xv[j] = BF ? bf2f(static_cast<const bf16_t*>(x)[(size_t)row * k_in + in0 + j]) : static_cast<const float*>(x)[(size_t)row * k_in + in0 + j];
        float hv[WCO];
        unpack<WCO, BF>(r_h, hv);
        float sm = 0.0f;
#pragma unroll
        for (int c = 0; c < WCO; ++c) sm += hv[c];
        sm = wave_sum(sm);
        if (lane == 0) red[par][0][wave] = sm;
        __syncthreads();
        float tot = 0.0f;
#pragma unroll
        for (int w = 0; w < RW; ++w) tot += red[par][0][w];
        const float mean = tot * inv_n;
        float q = 0.0f;
#pragma unroll
        for (int c = 0; c < WCO; ++c) { const float d = hv[c] - mean; q += d * d; }
        q = wave_sum(q);
        if (lane == 0) red[par][1][wave] = q;
        __syncthreads();
        float qt = 0.0f;
#pragma unroll
        for (int w = 0; w < RW; ++w) qt += red[par][1][w];
        const float rstd = rsqrtf(qt * inv_n + LN_EPS);
        const unsigned rkey = p_drop > 0.0f ? dropout_row_key(seed_l, (uint64_t)row) : 0u;
        float o[WCO];
#pragma unroll
        for (int c = 0; c < WCO; ++c) {
            const float ln = (hv[c] - mean) * rstd * g[c] + b[c];
            float act, unused;
            if (FASTG) gelu_fast(ln, act, unused);
            else act = gelu_erf(ln);
            o[c] = act + xv[c / WR];
            if (p_drop > 0.0f) o[c] *= dropout_scale(rkey, (unsigned)(col0 + c), p_drop, inv_keep);
        }
        st_span<WCO>(out, (size_t)row * n + col0, BF, o);
        if (threadIdx.x == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
    }
}

template <int BF, bool FASTG>
__global__ __launch_bounds__(RT) void tail_wide_bwd_kernel(const void* __restrict__ dout, const void* __restrict__ h, const float* __restrict__ mean_i,
                                                           const float* __restrict__ rstd_i, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, void* __restrict__ dh, void* __restrict__ dxp,
                                                           float* __restrict__ partials, int rows, float p_drop, uint64_t seed,
                                                           const void* __restrict__ dx_add) {
    constexpr int n = RW * 64 * WCO, k_in = n / WR;
    constexpr float inv_n = 1.0f / (float)n;
    __shared__ float red[2][2][RW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col0 = (wave * 64 + lane) * WCO, in0 = (wave * 64 + lane) * WCI;
    const float inv_keep = p_drop > 0.0f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const uint64_t seed_l = live_seed(seed);
    float g[WCO], b[WCO], acc[3][WCO];
    ld_span<WCO>(gamma, col0, 0, g);
    ld_span<WCO>(beta, col0, 0, b);
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int c = 0; c < WCO; ++c) acc[p][c] = 0.0f;
    int par = 0;
    for (int row = blockIdx.x; row < rows; row += gridDim.x, par ^= 1) {
        RawSpan<WCO, BF> r_h, r_d;
        ld_raw<WCO, BF>(dout, (size_t)row * n + col0, r_d);
        ld_raw<WCO, BF>(h, (size_t)row * n + col0, r_h);
        float ad[WCI];
#pragma unroll
        for (int j = 0; j < WCI; ++j) ad[j] = 0.0f;
        if (dxp != nullptr && dx_add != nullptr) {
#pragma unroll
            for (int j = 0; j < WCI; ++j)
                ad[j] = BF ? bf2f(static_cast<const bf16_t*>(dx_add)[(size_t)row * k_in + in0 + j]) : static_cast<const float*>(dx_add)[(size_t)row * k_in + in0 + j];
        }
        const float mean = mean_i[row], rstd = rstd_i[row];
        const unsigned rkey = p_drop > 0.0f ? dropout_row_key(seed_l, (uint64_t)row) : 0u;
        float hv[WCO], dv[WCO], xh[WCO], t[WCO];
        unpack<WCO, BF>(r_h, hv);
        unpack<WCO, BF>(r_d, dv);
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int c = 0; c < WCO; ++c) {
            if (p_drop > 0.0f) dv[c] *= dropout_scale(rkey, (unsigned)(col0 + c), p_drop, inv_keep);
            xh[c] = (hv[c] - mean) * rstd;
            float dgel, unused;
            if (FASTG) gelu_fast(xh[c] * g[c] + b[c], unused, dgel);
            else dgel = gelu_erf_grad(xh[c] * g[c] + b[c]);
            const float dln = dv[c] * dgel;
            acc[0][c] += dln * xh[c];
            acc[1][c] += dln;
            t[c] = dln * g[c];
            s1 += t[c];
            s2 += t[c] * xh[c];
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) { red[par][0][wave] = s1; red[par][1][wave] = s2; }
        __syncthreads();
        float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
        for (int w = 0; w < RW; ++w) { m1 += red[par][0][w]; m2 += red[par][1][w]; }
        m1 *= inv_n;
        m2 *= inv_n;
        float o[WCO];
#pragma unroll
        for (int c = 0; c < WCO; ++c) {
            o[c] = rstd * (t[c] - m1 - xh[c] * m2);
            acc[2][c] += o[c];
        }
        st_span<WCO>(dh, (size_t)row * n + col0, BF, o);
        if (dxp != nullptr) {   // transposed repeat: an input collects the (masked) gradients of its WR outputs
#pragma unroll
            for (int j = 0; j < WCI; ++j) {
                float a = ad[j];
#pragma unroll
                for (int r = 0; r < WR; ++r) a += dv[j * WR + r];
                if (BF) static_cast<bf16_t*>(dxp)[(size_t)row * k_in + in0 + j] = f2bf(a);
                else static_cast<float*>(dxp)[(size_t)row * k_in + in0 + j] = a;
            }
        }
    }
    float* slab = partials + (size_t)blockIdx.x * 3 * n;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int c = 0; c < WCO; c += 4)
            *reinterpret_cast<float4*>(slab + (size_t)p * n + col0 + c) = make_float4(acc[p][c], acc[p][c + 1], acc[p][c + 2], acc[p][c + 3]);
}

#define TAIL_DISPATCH(cfg, fast, KERNEL, grid, lds_bytes, st, ...)                                              \
    do {                                                                                                       \
        if (fast) {                                                                                            \
            if (cfg.vec == 4 && cfg.maxi == 2) hipLaunchKernelGGL((KERNEL<4, 2, true>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__);        \
            else if (cfg.vec == 4 && cfg.maxi == 3) hipLaunchKernelGGL((KERNEL<4, 3, true>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__);   \
            else if (cfg.vec == 4 && cfg.maxi == 4) hipLaunchKernelGGL((KERNEL<4, 4, true>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__);   \
            else if (cfg.vec == 4 && cfg.maxi == 12) hipLaunchKernelGGL((KERNEL<4, 12, true>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__); \
            else if (cfg.vec == 4 && cfg.maxi == 16) hipLaunchKernelGGL((KERNEL<4, 16, true>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<1, 16, true>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__);         \
        } else {                                                                                               \
            if (cfg.vec == 4 && cfg.maxi == 2) hipLaunchKernelGGL((KERNEL<4, 2, false>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__);        \
            else if (cfg.vec == 4 && cfg.maxi == 3) hipLaunchKernelGGL((KERNEL<4, 3, false>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__);   \
            else if (cfg.vec == 4 && cfg.maxi == 4) hipLaunchKernelGGL((KERNEL<4, 4, false>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__);   \
            else if (cfg.vec == 4 && cfg.maxi == 12) hipLaunchKernelGGL((KERNEL<4, 12, false>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__); \
            else if (cfg.vec == 4 && cfg.maxi == 16) hipLaunchKernelGGL((KERNEL<4, 16, false>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<1, 16, false>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__);        \
        }                                                                                                      \
    } while (0)

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
        if (cfg.vec == 4 && cfg.maxi == 2) hipLaunchKernelGGL((KERNEL<4, 2>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__);        \
        else if (cfg.vec == 4 && cfg.maxi == 3) hipLaunchKernelGGL((KERNEL<4, 3>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__);   \
        else if (cfg.vec == 4 && cfg.maxi == 4) hipLaunchKernelGGL((KERNEL<4, 4>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__);   \
        else if (cfg.vec == 4 && cfg.maxi == 12) hipLaunchKernelGGL((KERNEL<4, 12>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__); \
        else if (cfg.vec == 4 && cfg.maxi == 16) hipLaunchKernelGGL((KERNEL<4, 16>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<1, 16>), grid, dim3(RT), lds_bytes, st, __VA_ARGS__);                  \
    } while (0)

inline int check_dtype(int d) { return d == SPV_F32 || d == SPV_BF16; }
inline int pool_mode_of(int n, int k_in) {
    return k_in == n ? POOL_IDENT : (k_in % n == 0 ? POOL_EXACT : (n % k_in == 0 ? POOL_REPEAT : POOL_TABLE));
}

}  // namespace

extern "C" int64_t spv_rowop_partial_floats(int n) { return (int64_t)BWD_MAX_WG * 3 * n; }
// partial slabs a tail backward over `rows` rows writes (its grid): what a caller that folds them itself (spv_gemm_tn_fold) passes on
extern "C" int spv_tail_bwd_parts(int rows) { return std::min(cdiv(rows, RW), BWD_MAX_WG); }

extern "C" int spv_spectre_tail_fwd(const void* h, const void* x, const float* gamma, const float* beta, void* out,
                                    float* mean, float* rstd, int rows, int n, int k_in, int dtype, int out_dtype,
                                    float p_drop, uint64_t seed, void* stream) {
    SPV_CHECK(rows > 0 && n > 0 && k_in > 0, "spv_spectre_tail_fwd: empty");
    SPV_CHECK(check_dtype(dtype) && check_dtype(out_dtype), "spv_spectre_tail_fwd: bad dtype");
    SPV_CHECK(p_drop >= 0.0f && p_drop < 1.0f, "spv_spectre_tail_fwd: p_drop=%f", p_drop);
    RowCfg cfg;
    SPV_CHECK(pick_cfg(n, cfg), "spv_spectre_tail_fwd: unsupported row length %d", n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    SPV_CHECK((int64_t)n * k_in < (1ll << 31), "spv_spectre_tail_fwd: n*k_in too large");
    {
        const int fast = dtype == SPV_BF16, bfl = dtype == SPV_BF16, obf = out_dtype == SPV_BF16;
        static const int fwd_wgs = SPV_LAB_INT("SPV_TAIL_FWD_WGS", 2048);   // tuning aid (lab build)
        dim3 lgrid(std::min(cdiv(rows, RW), fwd_wgs));
#define LC_FWD(CO, CI)                                                                                                        \
        if (n == 64 * CO && k_in == 64 * CI && obf == bfl) {                                                                  \
            SPV_COUNT_PATH(SPV_PATH_TAIL_LC); \
            if (fast) hipLaunchKernelGGL((tail_fwd_lc_kernel<CO, CI, true, false>), lgrid, dim3(RT), 0, static_cast<hipStream_t>(stream), h, x, gamma, beta, out, mean, rstd, rows, bfl, obf, p_drop, seed, TailLn2{}); \
            else hipLaunchKernelGGL((tail_fwd_lc_kernel<CO, CI, false, false>), lgrid, dim3(RT), 0, static_cast<hipStream_t>(stream), h, x, gamma, beta, out, mean, rstd, rows, bfl, obf, p_drop, seed, TailLn2{});    \
            SPV_LAUNCH_CHECK("spv_spectre_tail_fwd(lc)");                                                                     \
            return 0;                                                                                                         \
        }
        LC_FWD(12, 8)
        LC_FWD(8, 12)
        LC_FWD(8, 8)   // 512 -> 512, identity skip: the MHPermutMix linear with its skip pooled by the gather
        LC_FWD(12, 48) // 3072 -> 768: linear3 at the Base width (exact windows of 4)
#undef LC_FWD
    }
    static const bool no_wide = SPV_LAB_SET("SPV_TAIL_NO_WIDE");   // A/B switch
    if (!no_wide && n == RW * 64 * WCO && k_in * WR == n && (out_dtype == SPV_BF16) == (dtype == SPV_BF16) &&
        (((uintptr_t)h | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0) {
        const dim3 wgrid(std::min(rows, 2048));
        if (dtype == SPV_BF16) hipLaunchKernelGGL((tail_wide_fwd_kernel<1, true>), wgrid, dim3(RT), 0, st, h, x, gamma, beta, out, mean, rstd, rows, p_drop, seed);
        else hipLaunchKernelGGL((tail_wide_fwd_kernel<0, false>), wgrid, dim3(RT), 0, st, h, x, gamma, beta, out, mean, rstd, rows, p_drop, seed);
        SPV_LAUNCH_CHECK("spv_spectre_tail_fwd(wide)");
        return 0;
    }
    const int pm = pool_mode_of(n, k_in);
    dim3 grid(std::min(cdiv(rows, RW), 2048));
    const size_t lds_f = pm == POOL_TABLE ? ((size_t)2 * n + (size_t)RW * k_in) * sizeof(float)
                         : (pm == POOL_EXACT && (k_in / n) % 4 == 0 ? (size_t)RW * (k_in / 4) * sizeof(float) : 0);
    SPV_CHECK(lds_f <= 64 * 1024, "spv_spectre_tail_fwd: n=%d k_in=%d needs %zu bytes of LDS", n, k_in, lds_f);
    TAIL_DISPATCH(cfg, dtype == SPV_BF16, tail_fwd_kernel, grid, lds_f, st, h, x, gamma, beta, out, mean,
                 rstd, rows, n, k_in, dtype == SPV_BF16, out_dtype == SPV_BF16, p_drop, seed, pm);
    SPV_LAUNCH_CHECK("spv_spectre_tail_fwd");
    return 0;
}

static int tail_bwd_impl(const void* dout, const void* h, const float* mean, const float* rstd, const float* gamma, const float* beta,
                         void* dh, void* dx_pool, float* dgamma, float* dbeta, float* dbias, float* partials, int rows, int n,
                         int k_in, int dtype, int dout_dtype, float p_drop, uint64_t seed, const void* dx_add, void* stream, TailUp up) {
    SPV_CHECK(rows > 0 && n > 0 && k_in > 0, "spv_spectre_tail_bwd: empty");
    SPV_CHECK(up.src == nullptr || (n == 768 && k_in == 512), "spv_spectre_tail_bwd_up: only the 512 -> 768 layer takes the skip gradient of the layer above");
    SPV_CHECK(check_dtype(dtype) && check_dtype(dout_dtype), "spv_spectre_tail_bwd: bad dtype");
    RowCfg cfg;
    SPV_CHECK(pick_cfg(n, cfg), "spv_spectre_tail_bwd: unsupported row length %d", n);
    hipStream_t st = static_cast<hipStream_t>(stream);
    SPV_CHECK((int64_t)n * k_in < (1ll << 31), "spv_spectre_tail_bwd: n*k_in too large");
    {
        const int fast = dtype == SPV_BF16, bfl = dtype == SPV_BF16, dbf = dout_dtype == SPV_BF16;
        static const int bwd_wgs = SPV_LAB_INT("SPV_TAIL_BWD_WGS", BWD_MAX_WG);   // tuning aid (lab build), <= BWD_MAX_WG
        const int lwgs = std::min(cdiv(rows, RW), std::min(bwd_wgs, BWD_MAX_WG));
        // no skip gradient asked for (the data-gradient GEMM adds it in its epilogue): the input width plays no part
        const int k_lc = (dx_pool == nullptr && up.src == nullptr) ? n : k_in;
#define LC_BWD(CO, CI)                                                                                                        \
        if (n == 64 * CO && k_lc == 64 * CI && (dbf == bfl || up.src != nullptr)) {                                          \
            SPV_CHECK(dbf == bfl, "spv_spectre_tail_bwd_up: dout must have the tensors' dtype");                              \
            hipStream_t lst = static_cast<hipStream_t>(stream);                                                               \
            SPV_COUNT_PATH(up.src ? SPV_PATH_TAIL_UP : SPV_PATH_TAIL_LC); \
            if (fast) hipLaunchKernelGGL((tail_bwd_lc_kernel<CO, CI, true, false>), dim3(lwgs), dim3(RT), (size_t)RW * 3 * n * sizeof(float), lst, dout, h, mean, rstd, gamma, beta, dh, dx_pool, partials, rows, bfl, dbf, p_drop, seed, dx_add, TailLn2{}, up); \
            else hipLaunchKernelGGL((tail_bwd_lc_kernel<CO, CI, false, false>), dim3(lwgs), dim3(RT), (size_t)RW * 3 * n * sizeof(float), lst, dout, h, mean, rstd, gamma, beta, dh, dx_pool, partials, rows, bfl, dbf, p_drop, seed, dx_add, TailLn2{}, up);    \
            SPV_LAUNCH_CHECK("spv_spectre_tail_bwd(lc)");                                                                     \
            if (dgamma != nullptr) hipLaunchKernelGGL(fold_partials_kernel, dim3(cdiv(3 * n, FOLD_COLS)), dim3(FOLD_COLS * FOLD_ROWS), 0, lst, partials, dgamma, dbeta, dbias, lwgs, 3, n); \
            SPV_LAUNCH_CHECK("spv_spectre_tail_bwd(lc fold)");                                                                \
            return 0;                                                                                                         \
        }
        LC_BWD(12, 8)
        LC_BWD(8, 12)
        LC_BWD(8, 8)
        LC_BWD(12, 48)
#undef LC_BWD
    }
    static const bool no_wide = SPV_LAB_SET("SPV_TAIL_NO_WIDE");   // A/B switch
    if (!no_wide && up.src == nullptr && n == RW * 64 * WCO && k_in * WR == n && (dout_dtype == SPV_BF16) == (dtype == SPV_BF16) &&
        (((uintptr_t)h | (uintptr_t)dout | (uintptr_t)dh | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)partials) & 15) == 0) {
        // one workgroup per row at a time; the slab count is the same function of rows as everywhere (spv_tail_bwd_parts)
        const int wwgs = std::min(cdiv(rows, RW), BWD_MAX_WG);
        if (dtype == SPV_BF16)
            hipLaunchKernelGGL((tail_wide_bwd_kernel<1, true>), dim3(wwgs), dim3(RT), 0, st, dout, h, mean, rstd, gamma, beta, dh, dx_pool, partials, rows, p_drop, seed, dx_add);
        else
            hipLaunchKernelGGL((tail_wide_bwd_kernel<0, false>), dim3(wwgs), dim3(RT), 0, st, dout, h, mean, rstd, gamma, beta, dh, dx_pool, partials, rows, p_drop, seed, dx_add);
        SPV_LAUNCH_CHECK("spv_spectre_tail_bwd(wide)");
        if (dgamma != nullptr) hipLaunchKernelGGL(fold_partials_kernel, dim3(cdiv(3 * n, FOLD_COLS)), dim3(FOLD_COLS * FOLD_ROWS), 0, st, partials, dgamma, dbeta, dbias, wwgs, 3, n);
        SPV_LAUNCH_CHECK("spv_spectre_tail_bwd(wide fold)");
        return 0;
    }
    const int pm = pool_mode_of(n, k_in);
    const int wgs = std::min(cdiv(rows, RW), BWD_MAX_WG);
    const size_t lds = (size_t)3 * n * sizeof(float) + (pm == POOL_TABLE ? (size_t)(2 * k_in + n + RW * n) * sizeof(int) : 0);
    SPV_CHECK(lds <= 64 * 1024, "spv_spectre_tail_bwd: n=%d k_in=%d needs %zu bytes of LDS", n, k_in, lds);
    TAIL_DISPATCH(cfg, dtype == SPV_BF16, tail_bwd_kernel, dim3(wgs), lds, st, dout, h, mean, rstd, gamma, beta, dh,
                 dx_pool, partials, rows, n, k_in, dtype == SPV_BF16, dout_dtype == SPV_BF16, p_drop, seed, pm, dx_add);
    SPV_LAUNCH_CHECK("spv_spectre_tail_bwd");
    if (dgamma != nullptr) hipLaunchKernelGGL(fold_partials_kernel, dim3(cdiv(3 * n, FOLD_COLS)), dim3(FOLD_COLS * FOLD_ROWS), 0, st, partials, dgamma, dbeta, dbias, wgs, 3, n);
    SPV_LAUNCH_CHECK("spv_spectre_tail_bwd(fold)");
    return 0;
}

extern "C" int spv_spectre_tail_bwd(const void* dout, const void* h, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, void* dh, void* dx_pool, float* dgamma,
                                    float* dbeta, float* dbias, float* partials, int rows, int n, int k_in, int dtype,
                                    int dout_dtype, float p_drop, uint64_t seed, const void* dx_add, void* stream) {
    return tail_bwd_impl(dout, h, mean, rstd, gamma, beta, dh, dx_pool, dgamma, dbeta, dbias, partials, rows, n, k_in, dtype, dout_dtype,
                         p_drop, seed, dx_add, stream, TailUp{nullptr, 0.0f, 0});
}

extern "C" int spv_tail_up_supported(int n, int k_in, int dtype) {
    static const bool off = SPV_LAB_SET("SPV_TAIL_NO_UP");
    return (!off && n == 768 && k_in == 512 && check_dtype(dtype)) ? 1 : 0;
}

extern "C" int spv_spectre_tail_bwd_up(const void* dout, const void* h, const float* mean, const float* rstd, const float* gamma,
                                       const float* beta, void* dh, void* dx_pool, float* dgamma, float* dbeta, float* dbias,
                                       float* partials, int rows, int n, int k_in, int dtype, int dout_dtype, float p_drop,
                                       uint64_t seed, const void* dx_add, const void* up_src, float up_p_drop, uint64_t up_seed,
                                       void* stream) {
    SPV_CHECK(up_src != nullptr && spv_tail_up_supported(n, k_in, dtype), "spv_spectre_tail_bwd_up: unsupported shape %d -> %d", k_in, n);
    SPV_CHECK(up_p_drop >= 0.0f && up_p_drop < 1.0f, "spv_spectre_tail_bwd_up: up_p_drop=%f", up_p_drop);
    return tail_bwd_impl(dout, h, mean, rstd, gamma, beta, dh, dx_pool, dgamma, dbeta, dbias, partials, rows, n, k_in, dtype, dout_dtype,
                         p_drop, seed, dx_add, stream, TailUp{up_src, up_p_drop, up_seed});
}

// ---- linear3 tail + residual + LayerNorm-2 as one kernel each way (512 outputs from 768 inputs: the lane-contiguous <8, 12> kernels)
extern "C" int spv_tail_ln_supported(int n, int k_in, int dtype) {
    static const bool off = SPV_LAB_SET("SPV_TAIL_NO_FUSE");
    return (!off && n == 512 && k_in == 768 && check_dtype(dtype)) ? 1 : 0;
}
extern "C" int64_t spv_tail_ln_partial_floats(int n) { return (int64_t)BWD_MAX_WG * 5 * n; }

extern "C" int spv_spectre_tail_ln_fwd(const void* h, const void* x, const float* gamma, const float* beta, void* out, float* mean,
                                       float* rstd, const void* res, const float* gamma2, const float* beta2, void* out2, float* mean2,
                                       float* rstd2, int rows, int n, int k_in, int dtype, float p_drop, uint64_t seed, void* stream) {
    SPV_CHECK(rows > 0 && spv_tail_ln_supported(n, k_in, dtype), "spv_spectre_tail_ln_fwd: unsupported shape %d -> %d / dtype %d", k_in, n, dtype);
    SPV_CHECK(p_drop >= 0.0f && p_drop < 1.0f, "spv_spectre_tail_ln_fwd: p_drop=%f", p_drop);
    SPV_CHECK(h && x && gamma && beta && out && mean && rstd && res && gamma2 && beta2 && out2 && mean2 && rstd2, "spv_spectre_tail_ln_fwd: null pointer");
    const int bfl = dtype == SPV_BF16;
    TailLn2 ln{res, gamma2, beta2, out2, mean2, rstd2, nullptr, nullptr, nullptr};
    dim3 lgrid(std::min(cdiv(rows, RW), 2048));
    hipStream_t st = static_cast<hipStream_t>(stream);
    SPV_COUNT_PATH(SPV_PATH_TAIL_LN);
    if (bfl) hipLaunchKernelGGL((tail_fwd_lc_kernel<8, 12, true, true>), lgrid, dim3(RT), 0, st, h, x, gamma, beta, out, mean, rstd, rows, bfl, bfl, p_drop, seed, ln);
    else hipLaunchKernelGGL((tail_fwd_lc_kernel<8, 12, false, true>), lgrid, dim3(RT), 0, st, h, x, gamma, beta, out, mean, rstd, rows, bfl, bfl, p_drop, seed, ln);
    SPV_LAUNCH_CHECK("spv_spectre_tail_ln_fwd");
    return 0;
}

extern "C" int spv_spectre_tail_ln_bwd(const void* dout2, const void* f3, const void* res, const float* mean2, const float* rstd2,
                                       const float* gamma2, void* ds, float* dgamma2, float* dbeta2, const void* h, const float* mean,
                                       const float* rstd, const float* gamma, const float* beta, void* dh, void* dx_pool, float* dgamma,
                                       float* dbeta, float* dbias, float* partials, int rows, int n, int k_in, int dtype, float p_drop,
                                       uint64_t seed, void* stream) {
    SPV_CHECK(rows > 0 && spv_tail_ln_supported(n, k_in, dtype), "spv_spectre_tail_ln_bwd: unsupported shape %d -> %d / dtype %d", k_in, n, dtype);
    SPV_CHECK(dout2 && f3 && res && mean2 && rstd2 && gamma2 && ds && h && mean && rstd && gamma && beta && dh && partials,
              "spv_spectre_tail_ln_bwd: null pointer");
    SPV_CHECK((dgamma && dbeta && dbias && dgamma2 && dbeta2) || (!dgamma && !dbeta && !dbias && !dgamma2 && !dbeta2),
              "spv_spectre_tail_ln_bwd: the five parameter gradients are given together, or all NULL (the caller folds the partials)");
    const int bfl = dtype == SPV_BF16;
    TailLn2 ln{res, gamma2, nullptr, nullptr, const_cast<float*>(mean2), const_cast<float*>(rstd2), dout2, f3, ds};
    const int lwgs = std::min(cdiv(rows, RW), BWD_MAX_WG);
    const size_t lds = (size_t)RW * 5 * n * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    SPV_COUNT_PATH(SPV_PATH_TAIL_LN);
    if (bfl) hipLaunchKernelGGL((tail_bwd_lc_kernel<8, 12, true, true>), dim3(lwgs), dim3(RT), lds, st, nullptr, h, mean, rstd, gamma, beta, dh, dx_pool, partials, rows, bfl, bfl, p_drop, seed, nullptr, ln, TailUp{nullptr, 0.0f, 0});
    else hipLaunchKernelGGL((tail_bwd_lc_kernel<8, 12, false, true>), dim3(lwgs), dim3(RT), lds, st, nullptr, h, mean, rstd, gamma, beta, dh, dx_pool, partials, rows, bfl, bfl, p_drop, seed, nullptr, ln, TailUp{nullptr, 0.0f, 0});
    SPV_LAUNCH_CHECK("spv_spectre_tail_ln_bwd");
    if (dgamma != nullptr) hipLaunchKernelGGL(fold_partials_kernel, dim3(cdiv(5 * n, FOLD_COLS)), dim3(FOLD_COLS * FOLD_ROWS), 0, st, partials, dgamma, dbeta, dbias, lwgs, 5, n, dgamma2, dbeta2);
    SPV_LAUNCH_CHECK("spv_spectre_tail_ln_bwd(fold)");
    return 0;
}

extern "C" int spv_haar_ln_supported(int dim, int dtype) {
    static const bool off = SPV_LAB_SET("SPV_HAAR_NO_FUSE");   // A/B switch
    return (!off && dtype == SPV_BF16 && (dim == 512 || dim == 1024)) ? 1 : 0;
}

extern "C" int spv_haar_ln_fwd(const void* x, const float* gamma, const float* beta, void* out, float* mean, float* rstd, int rows, int dim,
                               int dtype, void* stream) {
    SPV_CHECK(rows > 0 && spv_haar_ln_supported(dim, dtype), "spv_haar_ln_fwd: unsupported dim %d / dtype %d", dim, dtype);
    SPV_CHECK(x && gamma && beta && out && mean && rstd, "spv_haar_ln_fwd: null pointer");
    SPV_CHECK((((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "spv_haar_ln_fwd: pointers must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(std::min(cdiv(rows, RW), 2048));
    if (dim == 512) hipLaunchKernelGGL((haar_ln_fwd_kernel<1>), grid, dim3(RT), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)out, mean, rstd, rows);
    else hipLaunchKernelGGL((haar_ln_fwd_kernel<2>), grid, dim3(RT), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)out, mean, rstd, rows);
    SPV_LAUNCH_CHECK("spv_haar_ln_fwd");
    return 0;
}

// dgamma == NULL: the caller folds the partials (parts = spv_tail_bwd_parts(rows), nsum = 2, n = dim); partials: spv_rowop_partial_floats(dim)
extern "C" int spv_haar_ln_bwd(const void* dout, const void* x, const float* mean, const float* rstd, const float* gamma, void* dx,
                               float* dgamma, float* dbeta, float* partials, int rows, int dim, int dtype, void* stream) {
    SPV_CHECK(rows > 0 && spv_haar_ln_supported(dim, dtype), "spv_haar_ln_bwd: unsupported dim %d / dtype %d", dim, dtype);
    SPV_CHECK(dout && x && mean && rstd && gamma && dx && partials && ((dgamma != nullptr) == (dbeta != nullptr)), "spv_haar_ln_bwd: null pointer");
    SPV_CHECK((((uintptr_t)x | (uintptr_t)dout | (uintptr_t)dx | (uintptr_t)gamma) & 15) == 0, "spv_haar_ln_bwd: pointers must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int wgs = std::min(cdiv(rows, RW), BWD_MAX_WG);
    const size_t lds = (size_t)2 * dim * sizeof(float);
    if (dim == 512) hipLaunchKernelGGL((haar_ln_bwd_kernel<1>), dim3(wgs), dim3(RT), lds, st, (const bf16_t*)dout, (const bf16_t*)x, mean, rstd, gamma, (bf16_t*)dx, partials, rows);
    else hipLaunchKernelGGL((haar_ln_bwd_kernel<2>), dim3(wgs), dim3(RT), lds, st, (const bf16_t*)dout, (const bf16_t*)x, mean, rstd, gamma, (bf16_t*)dx, partials, rows);
    SPV_LAUNCH_CHECK("spv_haar_ln_bwd");
    if (dgamma != nullptr) hipLaunchKernelGGL(fold_partials_kernel, dim3(cdiv(2 * dim, FOLD_COLS)), dim3(FOLD_COLS * FOLD_ROWS), 0, st, partials, dgamma, dbeta, (float*)nullptr, wgs, 2, dim);
    SPV_LAUNCH_CHECK("spv_haar_ln_bwd(fold)");
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
    dim3 grid(std::min(cdiv(rows, RW), 2048));
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
    const int wgs = std::min(cdiv(rows, RW), BWD_MAX_WG);
    static const bool no_fast = SPV_LAB_SET("SPV_ADDLN_GENERIC");   // A/B switch
    const bool aligned16 = (((uintptr_t)dout | (uintptr_t)a | (uintptr_t)(b ? b : a) | (uintptr_t)din | (uintptr_t)gamma) & 15) == 0;
    if (!no_fast && aligned16 && (n == 512 || n == 768 || n == 1024)) {
        const size_t lds = (size_t)2 * n * sizeof(float);
        const bool bf = dtype == SPV_BF16;
#define SPV_ADDLN_FAST(MI)                                                                                                                 \
    do {                                                                                                                                   \
        if (bf && mode == 1) hipLaunchKernelGGL((addln_bwd_fast_kernel<MI, true, 1>), dim3(wgs), dim3(RT), lds, st, dout, a, b, mean, rstd, gamma, din, partials, rows);   \
        else if (bf) hipLaunchKernelGGL((addln_bwd_fast_kernel<MI, true, 0>), dim3(wgs), dim3(RT), lds, st, dout, a, b, mean, rstd, gamma, din, partials, rows);           \
        else if (mode == 1) hipLaunchKernelGGL((addln_bwd_fast_kernel<MI, false, 1>), dim3(wgs), dim3(RT), lds, st, dout, a, b, mean, rstd, gamma, din, partials, rows);   \
        else hipLaunchKernelGGL((addln_bwd_fast_kernel<MI, false, 0>), dim3(wgs), dim3(RT), lds, st, dout, a, b, mean, rstd, gamma, din, partials, rows);                  \
    } while (0)
        if (n == 512) SPV_ADDLN_FAST(2);
        else if (n == 768) SPV_ADDLN_FAST(3);
        else SPV_ADDLN_FAST(4);
#undef SPV_ADDLN_FAST
    } else {
        ROW_DISPATCH(cfg, addln_bwd_kernel, dim3(wgs), (size_t)2 * n * sizeof(float), st, dout, a, b, mean, rstd, gamma, din,
                     partials, rows, n, mode, dtype == SPV_BF16);
    }
    SPV_LAUNCH_CHECK("spv_add_layernorm_bwd");
    hipLaunchKernelGGL(fold_partials_kernel, dim3(cdiv(2 * n, FOLD_COLS)), dim3(FOLD_COLS * FOLD_ROWS), 0, st, partials, dgamma, dbeta, (float*)nullptr, wgs, 2, n);
    SPV_LAUNCH_CHECK("spv_add_layernorm_bwd(fold)");
    return 0;
}

int spv_seed_ptr_set_rowops(const unsigned long long* p) { return spv_seed_symbol_set(p); }
