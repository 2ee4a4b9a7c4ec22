// spv_permut.hip -- MHPermutMix gather (signed multi-head permutation) forward and backward.
//
// forward : g[b, h, j] = x[b, perm[h][j]] * sign[h][j]
// backward: dx[b, i]   = sum_h sign[h][inv_h(i)] * dg[b, h, inv_h(i)]     (each perm[h] is a bijection)
// Both directions are GATHERS through a packed uint32 table (bit 31 = sign, bits 0..30 = source index):
// `fwd` table idx[h][j] = perm[h][j], `inv` table idx[h][i] = inv_h(i).  No atomics, deterministic.
//
// One workgroup per sample stages the source row (forward: the d-element token row; backward: one head's
// d-element slice at a time) in LDS with coalesced 16-byte loads, then gathers from LDS (random 2/4-byte
// LDS reads instead of random global reads) and writes the destination coalesced.  Rows too large for the
// 160 KB LDS (Base/224: d = 151 296) gather straight from global memory (L2 resident per sample).
#include "spv_common.h"

namespace {

constexpr int PT = 1024;         // threads per workgroup
constexpr int LDS_LIMIT = 150 * 1024;

__global__ __launch_bounds__(256) void permut_pack_kernel(const int64_t* __restrict__ perms, const float* __restrict__ signs,
                                                          uint32_t* __restrict__ fwd, uint32_t* __restrict__ inv, int heads, int d) {
    const int64_t total = (int64_t)heads * d;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (int64_t)gridDim.x * blockDim.x) {
        const int h = (int)(f / d), j = (int)(f % d);
        const uint32_t p = (uint32_t)perms[f];
        const uint32_t s = signs[f] < 0.0f ? 0x80000000u : 0u;
        fwd[f] = p | s;
        inv[(int64_t)h * d + p] = (uint32_t)j | s;
    }
}

template <typename T> struct raw;
template <> struct raw<float> { using type = uint32_t; static constexpr uint32_t SIGN = 0x80000000u; };
template <> struct raw<bf16_t> { using type = uint16_t; static constexpr uint32_t SIGN = 0x8000u; };

// forward, LDS path: grid = batch
template <typename T>
__global__ __launch_bounds__(PT) void gather_fwd_lds_kernel(const T* __restrict__ x, const uint32_t* __restrict__ idx,
                                                            T* __restrict__ g, int heads, int d, T* __restrict__ pooled, int pw) {
    using R = typename raw<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R* row = reinterpret_cast<R*>(smem);
    const int b = blockIdx.x;
    const R* xr = reinterpret_cast<const R*>(x) + (size_t)b * d;
    constexpr int VE = 16 / sizeof(T);  // elements per 16 B
    const int nv = d / VE;
    for (int v = threadIdx.x; v < nv; v += PT) reinterpret_cast<uint4*>(row)[v] = reinterpret_cast<const uint4*>(xr)[v];
    for (int e = nv * VE + threadIdx.x; e < d; e += PT) row[e] = xr[e];
    __syncthreads();
    const int64_t total = (int64_t)heads * d;
    R* go = reinterpret_cast<R*>(g) + (size_t)b * total;
    const int64_t t4 = total >> 2;
    for (int64_t q = threadIdx.x; q < t4; q += PT) {
        const uint4 id = reinterpret_cast<const uint4*>(idx)[q];
        R o[4];
        o[0] = row[id.x & 0x7fffffffu] ^ (R)((id.x >> 31) * raw<T>::SIGN);
        o[1] = row[id.y & 0x7fffffffu] ^ (R)((id.y >> 31) * raw<T>::SIGN);
        o[2] = row[id.z & 0x7fffffffu] ^ (R)((id.z >> 31) * raw<T>::SIGN);
        o[3] = row[id.w & 0x7fffffffu] ^ (R)((id.w >> 31) * raw<T>::SIGN);
        if constexpr (sizeof(T) == 2) {
            uint2 w;
            w.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
            w.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
            reinterpret_cast<uint2*>(go)[q] = w;
        } else {
            reinterpret_cast<uint4*>(go)[q] = make_uint4(o[0], o[1], o[2], o[3]);
        }
        if (pooled != nullptr) {
            // the SpectreLinear skip averages pw consecutive gathered elements (layers.py:93,101): this lane has 4 of them,
            // pw / 4 neighbouring lanes (consecutive q) the rest -> DPP quad / row reduction, first lane of the group writes
            float sm;
            if constexpr (sizeof(T) == 2)
                sm = (bf2f(o[0]) + bf2f(o[1])) + (bf2f(o[2]) + bf2f(o[3]));
            else
                sm = (__uint_as_float(o[0]) + __uint_as_float(o[1])) + (__uint_as_float(o[2]) + __uint_as_float(o[3]));
            const int gl = pw >> 2;  // lanes per window: 1, 2, 4 or 8 (host checks)
            if (gl >= 2) sm += dpp_mov<0xB1>(sm);   // quad_perm [1,0,3,2]
            if (gl >= 4) sm += dpp_mov<0x4E>(sm);   // quad_perm [2,3,0,1]
            if (gl >= 8) sm += dpp_mov<0x124>(sm);  // row_ror:4 -> lanes 0-3 also hold lanes 4-7's sum
            if ((q & (gl - 1)) == 0) io<T>::st(pooled + (size_t)b * (total / pw) + q / gl, sm * (1.0f / (float)pw));
        }
    }
    for (int64_t f = (t4 << 2) + threadIdx.x; f < total; f += PT) {
        const uint32_t id = idx[f];
        go[f] = row[id & 0x7fffffffu] ^ (R)((id >> 31) * raw<T>::SIGN);
    }
}

// forward, global path: grid = (chunks, batch)
template <typename T>
__global__ __launch_bounds__(256) void gather_fwd_global_kernel(const T* __restrict__ x, const uint32_t* __restrict__ idx,
                                                                T* __restrict__ g, int heads, int d) {
    using R = typename raw<T>::type;
    const int b = blockIdx.y;
    const R* xr = reinterpret_cast<const R*>(x) + (size_t)b * d;
    const int64_t total = (int64_t)heads * d;
    R* go = reinterpret_cast<R*>(g) + (size_t)b * total;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t id = idx[f];
        go[f] = xr[id & 0x7fffffffu] ^ (R)((id >> 31) * raw<T>::SIGN);
    }
}

// backward, LDS path: grid = batch; thread owns elements i = (it*PT + tid)*4 .. +3, it < MAX_IT
constexpr int MAX_IT = 10;  // d <= 4 * PT * MAX_IT = 40 960
template <typename T>
__global__ __launch_bounds__(PT) void gather_bwd_lds_kernel(const T* __restrict__ dg, const uint32_t* __restrict__ inv,
                                                            T* __restrict__ dx, int heads, int d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* slice = reinterpret_cast<T*>(smem);
    const int b = blockIdx.x;
    constexpr int VE = 16 / sizeof(T);
    const int nv = d / VE;
    float acc[MAX_IT][4];
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[it][k] = 0.0f;
    for (int h = 0; h < heads; ++h) {
        const T* src = dg + ((size_t)b * heads + h) * d;
        __syncthreads();  // previous head's gathers are done before the slice is overwritten
        for (int v = threadIdx.x; v < nv; v += PT) reinterpret_cast<uint4*>(slice)[v] = reinterpret_cast<const uint4*>(src)[v];
        for (int e = nv * VE + threadIdx.x; e < d; e += PT) slice[e] = src[e];
        __syncthreads();
        const uint32_t* ih = inv + (size_t)h * d;
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it) {
            const int i0 = (it * PT + threadIdx.x) * 4;
            if (i0 + 3 < d) {
                const uint4 id = *reinterpret_cast<const uint4*>(ih + i0);
                float v0 = io<T>::ld(slice + (id.x & 0x7fffffffu)), v1 = io<T>::ld(slice + (id.y & 0x7fffffffu));
                float v2 = io<T>::ld(slice + (id.z & 0x7fffffffu)), v3 = io<T>::ld(slice + (id.w & 0x7fffffffu));
                acc[it][0] += (id.x >> 31) ? -v0 : v0;
                acc[it][1] += (id.y >> 31) ? -v1 : v1;
                acc[it][2] += (id.z >> 31) ? -v2 : v2;
                acc[it][3] += (id.w >> 31) ? -v3 : v3;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i0 + k < d) {
                        const uint32_t id = ih[i0 + k];
                        float v = io<T>::ld(slice + (id & 0x7fffffffu));
                        acc[it][k] += (id >> 31) ? -v : v;
                    }
            }
        }
    }
    T* out = dx + (size_t)b * d;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int i0 = (it * PT + threadIdx.x) * 4;
        if (i0 + 3 < d) io<T>::st4(out + i0, acc[it]);
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k < d) io<T>::st(out + i0 + k, acc[it][k]);
        }
    }
}

// backward, global path: grid = (chunks, batch)
template <typename T>
__global__ __launch_bounds__(256) void gather_bwd_global_kernel(const T* __restrict__ dg, const uint32_t* __restrict__ inv,
                                                                T* __restrict__ dx, int heads, int d) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d; i += gridDim.x * blockDim.x) {
        float a = 0.0f;
        for (int h = 0; h < heads; ++h) {
            const uint32_t id = inv[(size_t)h * d + i];
            float v = io<T>::ld(dg + ((size_t)b * heads + h) * d + (id & 0x7fffffffu));
            a += (id >> 31) ? -v : v;
        }
        io<T>::st(dx + (size_t)b * d + i, a);
    }
}

}  // namespace

// idx: uint32 [2][heads][d]: [0] = forward table, [1] = inverse table
extern "C" int spv_permut_pack(const int64_t* perms, const float* signs, uint32_t* idx, int heads, int d, void* stream) {
    SPV_CHECK(heads > 0 && d > 0, "spv_permut_pack: empty");
    SPV_CHECK((int64_t)d < (1ll << 31), "spv_permut_pack: d too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)heads * d;
    hipLaunchKernelGGL(permut_pack_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, st, perms, signs,
                       idx, idx + total, heads, d);
    SPV_LAUNCH_CHECK("spv_permut_pack");
    return 0;
}

extern "C" int spv_permut_gather_fwd(const void* x, const uint32_t* idx, void* g, void* pooled, int pool_window, int batch,
                                     int heads, int d, int dtype, void* stream) {
    SPV_CHECK(batch > 0 && heads > 0 && d > 0, "spv_permut_gather_fwd: empty");
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_permut_gather_fwd: bad dtype %d", dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t es = dtype == SPV_BF16 ? 2 : 4;
    const bool aligned = ((size_t)d * es) % 16 == 0 && ((int64_t)heads * d) % 4 == 0;
    if (pooled != nullptr) {
        const int64_t total = (int64_t)heads * d;
        SPV_CHECK(aligned && (size_t)d * es <= (size_t)LDS_LIMIT, "spv_permut_gather_fwd: pooled output needs the LDS path");
        SPV_CHECK((pool_window == 4 || pool_window == 8 || pool_window == 16 || pool_window == 32) && total % pool_window == 0 &&
                      (total / 4) % PT == 0,
                  "spv_permut_gather_fwd: unsupported pool window %d", pool_window);
    }
    if (aligned && (size_t)d * es <= (size_t)LDS_LIMIT) {
        const size_t lds = (size_t)d * es;
        if (dtype == SPV_BF16) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_fwd_lds_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
            SPV_COUNT_PATH(SPV_PATH_GATHER_LDS);
            hipLaunchKernelGGL((gather_fwd_lds_kernel<bf16_t>), dim3(batch), dim3(PT), lds, st, (const bf16_t*)x, idx, (bf16_t*)g, heads, d, (bf16_t*)pooled, pool_window);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_fwd_lds_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
            hipLaunchKernelGGL((gather_fwd_lds_kernel<float>), dim3(batch), dim3(PT), lds, st, (const float*)x, idx, (float*)g, heads, d, (float*)pooled, pool_window);
        }
    } else {
        dim3 grid((unsigned)std::min<int64_t>(((int64_t)heads * d + 255) / 256, 1024), batch);
        if (dtype == SPV_BF16)
            hipLaunchKernelGGL((gather_fwd_global_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, idx, (bf16_t*)g, heads, d);
        else
            hipLaunchKernelGGL((gather_fwd_global_kernel<float>), grid, dim3(256), 0, st, (const float*)x, idx, (float*)g, heads, d);
    }
    SPV_LAUNCH_CHECK("spv_permut_gather_fwd");
    return 0;
}

extern "C" int spv_permut_gather_bwd(const void* dg, const uint32_t* idx, void* dx, int batch, int heads, int d, int dtype,
                                     void* stream) {
    SPV_CHECK(batch > 0 && heads > 0 && d > 0, "spv_permut_gather_bwd: empty");
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_permut_gather_bwd: bad dtype %d", dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint32_t* inv = idx + (size_t)heads * d;
    const size_t es = dtype == SPV_BF16 ? 2 : 4;
    const bool aligned = ((size_t)d * es) % 16 == 0 && d % 4 == 0;
    if (aligned && (size_t)d * es <= (size_t)LDS_LIMIT && d <= 4 * PT * MAX_IT) {
        const size_t lds = (size_t)d * es;
        if (dtype == SPV_BF16) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_bwd_lds_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
            hipLaunchKernelGGL((gather_bwd_lds_kernel<bf16_t>), dim3(batch), dim3(PT), lds, st, (const bf16_t*)dg, inv, (bf16_t*)dx, heads, d);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_bwd_lds_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
            hipLaunchKernelGGL((gather_bwd_lds_kernel<float>), dim3(batch), dim3(PT), lds, st, (const float*)dg, inv, (float*)dx, heads, d);
        }
    } else {
        dim3 grid((unsigned)std::min((d + 255) / 256, 1024), batch);
        if (dtype == SPV_BF16)
            hipLaunchKernelGGL((gather_bwd_global_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)dg, inv, (bf16_t*)dx, heads, d);
        else
            hipLaunchKernelGGL((gather_bwd_global_kernel<float>), grid, dim3(256), 0, st, (const float*)dg, inv, (float*)dx, heads, d);
    }
    SPV_LAUNCH_CHECK("spv_permut_gather_bwd");
    return 0;
}
