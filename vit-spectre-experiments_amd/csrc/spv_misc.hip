// spv_misc.hip -- error plumbing + small bandwidth kernels (casts, transposes, GELU, column sums).
#include "spv_common.h"

#include <string.h>

static thread_local char g_err[512] = "";

int spv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

// dispatch census (test aid): which kernel family served a call -- spv_path_count(SPV_PATH_*)
long long g_spv_path_counts[SPV_PATH_COUNT] = {0};
extern "C" long long spv_path_count(int which) {
    return (which >= 0 && which < SPV_PATH_COUNT) ? g_spv_path_counts[which] : -1;
}

extern "C" int spv_version(void) { return SPV_ABI_VERSION; }
extern "C" const char* spv_last_error(void) { return g_err; }

namespace {

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float v[4];
        io<TI>::ld4(src + 4 * i, v);
        io<TO>::st4(dst + 4 * i, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        int64_t i = (n4 << 2) + threadIdx.x;
        io<TO>::st(dst + i, io<TI>::ld(src + i));
    }
}

// dst[c][r] = src[map(r)][c], dst leading dimension ld (>= rows, pad zero filled); 64x64 tile through LDS
// (+1 pad), 256 threads.  map(r) = r, or with rg > 0 the grouped row (r / rg) * gs + roff + r % rg
// (token rows of each image minus its CLS row).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int rows,
                                                             int cols, int ld, int rg, int gs, int roff) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        int r = r0 + i, c = c0 + tx;
        float v = 0.0f;
        if (r < rows && c < cols) {
            const size_t sr = rg > 0 ? (size_t)(r / rg) * gs + roff + (r % rg) : (size_t)r;
            v = io<TI>::ld(src + sr * cols + c);
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        int c = c0 + i, r = r0 + tx;
        if (c < cols && r < ld) io<TO>::st(dst + (size_t)c * ld + r, tile[tx][i]);
    }
}

// One pass over an fp32 weight [rows, cols]: the plain copy in the compute dtype (skipped when plain == nullptr) and the
// transposed copy [cols, ld] (zero beyond rows) -- the two operand layouts the NT GEMMs read.  Run once per weight per
// training step (the copies cannot be cached: see hip_ops._ShadowCache).
template <typename TO>
__device__ __forceinline__ void weight_shadow_tile(const float* __restrict__ src, TO* __restrict__ plain, TO* __restrict__ tr, int rows,
                                                   int cols, int ld, int bx, int by, float (*tile)[65]) {
    // 32 (rows) x 64 (cols) tile: 16-byte loads along the columns, the transposed copy leaves as 8 consecutive rows per thread
    const int r0 = by * 32, c0 = bx * 64;
    const int t = threadIdx.x;
    const bool vec = (cols & 3) == 0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int rl = pass * 16 + (t >> 4), c4 = (t & 15) * 4;
        const int r = r0 + rl, c = c0 + c4;
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (r < rows) {
            if (vec && c + 3 < cols) {
                const float4 f = *reinterpret_cast<const float4*>(src + (size_t)r * cols + c);
                v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
                if (plain != nullptr) io<TO>::st4(plain + (size_t)r * cols + c, v);
            } else {
                for (int u = 0; u < 4; ++u)
                    if (c + u < cols) {
                        v[u] = src[(size_t)r * cols + c + u];
                        if (plain != nullptr) io<TO>::st(plain + (size_t)r * cols + c + u, v[u]);
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) tile[rl][c4 + u] = v[u];
    }
    __syncthreads();
    const int cl = t >> 2, r8 = (t & 3) * 8;  // column cl of the tile, rows r8 .. r8 + 7
    const int c = c0 + cl;
    if (c < cols) {
        TO* o = tr + (size_t)c * ld + r0 + r8;
        if (r0 + r8 + 7 < ld && (ld & 7) == 0) {
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = tile[r8 + u][cl]; b[u] = tile[r8 + 4 + u][cl]; }
            io<TO>::st4(o, a);
            io<TO>::st4(o + 4, b);
        } else {
            for (int u = 0; u < 8; ++u)
                if (r0 + r8 + u < ld) io<TO>::st(o + u, tile[r8 + u][cl]);
        }
    }
}

template <typename TO>
__global__ __launch_bounds__(256) void weight_shadow_kernel(const float* __restrict__ src, TO* __restrict__ plain, TO* __restrict__ tr,
                                                            int rows, int cols, int ld) {
    __shared__ float tile[32][65];
    weight_shadow_tile<TO>(src, plain, tr, rows, cols, ld, blockIdx.x, blockIdx.y, tile);
}

// every weight of the model in ONE launch (8 launches of 4.9 us each per training step otherwise): workgroup b serves tile
// (tile_x[b], tile_y[b]) of tensor tile_tensor[b]
struct ShadowTensor { const float* src; void* plain; void* tr; int rows, cols, ld, pad; };
template <typename TO>
__global__ __launch_bounds__(256) void weight_shadow_multi_kernel(const ShadowTensor* __restrict__ tab, const int* __restrict__ tile_tensor,
                                                                  const int* __restrict__ tile_x, const int* __restrict__ tile_y) {
    __shared__ float tile[32][65];
    const ShadowTensor t = tab[tile_tensor[blockIdx.x]];
    weight_shadow_tile<TO>(t.src, static_cast<TO*>(t.plain), static_cast<TO*>(t.tr), t.rows, t.cols, t.ld, tile_x[blockIdx.x],
                           tile_y[blockIdx.x], tile);
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        io<T>::st(y + i, gelu_erf(io<T>::ld(x + i)));
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                       T* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        io<T>::st(dx + i, io<T>::ld(dy + i) * gelu_erf_grad(io<T>::ld(x + i)));
}
template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ out,
                                                    float a, float b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        io<T>::st(out + i, a * io<T>::ld(x + i) + b * io<T>::ld(y + i));
}

// column sums: block b sums rows b, b+G, ... for all columns into partials[b][n]; second kernel folds.
// column sums, stage 1: grid (column blocks, row slabs).  A 256-thread workgroup is TX column threads (VEC consecutive
// columns each: 16-/8-byte loads) x TY = 256/TX row threads; slab y sums rows y*TY + ty, + gridDim.y*TY, ...; the TY
// partial rows meet in LDS in a fixed order; one fp32 row of partials per slab.  Wide-and-short inputs (512 x 33280,
// the position-embedding gradient) get many column blocks, tall-and-narrow ones (33280 x 512) many slabs.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ partials, int rows,
                                                             int n, int tx_log2) {
    __shared__ float red[256 * VEC];
    const int TX = 1 << tx_log2, TY = 256 >> tx_log2;
    const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x >> tx_log2;
    const int c0 = (blockIdx.x * TX + tx) * VEC;
    float s[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s[k] = 0.0f;
    if (c0 < n) {
        const int step = gridDim.y * TY;
        for (int r = blockIdx.y * TY + ty; r < rows; r += step) {
            if constexpr (VEC == 4) {
                float v[4];
                io<T>::ld4(x + (size_t)r * n + c0, v);
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += v[k];
            } else {
                s[0] += io<T>::ld(x + (size_t)r * n + c0);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) red[(ty * TX + tx) * VEC + k] = s[k];
    __syncthreads();
    if (ty == 0 && c0 < n) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float t = s[k];
            for (int q = 1; q < TY; ++q) t += red[(q * TX + tx) * VEC + k];
            if (c0 + k < n) partials[(size_t)blockIdx.y * n + c0 + k] = t;
        }
    }
}
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* __restrict__ partials, float* __restrict__ out, int parts,
                                                          int n) {
    // a workgroup owns 32 columns; 8 thread rows split the partial slabs; fixed summation order
    __shared__ float red[8][33];
    const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    float s = 0.0f;
    if (c < n)
        for (int p = py; p < parts; p += 8) s += partials[(size_t)p * n + c];
    red[py][cx] = s;
    __syncthreads();
    if (py == 0 && c < n) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q][cx];
        out[c] = t;
    }
}

inline int ew_blocks(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 2048); }

}  // namespace

extern "C" int spv_cast(const void* src, int sd, void* dst, int dd, int64_t n, void* stream) {
    SPV_CHECK(n >= 0, "spv_cast: n < 0");
    if (n == 0) return 0;
    SPV_CHECK(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "spv_cast: pointers must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int g = ew_blocks((n + 3) / 4);
    if (sd == SPV_F32 && dd == SPV_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(g), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n);
    else if (sd == SPV_BF16 && dd == SPV_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n);
    else if (sd == SPV_F32 && dd == SPV_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)src, (float*)dst, n);
    else if (sd == SPV_BF16 && dd == SPV_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
    else
        return spv_set_error("spv_cast: bad dtypes %d -> %d", sd, dd);
    SPV_LAUNCH_CHECK("spv_cast");
    return 0;
}

extern "C" int spv_cast_transpose(const void* src, int sd, void* dst, int dd, int rows, int cols, int ld, int rg, int gs,
                                  int roff, void* stream) {
    SPV_CHECK(rows > 0 && cols > 0 && ld >= rows, "spv_cast_transpose: bad shape rows=%d cols=%d ld=%d", rows, cols, ld);
    SPV_CHECK(rg == 0 || (rg > 0 && gs >= rg + roff && roff >= 0), "spv_cast_transpose: bad grouping");
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(cdiv(cols, 64), cdiv(ld, 64));
    if (sd == SPV_F32 && dd == SPV_BF16)
        hipLaunchKernelGGL((cast_transpose_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float*)src, (bf16_t*)dst, rows, cols, ld, rg, gs, roff);
    else if (sd == SPV_BF16 && dd == SPV_F32)
        hipLaunchKernelGGL((cast_transpose_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const bf16_t*)src, (float*)dst, rows, cols, ld, rg, gs, roff);
    else if (sd == SPV_F32 && dd == SPV_F32)
        hipLaunchKernelGGL((cast_transpose_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, (float*)dst, rows, cols, ld, rg, gs, roff);
    else if (sd == SPV_BF16 && dd == SPV_BF16)
        hipLaunchKernelGGL((cast_transpose_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, rows, cols, ld, rg, gs, roff);
    else
        return spv_set_error("spv_cast_transpose: bad dtypes %d -> %d", sd, dd);
    SPV_LAUNCH_CHECK("spv_cast_transpose");
    return 0;
}

extern "C" int spv_weight_shadows_multi(const void* table, const int* tile_tensor, const int* tile_x, const int* tile_y, int ntiles,
                                        int dtype, void* stream) {
    SPV_CHECK(ntiles >= 0, "spv_weight_shadows_multi: ntiles = %d", ntiles);
    if (ntiles == 0) return 0;
    SPV_CHECK(table && tile_tensor && tile_x && tile_y, "spv_weight_shadows_multi: null table");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == SPV_BF16)
        hipLaunchKernelGGL((weight_shadow_multi_kernel<bf16_t>), dim3(ntiles), dim3(256), 0, st, static_cast<const ShadowTensor*>(table),
                           tile_tensor, tile_x, tile_y);
    else if (dtype == SPV_F32)
        hipLaunchKernelGGL((weight_shadow_multi_kernel<float>), dim3(ntiles), dim3(256), 0, st, static_cast<const ShadowTensor*>(table),
                           tile_tensor, tile_x, tile_y);
    else
        return spv_set_error("spv_weight_shadows_multi: bad dtype %d", dtype);
    SPV_LAUNCH_CHECK("spv_weight_shadows_multi");
    return 0;
}

extern "C" int spv_weight_shadows(const float* w, void* plain, void* transposed, int rows, int cols, int ld, int dtype, void* stream) {
    SPV_CHECK(rows > 0 && cols > 0 && ld >= rows, "spv_weight_shadows: bad shape %d x %d, ld %d", rows, cols, ld);
    SPV_CHECK(w != nullptr && transposed != nullptr, "spv_weight_shadows: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(cdiv(cols, 64), cdiv(ld, 32));
    if (dtype == SPV_BF16)
        hipLaunchKernelGGL((weight_shadow_kernel<bf16_t>), grid, dim3(256), 0, st, w, (bf16_t*)plain, (bf16_t*)transposed, rows, cols, ld);
    else if (dtype == SPV_F32)
        hipLaunchKernelGGL((weight_shadow_kernel<float>), grid, dim3(256), 0, st, w, (float*)plain, (float*)transposed, rows, cols, ld);
    else
        return spv_set_error("spv_weight_shadows: bad dtype %d", dtype);
    SPV_LAUNCH_CHECK("spv_weight_shadows");
    return 0;
}

#define DISPATCH_T(dtype, name, ...)                                        \
    if (dtype == SPV_F32) { using T = float; __VA_ARGS__; }                 \
    else if (dtype == SPV_BF16) { using T = bf16_t; __VA_ARGS__; }          \
    else return spv_set_error("%s: bad dtype %d", name, dtype);

extern "C" int spv_gelu_fwd(const void* x, void* y, int64_t n, int dtype, void* stream) {
    if (n <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DISPATCH_T(dtype, "spv_gelu_fwd",
               hipLaunchKernelGGL((gelu_fwd_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, st, (const T*)x, (T*)y, n));
    SPV_LAUNCH_CHECK("spv_gelu_fwd");
    return 0;
}
extern "C" int spv_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, void* stream) {
    if (n <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DISPATCH_T(dtype, "spv_gelu_bwd",
               hipLaunchKernelGGL((gelu_bwd_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, st, (const T*)dy, (const T*)x, (T*)dx, n));
    SPV_LAUNCH_CHECK("spv_gelu_bwd");
    return 0;
}
extern "C" int spv_axpby(const void* x, const void* y, void* out, float a, float b, int64_t n, int dtype, void* stream) {
    if (n <= 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DISPATCH_T(dtype, "spv_axpby",
               hipLaunchKernelGGL((axpby_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, st, (const T*)x, (const T*)y, (T*)out, a, b, n));
    SPV_LAUNCH_CHECK("spv_axpby");
    return 0;
}

// partials: fp32 scratch of >= min(rows,512) * n floats
extern "C" int spv_colsum(const void* x, float* out, float* partials, int rows, int n, int dtype, void* stream) {
    SPV_CHECK(rows > 0 && n > 0, "spv_colsum: empty");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int vec = (n % 4 == 0 && ((uintptr_t)x & 15) == 0) ? 4 : 1;
    const int cthreads = cdiv(n, vec);
    int tx_log2 = 5;  // at least 32 column threads
    while (tx_log2 < 8 && (1 << tx_log2) < cthreads) ++tx_log2;
    const int ty = 256 >> tx_log2;
    const int gx = cdiv(cthreads, 1 << tx_log2);
    // ~1024 workgroups in all, at most 512 slabs (the caller's partial buffer holds min(rows, 512) rows of n floats)
    const int parts = std::max(1, std::min(std::min(cdiv(rows, ty), 512), std::max(1, 1024 / gx)));
    if (vec == 4) {
        DISPATCH_T(dtype, "spv_colsum",
                   hipLaunchKernelGGL((colsum_partial_kernel<T, 4>), dim3(gx, parts), dim3(256), 0, st, (const T*)x, partials, rows, n, tx_log2));
    } else {
        DISPATCH_T(dtype, "spv_colsum",
                   hipLaunchKernelGGL((colsum_partial_kernel<T, 1>), dim3(gx, parts), dim3(256), 0, st, (const T*)x, partials, rows, n, tx_log2));
    }
    SPV_LAUNCH_CHECK("spv_colsum");
    hipLaunchKernelGGL(colsum_fold_kernel, dim3(cdiv(n, 32)), dim3(256), 0, st, partials, out, parts, n);
    SPV_LAUNCH_CHECK("spv_colsum(fold)");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// AdamW over many tensors in ONE launch (the optimizer step the training script drives: torch.optim.AdamW, reference
// spectre_vit/repl/train.py:199-201,237).  torch's fused multi-tensor kernel took 103-117 us per step for the FFT model's
// 63 tensors / 13 MB of parameters (latency bound: most tensors are 512-element LayerNorm vectors); one launch whose
// workgroups are dealt 2048-element chunks from a host-built chunk table streams the same bytes in a few microseconds.
// Arithmetic = torch.optim.AdamW (decoupled weight decay, bias correction, amsgrad off, maximize off).
namespace {
struct AdamTensor { float* p; const float* g; float* m; float* v; };

__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamTensor* __restrict__ tab, const int* __restrict__ chunk_tensor,
                                                          const int* __restrict__ chunk_off, const int* __restrict__ sizes, float lr,
                                                          float beta1, float beta2, float omb1, float omb2, float eps, float wd,
                                                          float bc1, float bc2, const float* __restrict__ step_dev) {
    const int t = chunk_tensor[blockIdx.x];
    const int off = chunk_off[blockIdx.x];
    const AdamTensor a = tab[t];
    const int n = sizes[t];
    if (step_dev != nullptr) {  // capturable mode: the step count lives on the device (already advanced for this step)
        const float s = *step_dev;
        bc1 = 1.0f - powf(beta1, s);
        bc2 = 1.0f - powf(beta2, s);
    }
    const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2), decay = 1.0f - lr * wd;
    const int base = off + threadIdx.x * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = base + h * 1024;
        if (i + 3 < n && ((reinterpret_cast<uintptr_t>(a.p) | reinterpret_cast<uintptr_t>(a.g) | reinterpret_cast<uintptr_t>(a.m) |
                           reinterpret_cast<uintptr_t>(a.v)) & 15) == 0) {
            float4 p = *reinterpret_cast<const float4*>(a.p + i), m = *reinterpret_cast<const float4*>(a.m + i);
            float4 v = *reinterpret_cast<const float4*>(a.v + i);
            const float4 g = *reinterpret_cast<const float4*>(a.g + i);
            float* pp = &p.x; float* mm = &m.x; float* vv = &v.x; const float* gg = &g.x;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mm[u] = beta1 * mm[u] + omb1 * gg[u];
                vv[u] = beta2 * vv[u] + omb2 * gg[u] * gg[u];
                pp[u] = pp[u] * decay - step_size * mm[u] / (sqrtf(vv[u]) * inv_sqrt_bc2 + eps);
            }
            *reinterpret_cast<float4*>(a.p + i) = p;
            *reinterpret_cast<float4*>(a.m + i) = m;
            *reinterpret_cast<float4*>(a.v + i) = v;
        } else {
            for (int u = 0; u < 4; ++u) {
                const int j = i + u;
                if (j >= n) break;
                const float g = a.g[j];
                const float m = beta1 * a.m[j] + omb1 * g;
                const float v = beta2 * a.v[j] + omb2 * g * g;
                a.m[j] = m;
                a.v[j] = v;
                a.p[j] = a.p[j] * decay - step_size * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
            }
        }
    }
}
}  // namespace

extern "C" int spv_adamw_multi(const void* table, const int* chunk_tensor, const int* chunk_off, const int* sizes, int nchunks, float lr,
                               float beta1, float beta2, float one_minus_beta1, float one_minus_beta2, float eps, float weight_decay,
                               float bias_correction1, float bias_correction2, const float* step_dev, void* stream) {
    SPV_CHECK(nchunks >= 0, "spv_adamw_multi: nchunks = %d", nchunks);
    if (nchunks == 0) return 0;
    SPV_CHECK(table && chunk_tensor && chunk_off && sizes, "spv_adamw_multi: null table");
    SPV_CHECK(step_dev != nullptr || (bias_correction1 > 0.0f && bias_correction2 > 0.0f), "spv_adamw_multi: bias corrections must be > 0");
    hipLaunchKernelGGL(adamw_multi_kernel, dim3(nchunks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const AdamTensor*>(table), chunk_tensor, chunk_off, sizes, lr, beta1, beta2, one_minus_beta1,
                       one_minus_beta2, eps, weight_decay, bias_correction1, bias_correction2, step_dev);
    SPV_LAUNCH_CHECK("spv_adamw_multi");
    return 0;
}

// ---- device-side seed word for HIP-graph capture (see spv_common.h: g_seed_dev)
int spv_seed_ptr_set_rowops(const unsigned long long* p);
int spv_seed_ptr_set_patch(const unsigned long long* p);
int spv_seed_ptr_set_attn(const unsigned long long* p);
int spv_seed_ptr_set_gemm(const unsigned long long* p);
namespace {
__global__ void seed_advance_kernel(unsigned long long* p) { *p += 0x9e3779b97f4a7c15ull; }
}
extern "C" int spv_set_seed_device_ptr(const void* seed_word) {
    const unsigned long long* p = static_cast<const unsigned long long*>(seed_word);
    SPV_CHECK(spv_seed_ptr_set_rowops(p) == 0 && spv_seed_ptr_set_patch(p) == 0 && spv_seed_ptr_set_attn(p) == 0 && spv_seed_ptr_set_gemm(p) == 0,
              "spv_set_seed_device_ptr: hipMemcpyToSymbol failed");
    return 0;
}
extern "C" int spv_seed_advance(void* seed_word, void* stream) {
    SPV_CHECK(seed_word != nullptr, "spv_seed_advance: null pointer");
    hipLaunchKernelGGL(seed_advance_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), static_cast<unsigned long long*>(seed_word));
    SPV_LAUNCH_CHECK("spv_seed_advance");
    return 0;
}
