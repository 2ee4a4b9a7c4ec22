// Shared device/host helpers for libspv_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/spv.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---------------------------------------------------------------- error plumbing (host)
int spv_set_error(const char* fmt, ...);
#define SPV_CHECK(cond, ...)                                   \
    do {                                                       \
        if (!(cond)) return spv_set_error(__VA_ARGS__);        \
    } while (0)
#define SPV_LAUNCH_CHECK(name)                                                         \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) return spv_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

extern long long g_spv_path_counts[SPV_PATH_COUNT];
#define SPV_COUNT_PATH(which) (++g_spv_path_counts[which])

// Development switches.  The product library reads NO environment variable: every A/B, tuning and diagnosis switch below exists only
// in the lab build (`make lab` -> lib/libspv_hip_lab.so, compiled with -DSPV_LAB and selected by tools/ through SPV_LIB_PATH).
#ifdef SPV_LAB
#include <cstdlib>
#define SPV_LAB_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define SPV_LAB_SET(name) (getenv(name) != nullptr)
#else
#define SPV_LAB_INT(name, dflt) (dflt)
#define SPV_LAB_SET(name) false
#endif

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- bf16 <-> f32 (device)
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even through the hardware convert (keeps NaN a NaN, guide 'Correctness boundaries')
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

// two floats -> one dword of two bf16 (lo in bits 0..15): ONE v_cvt_pk_bf16_f32.  (Converting the halves separately and joining them
// with shift + or compiles to four VALU instructions; the row kernels and the FNet mixer are VALU bound.)
typedef __bf16 spv_bf16x2 __attribute__((ext_vector_type(2)));
typedef float spv_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const spv_f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, spv_bf16x2));
}

template <typename T> struct io;
template <> struct io<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void st4(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct io<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    static __device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
        uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void st4(bf16_t* p, const float (&v)[4]) {
        uint2 t;
        t.x = pack_bf16x2(v[0], v[1]);
        t.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = t;
    }
};

// ---------------------------------------------------------------- wave64 reductions
// DPP (data-parallel primitives) inside each 16-lane row, then the four row totals through v_readlane: ~12 cheap VALU
// ops.  (__shfl_xor lowers to ds_bpermute_b32 -- six serialised LDS round trips per reduction, which left the
// one-wave-per-row kernels waiting most of the time.)  Every lane gets the result.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_lane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x124>(v);   // row_ror:4
    v += dpp_mov<0x128>(v);   // row_ror:8  -> each lane holds the sum of its 16-lane row
    return (row_lane(v, 0) + row_lane(v, 16)) + (row_lane(v, 32) + row_lane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x124>(v));
    v = fmaxf(v, dpp_mov<0x128>(v));
    return fmaxf(fmaxf(row_lane(v, 0), row_lane(v, 16)), fmaxf(row_lane(v, 32), row_lane(v, 48)));
}

// ---------------------------------------------------------------- fold of per-workgroup partial column sums
// out[p][c] = sum_w partials[w][p][c] (p < np <= 5, c < n), in a fixed order.  One 1024-thread workgroup covers FOLD_COLS columns;
// block `bid` of cdiv(np * n, FOLD_COLS).  Used by the row kernels' own fold launch and -- so that the backward of a layer does not
// pay a launch for it -- as extra workgroups of the weight gradient's split-K reduce (spv_gemm_tn_fold).
// A wave reads 2 partial rows x 32 columns = two whole 128-byte lines per load instruction (16 columns x 64 thread rows, rounds 1-3, read
// 64-byte half lines: the fold workgroups riding in the batched weight gradients' reduce were 27 of its 40 us).
constexpr int FOLD_COLS = 32, FOLD_ROWS = 32;
struct FoldJob {
    const float* partials;
    float* o[5];
    int parts, np, n;
};
// RPT = thread rows per thread: 1 = FOLD_COLS x FOLD_ROWS threads (1024); 4 = FOLD_COLS x 8 threads (256: a workgroup that fits on a CU
// BESIDE a workgroup of the batched weight gradients, see spv_fold_multi), each doing four thread rows' sums one after the other --
// the same additions in the same order, so every fold of the library gives the same bits whichever launch carries it.
template <int RPT = 1>
__device__ __forceinline__ void fold_partials_block(const FoldJob& j, int bid, int tid) {
    __shared__ float red[FOLD_ROWS][FOLD_COLS + 1];
    constexpr int TROWS = FOLD_ROWS / RPT;   // thread rows of the workgroup
    const int cx = tid % FOLD_COLS, pq = tid / FOLD_COLS;
    const int c = bid * FOLD_COLS + cx;
    const int total = j.np * j.n;
    const float* partials = j.partials;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int py = pq + r * TROWS;
        float s = 0.0f;
        if (c < total) {
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = 0.0f;
            int w = py;
            for (; w + 7 * FOLD_ROWS < j.parts; w += 8 * FOLD_ROWS) {   // eight independent loads in flight per thread (and row)
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] += partials[(size_t)(w + u * FOLD_ROWS) * total + c];
            }
            for (; w < j.parts; w += FOLD_ROWS) a[0] += partials[(size_t)w * total + c];
            s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        }
        red[py][cx] = s;
    }
    __syncthreads();
    // FOLD_ROWS -> 4 -> 1 in a fixed order
    if (pq < 4) {
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < FOLD_ROWS / 4; ++q) t += red[pq * (FOLD_ROWS / 4) + q][cx];
        red[pq * (FOLD_ROWS / 4)][cx] = t;
    }
    __syncthreads();
    if (pq == 0 && c < total) {
        const float t = (red[0][cx] + red[FOLD_ROWS / 4][cx]) + (red[FOLD_ROWS / 2][cx] + red[3 * FOLD_ROWS / 4][cx]);
        const int p = c / j.n, cc = c % j.n;
        float* o = j.o[p];
        if (o) o[cc] = t;
    }
}

// ---------------------------------------------------------------- math
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * __expf(-0.5f * x * x) * 0.39894228040143267794f;
}

// bf16 path: GELU and its derivative from one exp and one rcp (Abramowitz-Stegun 7.1.26 erfc, |err| <= 1.5e-7, far
// below bf16 resolution); the fp32 parity path keeps erff.
__device__ __forceinline__ void gelu_fast(float x, float& g, float& dg) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float u = __expf(-z * z);  // = exp(-x^2 / 2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float half_erfc = 0.5f * poly * u;
    const float phi = x >= 0.0f ? 1.0f - half_erfc : half_erfc;
    g = x * phi;
    dg = fmaf(x * u, 0.39894228040143267794f, phi);
}

// The same on two columns at once: every non-transcendental step is a packed v_pk_*_f32 instruction (the row kernels are
// VALU bound: ~26 instructions per element with the scalar form).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ void gelu_fast2(f32x2 x, f32x2& g, f32x2& dg) {
    const f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    const f32x2 z = ax * 0.70710678118654752440f;
    const f32x2 d = z * 0.3275911f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 nz2 = -(z * z);
    const f32x2 u = {__expf(nz2.x), __expf(nz2.y)};
    f32x2 poly = t * 1.061405429f + (-1.453152027f);
    poly = poly * t + 1.421413741f;
    poly = poly * t + (-0.284496736f);
    poly = poly * t + 0.254829592f;
    poly = poly * t;
    const f32x2 half_erfc = poly * u * 0.5f;
    const f32x2 phi = {x.x >= 0.0f ? 1.0f - half_erfc.x : half_erfc.x, x.y >= 0.0f ? 1.0f - half_erfc.y : half_erfc.y};
    g = x * phi;
    dg = (x * u) * 0.39894228040143267794f + phi;
}

// Counter-based dropout mask: keep(seed, row, col) -- regenerated identically in the backward, so no mask tensor is
// stored.  (The stream differs from torch's Philox; parity runs use p = 0.)  A per-row key is hashed once; every
// element costs one more 32-bit finaliser (murmur3-style) on key + col * golden.
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// HIP-graph replay: kernel arguments are frozen at capture time, so a dropout seed passed by value would repeat the captured
// masks in every replay.  spv_set_seed_device_ptr() points every translation unit's g_seed_dev at one device word that the
// caller advances once per step (inside the graph); each dropout kernel adds it to its by-value seed.  NULL (the default,
// eager mode): seeds are used as passed.
#ifdef SPV_USES_SEED
__device__ const unsigned long long* g_seed_dev = nullptr;
__device__ __forceinline__ uint64_t live_seed(uint64_t seed) { return g_seed_dev ? seed + *g_seed_dev : seed; }
inline int spv_seed_symbol_set(const unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_seed_dev), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

__device__ __forceinline__ unsigned dropout_row_key(uint64_t seed, uint64_t row) {
    return mix32((unsigned)seed + (unsigned)row * 0x9e3779b1u) ^ mix32((unsigned)(seed >> 32) + (unsigned)(row >> 32));
}
// the same mask for column first_col + c when key_base = row_key + (first_col >> 1) * 0x9e3779b1u (first_col even) and c is a
// compile-time offset: the per-pair constant folds into an add with a literal instead of a loop-invariant register per
// pair (the lane-contiguous tail kernels carried ten of those across their row loop and spilled)
__device__ __forceinline__ float dropout_scale_at(unsigned key_base, int c, float p, float inv_keep) {
    const unsigned h = mix32(key_base + (unsigned)(c >> 1) * 0x9e3779b1u);
    const unsigned u16 = (c & 1) ? (h >> 16) : (h & 0xffffu);
    return u16 < (unsigned)(p * 65536.0f + 0.5f) ? 0.0f : inv_keep;
}
__device__ __forceinline__ float dropout_scale(unsigned row_key, unsigned col, float p, float inv_keep) {
    // one finaliser per PAIR of columns, 16 bits each (drop probability quantised to 1/65536)
    const unsigned h = mix32(row_key + (col >> 1) * 0x9e3779b1u);
    const unsigned u16 = (col & 1) ? (h >> 16) : (h & 0xffffu);
    return u16 < (unsigned)(p * 65536.0f + 0.5f) ? 0.0f : inv_keep;
}
