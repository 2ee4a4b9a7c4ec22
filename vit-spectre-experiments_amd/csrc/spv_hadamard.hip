// spv_hadamard.hip -- Walsh-Hadamard butterflies along the last axis.
// Reference: spectre_vit/models/spectre/hadamar.py -- fwht :12-32 and hadamard_transform :83-112 (natural / Sylvester order,
// optional n^-1/2), fwht_fast :58-80 (each stage writes sum / difference INTERLEAVED, so the output order differs and nothing is
// normalised), LearnableHadamard :115-141 (pad to a power of two, num_blocks x fwht_fast, crop, + residual; its parameters are
// unused: the `* p` is commented out at :136).
#include "spv_common.h"

namespace {

constexpr int HT = 256;

// One workgroup owns `rpw` rows; a row lives in LDS as fp32 (two buffers, ping-pong), one butterfly per thread and stage.
//   mode 0: natural order      stage h: (i, i+h) -> (i, i+h)            (its own transpose)
//   mode 1: fwht_fast          stage h: a=[0,h) b=[h,2h) of a 2h block -> out[2i] = a+b, out[2i+1] = a-b
//   mode 2: fwht_fast^T        stages in reverse; in[2i], in[2i+1] -> out[i] = sum, out[h+i] = difference
template <typename T>
__global__ __launch_bounds__(HT) void fwht_kernel(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ residual, int rows,
                                                  int n_in, int n, int n_out, int log2n, int mode, int repeat, float scale, int rpw) {
    extern __shared__ float lds[];
    float* buf0 = lds;
    float* buf1 = lds + (size_t)rpw * n;
    const int row0 = blockIdx.x * rpw;
    const int nrows = min(rpw, rows - row0);
    if (nrows <= 0) return;
    for (int e = threadIdx.x; e < nrows * n; e += HT) {
        const int r = e / n, c = e - r * n;
        buf0[e] = c < n_in ? io<T>::ld(x + (size_t)(row0 + r) * n_in + c) : 0.0f;  // F.pad(x, (0, pad))
    }
    __syncthreads();
    float* src = buf0;
    float* dst = buf1;
    const int half = n >> 1;
    for (int rep = 0; rep < repeat; ++rep) {
        for (int s = 0; s < log2n; ++s) {
            const int h = (mode == 2) ? (half >> s) : (1 << s);
            for (int e = threadIdx.x; e < nrows * half; e += HT) {
                const int r = e / half, j = e - r * half;
                const int blk = j / h, i = j - blk * h;
                const float* sp = src + (size_t)r * n + blk * 2 * h;
                float* dp = dst + (size_t)r * n + blk * 2 * h;
                if (mode == 2) {
                    const float u = sp[2 * i], v = sp[2 * i + 1];
                    dp[i] = u + v;
                    dp[h + i] = u - v;
                } else {
                    const float a = sp[i], b = sp[h + i];
                    if (mode == 0) { dp[i] = a + b; dp[h + i] = a - b; }
                    else { dp[2 * i] = a + b; dp[2 * i + 1] = a - b; }
                }
            }
            __syncthreads();
            float* t = src; src = dst; dst = t;
        }
    }
    for (int e = threadIdx.x; e < nrows * n_out; e += HT) {
        const int r = e / n_out, c = e - r * n_out;
        float v = src[(size_t)r * n + c] * scale;                                   // x[..., :orig_dim]
        if (residual) v += io<T>::ld(residual + (size_t)(row0 + r) * n_out + c);   // + residual
        io<T>::st(y + (size_t)(row0 + r) * n_out + c, v);
    }
}

}  // namespace

extern "C" int spv_fwht(const void* x, void* y, const void* residual, int rows, int n_in, int n, int n_out, int mode, int repeat,
                        float scale, int dtype, void* stream) {
    SPV_CHECK(x && y, "spv_fwht: null pointer");
    SPV_CHECK(rows >= 0 && n >= 1 && (n & (n - 1)) == 0 && n <= 16384, "spv_fwht: n = %d must be a power of two <= 16384", n);
    SPV_CHECK(n_in >= 1 && n_in <= n && n_out >= 1 && n_out <= n, "spv_fwht: n_in %d / n_out %d outside 1..%d", n_in, n_out, n);
    SPV_CHECK(mode >= 0 && mode <= 2 && repeat >= 1, "spv_fwht: bad mode %d / repeat %d", mode, repeat);
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_fwht: bad dtype %d", dtype);
    if (rows == 0) return 0;
    int log2n = 0;
    while ((1 << log2n) < n) ++log2n;
    const int rpw = n >= 2 * HT ? 1 : (2 * HT) / n;  // keep every thread busy on short rows
    const size_t lds = (size_t)2 * rpw * n * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(cdiv(rows, rpw));
    if (dtype == SPV_BF16) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fwht_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(fwht_kernel<bf16_t>, grid, dim3(HT), lds, st, (const bf16_t*)x, (bf16_t*)y, (const bf16_t*)residual, rows, n_in, n,
                           n_out, log2n, mode, repeat, scale, rpw);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fwht_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(fwht_kernel<float>, grid, dim3(HT), lds, st, (const float*)x, (float*)y, (const float*)residual, rows, n_in, n,
                           n_out, log2n, mode, repeat, scale, rpw);
    }
    SPV_LAUNCH_CHECK("spv_fwht");
    return 0;
}
