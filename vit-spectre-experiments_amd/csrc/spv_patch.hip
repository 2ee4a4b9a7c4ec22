// spv_patch.hip -- patch-embedding plumbing around the MFMA GEMM: pixel-block extraction, CLS/position rows,
// the SpectralPatchEmbed weight fold (Re(rfft2 ortho) and the learnable frequency weights folded into the
// projection matrix), dropout as a stand-alone counter-based mask.
#define SPV_USES_SEED
#include "spv_common.h"

namespace {

__device__ __forceinline__ void st_any(void* base, size_t off, int bf, float v) {
    if (bf) static_cast<bf16_t*>(base)[off] = f2bf(v);
    else static_cast<float*>(base)[off] = v;
}
__device__ __forceinline__ float ld_any(const void* base, size_t off, int bf) {
    return bf ? bf2f(static_cast<const bf16_t*>(base)[off]) : static_cast<const float*>(base)[off];
}

// out[row][k] (transposed = 0, leading dim ld >= K) or out[k][row] (transposed = 1, ld >= rows);
// row = b * Np + ih * nW + iw,  k = c * P * P + p * P + q  (spectre.py:130-133 / Conv2d weight order)
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, void* __restrict__ out, int B, int C, int H,
                                                       int W, int P, int ld, int transposed, int bf) {
    const int nH = H / P, nW = W / P, Np = nH * nW, K = C * P * P;
    const int64_t rows = (int64_t)B * Np;
    // transposed == 2: token rows [B][Np + 1][ld], row 0 of every image (the CLS slot) zero -- the layout the token GEMM and the TN
    // weight-gradient GEMM both read as it lies
    const int64_t total = transposed == 1 ? (int64_t)K * ld : (transposed == 2 ? (int64_t)B * (Np + 1) * ld : rows * ld);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t row;
        int k;
        if (transposed == 1) { k = (int)(e / ld); row = e % ld; }
        else { row = e / ld; k = (int)(e % ld); }
        if (transposed == 2) {
            const int64_t b2 = row / (Np + 1);
            const int t2 = (int)(row % (Np + 1));
            row = t2 == 0 ? rows : b2 * Np + t2 - 1;   // rows = "no such row": zero
        }
        float v = 0.0f;
        if (row < rows && k < K) {
            const int b = (int)(row / Np), n = (int)(row % Np);
            const int ih = n / nW, iw = n % nW;
            const int c = k / (P * P), p = (k / P) % P, q = k % P;
            v = img[(((size_t)b * C + c) * H + ih * P + p) * W + iw * P + q];
        }
        st_any(out, (size_t)e, bf, v);
    }
}

// The same patch rows straight from the loader's uint8 HWC image (spectre_vit/repl/train.py:102-112: ToTensor = /255,
// then Normalize(mean, std) per channel): out = (img[b][y][x][c] / 255 - mean[c]) * inv_std[c].  One pass instead of
// host-side float conversion + normalise + NCHW copy + patchify.
__global__ __launch_bounds__(256) void patchify_u8_kernel(const unsigned char* __restrict__ img, const float* __restrict__ mean,
                                                          const float* __restrict__ inv_std, void* __restrict__ out, int B, int C,
                                                          int H, int W, int P, int ld, int transposed, int bf) {
    const int nH = H / P, nW = W / P, Np = nH * nW, K = C * P * P;
    const int64_t rows = (int64_t)B * Np;
    // transposed == 2: token rows [B][Np + 1][ld], row 0 of every image (the CLS slot) zero -- the layout the token GEMM and the TN
    // weight-gradient GEMM both read as it lies
    const int64_t total = transposed == 1 ? (int64_t)K * ld : (transposed == 2 ? (int64_t)B * (Np + 1) * ld : rows * ld);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t row;
        int k;
        if (transposed == 1) { k = (int)(e / ld); row = e % ld; }
        else { row = e / ld; k = (int)(e % ld); }
        if (transposed == 2) {
            const int64_t b2 = row / (Np + 1);
            const int t2 = (int)(row % (Np + 1));
            row = t2 == 0 ? rows : b2 * Np + t2 - 1;   // rows = "no such row": zero
        }
        float v = 0.0f;
        if (row < rows && k < K) {
            const int b = (int)(row / Np), n = (int)(row % Np);
            const int ih = n / nW, iw = n % nW;
            const int c = k / (P * P), p = (k / P) % P, q = k % P;
            const float px = (float)img[(((size_t)b * H + ih * P + p) * W + iw * P + q) * C + c];
            v = (px * (1.0f / 255.0f) - mean[c]) * inv_std[c];
        }
        st_any(out, (size_t)e, bf, v);
    }
}

// posbias[t][e] = pos[1 + t][e] + bias[e]
// with cls: one more row in front, out[0][e] = cls[e] + pos[0][e] -- the token GEMM over the zero CLS patch row then writes the CLS token
__global__ __launch_bounds__(256) void posbias_kernel(const float* __restrict__ pos, const float* __restrict__ bias,
                                                      const float* __restrict__ cls, float* __restrict__ out, int Np, int E) {
    const int lead = cls != nullptr ? E : 0;
    const int total = Np * E + lead;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x)
        out[i] = i < lead ? cls[i] + pos[i] : pos[E + i - lead] + bias[(i - lead) % E];
}

// tokens[b][0][e] = cls[e] + pos[0][e]
__global__ __launch_bounds__(256) void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                       void* __restrict__ tokens, int B, int T, int E, int bf) {
    const int total = B * E;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / E, e = i % E;
        st_any(tokens, ((size_t)b * T) * E + e, bf, cls[e] + pos[e]);
    }
}

// mask(i) = hash(key(i >> 12), i & 4095): 8 consecutive elements per thread share the block key and four pair hashes
// (16-byte bf16 / 2 x 16-byte fp32 accesses); the scalar form spent three hashes on every element
__global__ __launch_bounds__(256) void dropout_kernel(const void* __restrict__ x, void* __restrict__ y, int64_t n, float p,
                                                      uint64_t seed, int bf) {
    const float inv_keep = 1.0f / (1.0f - p);
    const int64_t n8 = n >> 3;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n8; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = v << 3;
        const unsigned key = dropout_row_key(live_seed(seed), (uint64_t)i >> 12);
        const unsigned c0 = (unsigned)(i & 4095);
        float sc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) sc[u] = dropout_scale(key, c0 + u, p, inv_keep);
        if (bf) {
            const uint4 t = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(x) + i);
            const unsigned w[4] = {t.x, t.y, t.z, t.w};
            unsigned o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                o[u] = pack_bf16x2(__uint_as_float(w[u] << 16) * sc[2 * u], __uint_as_float(w[u] & 0xffff0000u) * sc[2 * u + 1]);
            *reinterpret_cast<uint4*>(static_cast<bf16_t*>(y) + i) = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
            const float4 a = *reinterpret_cast<const float4*>(static_cast<const float*>(x) + i);
            const float4 b = *reinterpret_cast<const float4*>(static_cast<const float*>(x) + i + 4);
            *reinterpret_cast<float4*>(static_cast<float*>(y) + i) = make_float4(a.x * sc[0], a.y * sc[1], a.z * sc[2], a.w * sc[3]);
            *reinterpret_cast<float4*>(static_cast<float*>(y) + i + 4) = make_float4(b.x * sc[4], b.y * sc[5], b.z * sc[6], b.w * sc[7]);
        }
    }
    // ragged tail (n not a multiple of 8): scalar
    for (int64_t i = (n8 << 3) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        st_any(y, (size_t)i, bf, ld_any(x, (size_t)i, bf) *
                                     dropout_scale(dropout_row_key(live_seed(seed), (uint64_t)i >> 12), (unsigned)(i & 4095), p, inv_keep));
}

__device__ __forceinline__ float rcoef(int u, int v, int p, int q, int P) {
    // Re(rfft2(norm="ortho")) kernel: cos(2 pi (u p + v q) / P) / P      (spectre.py:136)
    return cospif(2.0f * (float)((u * p + v * q) % P) / (float)P) / (float)P;
}

// W_full[e][c,p,q] = sum_{u,v} W[e][c,u,v] fh[u] fw[v] R[(u,v),(p,q)]
__global__ __launch_bounds__(256) void spectral_fold_kernel(const float* __restrict__ w, const float* __restrict__ fh,
                                                            const float* __restrict__ fw, float* __restrict__ wf, int E, int C, int P) {
    const int Pv = P / 2 + 1;
    const int total = E * C * P * P;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int q = i % P, p = (i / P) % P, c = (i / (P * P)) % C, e = i / (P * P * C);
        const float* wr = w + ((size_t)e * C + c) * P * Pv;
        float a = 0.0f;
        for (int u = 0; u < P; ++u)
            for (int v = 0; v < Pv; ++v) a = fmaf(wr[u * Pv + v] * fh[u] * fw[v], rcoef(u, v, p, q, P), a);
        wf[i] = a;
    }
}

// G[e][c,u,v] = sum_{p,q} dWf[e][c,p,q] R[(u,v),(p,q)];  dW = G fh fw;  gw = G * W (for the frequency-weight sums)
__global__ __launch_bounds__(256) void spectral_fold_bwd_kernel(const float* __restrict__ dwf, const float* __restrict__ w,
                                                                const float* __restrict__ fh, const float* __restrict__ fw,
                                                                float* __restrict__ dw, float* __restrict__ gw, int E, int C, int P) {
    const int Pv = P / 2 + 1;
    const int total = E * C * P * Pv;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int v = i % Pv, u = (i / Pv) % P, ec = i / (Pv * P);
        const float* dr = dwf + (size_t)ec * P * P;
        float g = 0.0f;
        for (int p = 0; p < P; ++p)
            for (int q = 0; q < P; ++q) g = fmaf(dr[p * P + q], rcoef(u, v, p, q, P), g);
        dw[i] = g * fh[u] * fw[v];
        gw[i] = g * w[i];
    }
}
// dfh[u] = sum_{e,c,v} gw fw[v];  dfw[v] = sum_{e,c,u} gw fh[u]; single workgroup, fixed summation order
__global__ __launch_bounds__(256) void freq_weight_grad_kernel(const float* __restrict__ gw, const float* __restrict__ fh,
                                                               const float* __restrict__ fw, float* __restrict__ dfh,
                                                               float* __restrict__ dfw, int EC, int P) {
    extern __shared__ float lds[];  // [256][P + Pv]
    const int Pv = P / 2 + 1, S = P + Pv;
    float* mine = lds + threadIdx.x * S;
    for (int j = 0; j < S; ++j) mine[j] = 0.0f;
    for (int ec = threadIdx.x; ec < EC; ec += blockDim.x) {
        const float* g = gw + (size_t)ec * P * Pv;
        for (int u = 0; u < P; ++u)
            for (int v = 0; v < Pv; ++v) {
                const float t = g[u * Pv + v];
                mine[u] += t * fw[v];
                mine[P + v] += t * fh[u];
            }
    }
    __syncthreads();
    if (threadIdx.x < S) {
        float a = 0.0f;
        for (int t = 0; t < (int)blockDim.x; ++t) a += lds[t * S + threadIdx.x];
        if (threadIdx.x < P) dfh[threadIdx.x] = a;
        else dfw[threadIdx.x - P] = a;
    }
}

inline int ew_blocks(int64_t n) { return (int)std::min<int64_t>((n + 255) / 256, 4096); }

}  // namespace

extern "C" int spv_patchify(const float* img, void* out, int batch, int chans, int height, int width, int patch, int ld,
                            int transposed, int out_dtype, void* stream) {
    SPV_CHECK(batch > 0 && chans > 0 && patch > 0 && height >= patch && width >= patch, "spv_patchify: bad shape");
    SPV_CHECK(out_dtype == SPV_F32 || out_dtype == SPV_BF16, "spv_patchify: bad dtype");
    const int Np = (height / patch) * (width / patch), K = chans * patch * patch;
    const int64_t rows = (int64_t)batch * Np;
    SPV_CHECK(transposed >= 0 && transposed <= 2 && (transposed == 1 ? ld >= rows : ld >= K), "spv_patchify: ld=%d too small", ld);
    const int64_t total = transposed == 1 ? (int64_t)K * ld : (transposed == 2 ? (int64_t)batch * (Np + 1) * ld : rows * ld);
    hipLaunchKernelGGL(patchify_kernel, dim3(ew_blocks(total)), dim3(256), 0, static_cast<hipStream_t>(stream), img, out, batch,
                       chans, height, width, patch, ld, transposed, out_dtype == SPV_BF16);
    SPV_LAUNCH_CHECK("spv_patchify");
    return 0;
}

extern "C" int spv_patchify_u8(const unsigned char* img_hwc, const float* mean, const float* inv_std, void* out, int batch,
                               int chans, int height, int width, int patch, int ld, int transposed, int out_dtype, void* stream) {
    SPV_CHECK(batch > 0 && chans > 0 && patch > 0 && height >= patch && width >= patch, "spv_patchify_u8: bad shape");
    SPV_CHECK(out_dtype == SPV_F32 || out_dtype == SPV_BF16, "spv_patchify_u8: bad dtype");
    SPV_CHECK(mean != nullptr && inv_std != nullptr, "spv_patchify_u8: mean / inv_std missing");
    const int Np = (height / patch) * (width / patch), K = chans * patch * patch;
    const int64_t rows = (int64_t)batch * Np;
    SPV_CHECK(transposed >= 0 && transposed <= 2 && (transposed == 1 ? ld >= rows : ld >= K), "spv_patchify_u8: ld=%d too small", ld);
    const int64_t total = transposed == 1 ? (int64_t)K * ld : (transposed == 2 ? (int64_t)batch * (Np + 1) * ld : rows * ld);
    hipLaunchKernelGGL(patchify_u8_kernel, dim3(ew_blocks(total)), dim3(256), 0, static_cast<hipStream_t>(stream), img_hwc, mean,
                       inv_std, out, batch, chans, height, width, patch, ld, transposed, out_dtype == SPV_BF16);
    SPV_LAUNCH_CHECK("spv_patchify_u8");
    return 0;
}

extern "C" int spv_embed_posbias(const float* pos, const float* bias, const float* cls, float* out, int patches, int embed, void* stream) {
    SPV_CHECK(patches > 0 && embed > 0, "spv_embed_posbias: empty");
    hipLaunchKernelGGL(posbias_kernel, dim3(ew_blocks((int64_t)(patches + 1) * embed)), dim3(256), 0, static_cast<hipStream_t>(stream), pos,
                       bias, cls, out, patches, embed);
    SPV_LAUNCH_CHECK("spv_embed_posbias");
    return 0;
}

extern "C" int spv_embed_cls_rows(const float* cls, const float* pos, void* tokens, int batch, int tokens_per_image, int embed,
                                  int dtype, void* stream) {
    SPV_CHECK(batch > 0 && tokens_per_image > 0 && embed > 0, "spv_embed_cls_rows: empty");
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_embed_cls_rows: bad dtype");
    hipLaunchKernelGGL(cls_rows_kernel, dim3(ew_blocks((int64_t)batch * embed)), dim3(256), 0, static_cast<hipStream_t>(stream), cls,
                       pos, tokens, batch, tokens_per_image, embed, dtype == SPV_BF16);
    SPV_LAUNCH_CHECK("spv_embed_cls_rows");
    return 0;
}

extern "C" int spv_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream) {
    SPV_CHECK(p >= 0.0f && p < 1.0f, "spv_dropout: p=%f", p);
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_dropout: bad dtype");
    if (n <= 0) return 0;
    SPV_CHECK(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "spv_dropout: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(dropout_kernel, dim3(ew_blocks((n + 7) / 8)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n, p, seed,
                       dtype == SPV_BF16);
    SPV_LAUNCH_CHECK("spv_dropout");
    return 0;
}

extern "C" int spv_spectral_fold(const float* proj_w, const float* freq_h, const float* freq_w, float* w_full, int embed, int chans,
                                 int patch, void* stream) {
    SPV_CHECK(embed > 0 && chans > 0 && patch > 0, "spv_spectral_fold: empty");
    hipLaunchKernelGGL(spectral_fold_kernel, dim3(ew_blocks((int64_t)embed * chans * patch * patch)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), proj_w, freq_h, freq_w, w_full, embed, chans, patch);
    SPV_LAUNCH_CHECK("spv_spectral_fold");
    return 0;
}

// scratch: fp32, embed * chans * patch * (patch/2+1) floats
extern "C" int spv_spectral_fold_bwd(const float* dw_full, const float* proj_w, const float* freq_h, const float* freq_w,
                                     float* dproj_w, float* dfreq_h, float* dfreq_w, float* scratch, int embed, int chans, int patch,
                                     void* stream) {
    SPV_CHECK(embed > 0 && chans > 0 && patch > 0 && patch <= 64, "spv_spectral_fold_bwd: bad shape");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int Pv = patch / 2 + 1;
    hipLaunchKernelGGL(spectral_fold_bwd_kernel, dim3(ew_blocks((int64_t)embed * chans * patch * Pv)), dim3(256), 0, st, dw_full,
                       proj_w, freq_h, freq_w, dproj_w, scratch, embed, chans, patch);
    SPV_LAUNCH_CHECK("spv_spectral_fold_bwd");
    hipLaunchKernelGGL(freq_weight_grad_kernel, dim3(1), dim3(256), (size_t)256 * (patch + Pv) * sizeof(float), st, scratch, freq_h,
                       freq_w, dfreq_h, dfreq_w, embed * chans, patch);
    SPV_LAUNCH_CHECK("spv_spectral_fold_bwd(freq)");
    return 0;
}

int spv_seed_ptr_set_patch(const unsigned long long* p) { return spv_seed_symbol_set(p); }
