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

// float NCHW, patch and width multiples of 4, row-major outputs (modes 0 and 2), ld % 4 == 0: one thread per FOUR consecutive k (one
// 16-byte pixel load, one 8/16-byte store).  The element-per-thread kernel above spends ~10 integer divisions per element: 14.4 us for
// the 3 MB of the CIFAR batch; this form ~3 us.
template <typename T>
__global__ __launch_bounds__(256) void patchify_vec4_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int C, int H, int W,
                                                            int P, int ld, int token_rows) {
    const int nH = H / P, nW = W / P, Np = nH * nW, K = C * P * P;
    const int T1 = token_rows ? Np + 1 : Np;
    const int ld4 = ld >> 2, P4 = P >> 2;
    const int64_t total = (int64_t)B * T1 * ld4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / ld4;
        const int k = (int)(e - row * ld4) * 4;
        const int b = (int)(row / T1), t = (int)(row - (int64_t)b * T1);
        const int n = token_rows ? t - 1 : t;
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (n >= 0 && k < K) {
            const int ih = n / nW, iw = n - ih * nW;
            const int c = k / (P * P), r = k - c * P * P, pr = r / P, q = r - pr * P;   // q is a multiple of 4
            const float4 px = *reinterpret_cast<const float4*>(img + (((size_t)b * C + c) * H + ih * P + pr) * W + iw * P + q);
            v[0] = px.x; v[1] = px.y; v[2] = px.z; v[3] = px.w;
        }
        (void)P4;
        io<T>::st4(out + (size_t)row * ld + k, v);
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
                                                            const float* __restrict__ fw, float* __restrict__ wf, int E, int C, int P,
                                                            bf16_t* __restrict__ wf_bf = nullptr) {
    const int Pv = P / 2 + 1;
    const int total = E * C * P * P;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int q = i % P, p = (i / P) % P, c = (i / (P * P)) % C, e = i / (P * P * C);
        const float* wr = w + ((size_t)e * C + c) * P * Pv;
        float a = 0.0f;
        for (int u = 0; u < P; ++u)
            for (int v = 0; v < Pv; ++v) a = fmaf(wr[u * Pv + v] * fh[u] * fw[v], rcoef(u, v, p, q, P), a);
        wf[i] = a;
        if (wf_bf != nullptr) wf_bf[i] = f2bf(a);   // the GEMM operand of a bf16 step: saves the cast launch that followed
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
// dfh[u] = sum_{e,c,v} gw fw[v];  dfw[v] = sum_{e,c,u} gw fh[u]; single workgroup (power-of-two size), fixed summation order: each
// thread's (e,c) slices into its own LDS row, then a binary tree over the rows.  (256 threads and a serial 256-term tail: 14.5 us for
// 18 432 numbers; this form ~4 us.)
__global__ __launch_bounds__(1024) void freq_weight_grad_kernel(const float* __restrict__ gw, const float* __restrict__ fh,
                                                                const float* __restrict__ fw, float* __restrict__ dfh,
                                                                float* __restrict__ dfw, int EC, int P) {
    extern __shared__ float lds[];  // [blockDim.x][P + Pv]
    const int Pv = P / 2 + 1, S = P + Pv;
    const int nt = blockDim.x;
    float* mine = lds + threadIdx.x * S;
    for (int j = 0; j < S; ++j) mine[j] = 0.0f;
    for (int ec = threadIdx.x; ec < EC; ec += nt) {
        const float* g = gw + (size_t)ec * P * Pv;
        for (int u = 0; u < P; ++u)
            for (int v = 0; v < Pv; ++v) {
                const float t = g[u * Pv + v];
                mine[u] += t * fw[v];
                mine[P + v] += t * fh[u];
            }
    }
    for (int stride = nt >> 1; stride >= 1; stride >>= 1) {
        __syncthreads();
        if ((int)threadIdx.x < stride) {
            const float* other = lds + (threadIdx.x + stride) * S;
            for (int j = 0; j < S; ++j) mine[j] += other[j];
        }
    }
    if (threadIdx.x == 0) {
        for (int j = 0; j < P; ++j) dfh[j] = mine[j];
        for (int j = 0; j < Pv; ++j) dfw[j] = mine[P + j];
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
    if (transposed != 1 && patch % 4 == 0 && width % 4 == 0 && ld % 4 == 0 && ((uintptr_t)img & 15) == 0 && ((uintptr_t)out & 15) == 0) {
        const int64_t vecs = (int64_t)batch * (Np + (transposed == 2 ? 1 : 0)) * (ld / 4);
        if (out_dtype == SPV_BF16)
            hipLaunchKernelGGL((patchify_vec4_kernel<bf16_t>), dim3(ew_blocks(vecs)), dim3(256), 0, static_cast<hipStream_t>(stream), img, (bf16_t*)out,
                               batch, chans, height, width, patch, ld, transposed == 2);
        else
            hipLaunchKernelGGL((patchify_vec4_kernel<float>), dim3(ew_blocks(vecs)), dim3(256), 0, static_cast<hipStream_t>(stream), img, (float*)out,
                               batch, chans, height, width, patch, ld, transposed == 2);
        SPV_LAUNCH_CHECK("spv_patchify(vec4)");
        return 0;
    }
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

// ---- backward of the token tensor, one pass: dtok = dropout_mask(g [+ gcls on the CLS rows]) and the column sums over the batch
// that the position / bias / class-token gradients are made of.  (Separately this was a strided add, a dropout pass, a column-sum
// pass and three small folds over a 34 MB tensor: seven launches.)  grid = (column blocks of 8-element vectors, groups of EB_GS samples).
constexpr int EB_GS = 16, EB_HB = 8;   // samples per workgroup row-group; loads in flight per thread
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_rows_kernel(const T* __restrict__ g, const T* __restrict__ gcls, T* __restrict__ dtok,
                                                            float* __restrict__ partials, int B, int TE, int E, float p, uint64_t seed) {
    const int v = blockIdx.x * 256 + threadIdx.x;          // 8-element vector of the [T * E] row
    if (v * 8 >= TE) return;
    const int col = v * 8;
    const int b0 = blockIdx.y * EB_GS;
    const float inv_keep = 1.0f / (1.0f - p);
    const uint64_t sd = live_seed(seed);
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.0f;
    const bool cls_row = gcls != nullptr && col < E;
    if constexpr (sizeof(T) == 2) {
        // bf16: one 16-byte load per row (two 8-byte loads were half the bytes per request)
        typedef unsigned raw4 __attribute__((ext_vector_type(4)));
        // (EB_HB rows at a time: < 128 VGPRs, so that this 256-thread workgroup fits on a CU BESIDE a workgroup of the batched weight
        // gradients -- 8 waves x 184 VGPRs leave 144 per SIMD -- which is where it runs: at 145 it waited for that kernel's CUs to drain)
        for (int h0 = 0; h0 < EB_GS; h0 += EB_HB) {
            raw4 xr[EB_HB], cr[EB_HB];
#pragma unroll
            for (int s = 0; s < EB_HB; ++s) xr[s] = *reinterpret_cast<const raw4*>(g + (size_t)min(b0 + h0 + s, B - 1) * TE + col);
            if (cls_row) {   // (one workgroup of the grid's column blocks)
#pragma unroll
                for (int s = 0; s < EB_HB; ++s) cr[s] = *reinterpret_cast<const raw4*>(gcls + (size_t)min(b0 + h0 + s, B - 1) * E + col);
            }
#pragma unroll
            for (int s = 0; s < EB_HB; ++s) {
                const int b = b0 + h0 + s;
                if (b >= B) break;
                const size_t i = (size_t)b * TE + col;
                float x[8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    x[2 * u] = __uint_as_float(xr[s][u] << 16);
                    x[2 * u + 1] = __uint_as_float(xr[s][u] & 0xffff0000u);
                }
                if (cls_row) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        x[2 * u] += __uint_as_float(cr[s][u] << 16);
                        x[2 * u + 1] += __uint_as_float(cr[s][u] & 0xffff0000u);
                    }
                }
                if (p > 0.0f) {
                    const unsigned key = dropout_row_key(sd, (uint64_t)i >> 12);
                    const unsigned c0 = (unsigned)(i & 4095);
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] *= dropout_scale(key, c0 + u, p, inv_keep);
                }
                raw4 o;
#pragma unroll
                for (int u = 0; u < 4; ++u) o[u] = pack_bf16x2(x[2 * u], x[2 * u + 1]);
                if (dtok != nullptr) *reinterpret_cast<raw4*>(dtok + i) = o;
#pragma unroll
                for (int u = 0; u < 4; ++u) {   // the sums are taken over what the weight-gradient GEMM reads: the rounded values
                    acc[2 * u] += __uint_as_float(o[u] << 16);
                    acc[2 * u + 1] += __uint_as_float(o[u] & 0xffff0000u);
                }
            }
        }
    } else {
    // EB_HB rows' loads are issued before the first is used (clamped addresses, masked sums): one row per trip was a chain of HBM
    // round trips (25 us for 68 MB)
    for (int h0 = 0; h0 < EB_GS; h0 += EB_HB) {
        float x[EB_HB][8], c[EB_HB][8];
#pragma unroll
        for (int s = 0; s < EB_HB; ++s) {
            const int b = min(b0 + h0 + s, B - 1);
            const size_t i = (size_t)b * TE + col;
            io<T>::ld4(g + i, *reinterpret_cast<float(*)[4]>(&x[s][0]));
            io<T>::ld4(g + i + 4, *reinterpret_cast<float(*)[4]>(&x[s][4]));
            if (cls_row) {   // block-uniform for all but one block
                io<T>::ld4(gcls + (size_t)b * E + col, *reinterpret_cast<float(*)[4]>(&c[s][0]));
                io<T>::ld4(gcls + (size_t)b * E + col + 4, *reinterpret_cast<float(*)[4]>(&c[s][4]));
            }
        }
#pragma unroll
        for (int s = 0; s < EB_HB; ++s) {
            const int b = b0 + h0 + s;
            if (b >= B) break;
            const size_t i = (size_t)b * TE + col;
            if (cls_row) {
#pragma unroll
                for (int u = 0; u < 8; ++u) x[s][u] += c[s][u];
            }
            if (p > 0.0f) {
                const unsigned key = dropout_row_key(sd, (uint64_t)i >> 12);
                const unsigned c0 = (unsigned)(i & 4095);
#pragma unroll
                for (int u = 0; u < 8; ++u) x[s][u] *= dropout_scale(key, c0 + u, p, inv_keep);
            }
            if (dtok != nullptr) {
                io<T>::st4(dtok + i, *reinterpret_cast<const float(*)[4]>(&x[s][0]));
                io<T>::st4(dtok + i + 4, *reinterpret_cast<const float(*)[4]>(&x[s][4]));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += x[s][u];
        }
    }
    }
    float* po = partials + (size_t)blockIdx.y * TE + col;
    *reinterpret_cast<float4*>(po) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(po + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// dpos[t][e] = sum_groups partials; dcls[e] = dpos[0][e]: one column of the flat [T * E] row per thread, its `groups` partials as
// independent loads, 256-thread workgroups with few registers -- they run on CUs that the batched weight gradients occupy (one
// 1024-thread workgroup per 16 columns, E / 16 = 32 of them reading 64-byte half lines, took 73-104 us there and held up the embedding
// weight gradient behind it on the main stream; so did 1024-thread workgroups of any shape: 16 waves do not fit beside that kernel).
// dbias[e] = sum_{t >= 1} dpos[t][e] follows as a launch of its own (E / 64 single-wave workgroups over 133 KB that are still in L2).
constexpr int EF_T = 256;
__global__ __launch_bounds__(EF_T) void embed_bwd_fold_kernel(const float* __restrict__ partials, float* __restrict__ dpos, float* __restrict__ dcls,
                                                             int groups, int TE, int E) {
    const int c = blockIdx.x * EF_T + threadIdx.x;
    if (c >= TE) return;
    float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int g = 0;
    for (; g + 8 <= groups; g += 8) {
        float q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = partials[(size_t)(g + u) * TE + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u & 3] += q[u];
    }
    for (; g < groups; ++g) a[0] += partials[(size_t)g * TE + c];
    const float v = (a[0] + a[1]) + (a[2] + a[3]);
    dpos[c] = v;
    if (c < E) dcls[c] = v;
}
__global__ __launch_bounds__(64) void embed_bwd_bias_kernel(const float* __restrict__ dpos, float* __restrict__ dbias, int T, int E) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= E) return;
    float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int t = 1;
    for (; t + 8 <= T; t += 8) {
        float q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = dpos[(size_t)(t + u) * E + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u & 3] += q[u];
    }
    for (; t < T; ++t) a[0] += dpos[(size_t)t * E + e];
    dbias[e] = (a[0] + a[1]) + (a[2] + a[3]);
}

extern "C" int spv_embed_bwd_groups(int batch) { return cdiv(batch, EB_GS); }

extern "C" int spv_embed_bwd(const void* g, const void* gcls, void* dtok, float* partials, float* dpos, float* dbias, float* dcls, int batch,
                             int tokens, int embed, float p_drop, uint64_t seed, int dtype, void* stream) {
    SPV_CHECK(batch > 0 && tokens > 0 && embed > 0 && embed % 8 == 0, "spv_embed_bwd: bad shape %d x %d x %d (embed must be a multiple of 8)", batch, tokens, embed);
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_embed_bwd: bad dtype");
    SPV_CHECK(p_drop >= 0.0f && p_drop < 1.0f, "spv_embed_bwd: p_drop=%f", p_drop);
    SPV_CHECK(g && partials && dpos && dbias && dcls, "spv_embed_bwd: null pointer");
    SPV_CHECK(dtok != nullptr || (gcls == nullptr && p_drop == 0.0f), "spv_embed_bwd: dtok may be NULL only when it would equal g");
    SPV_CHECK((int64_t)tokens * embed < (1ll << 28), "spv_embed_bwd: row too long");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int TE = tokens * embed, groups = cdiv(batch, EB_GS);
    const dim3 grid(cdiv(TE / 8, 256), groups);
    if (dtype == SPV_BF16)
        hipLaunchKernelGGL((embed_bwd_rows_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)g, (const bf16_t*)gcls, (bf16_t*)dtok, partials, batch, TE,
                           embed, p_drop, seed);
    else
        hipLaunchKernelGGL((embed_bwd_rows_kernel<float>), grid, dim3(256), 0, st, (const float*)g, (const float*)gcls, (float*)dtok, partials, batch, TE, embed,
                           p_drop, seed);
    SPV_LAUNCH_CHECK("spv_embed_bwd(rows)");
    hipLaunchKernelGGL(embed_bwd_fold_kernel, dim3(cdiv(TE, EF_T)), dim3(EF_T), 0, st, partials, dpos, dcls, groups, TE, embed);
    hipLaunchKernelGGL(embed_bwd_bias_kernel, dim3(cdiv(embed, 64)), dim3(64), 0, st, dpos, dbias, tokens, embed);
    SPV_LAUNCH_CHECK("spv_embed_bwd(fold)");
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

extern "C" int spv_spectral_fold_bf16(const float* proj_w, const float* freq_h, const float* freq_w, float* w_full, void* w_full_bf16, int embed,
                                      int chans, int patch, void* stream) {
    SPV_CHECK(embed > 0 && chans > 0 && patch > 0 && w_full_bf16 != nullptr, "spv_spectral_fold_bf16: bad arguments");
    hipLaunchKernelGGL(spectral_fold_kernel, dim3(ew_blocks((int64_t)embed * chans * patch * patch)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), proj_w, freq_h, freq_w, w_full, embed, chans, patch, static_cast<bf16_t*>(w_full_bf16));
    SPV_LAUNCH_CHECK("spv_spectral_fold_bf16");
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
    int nt = 1024;   // rows of [patch + Pv] floats within 48 KiB of LDS
    while (nt > 64 && (size_t)nt * (patch + Pv) * sizeof(float) > 48 * 1024) nt >>= 1;
    SPV_CHECK((size_t)nt * (patch + Pv) * sizeof(float) <= 64 * 1024, "spv_spectral_fold_bwd: patch %d too large", patch);
    hipLaunchKernelGGL(freq_weight_grad_kernel, dim3(1), dim3(nt), (size_t)nt * (patch + Pv) * sizeof(float), st, scratch, freq_h,
                       freq_w, dfreq_h, dfreq_w, embed * chans, patch);
    SPV_LAUNCH_CHECK("spv_spectral_fold_bwd(freq)");
    return 0;
}

int spv_seed_ptr_set_patch(const unsigned long long* p) { return spv_seed_symbol_set(p); }
