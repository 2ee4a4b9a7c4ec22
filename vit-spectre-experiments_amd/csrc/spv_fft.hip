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

#include <stdlib.h>

namespace {

constexpr int FT = 512;  // threads per workgroup of the fast path

__device__ __forceinline__ float ld1(const void* base, size_t off, int bf) {
    return bf ? bf2f(static_cast<const bf16_t*>(base)[off]) : static_cast<const float*>(base)[off];
}
__device__ __forceinline__ void st1(void* base, size_t off, int bf, float v) {
    if (bf) static_cast<bf16_t*>(base)[off] = f2bf(v);
    else static_cast<float*>(base)[off] = v;
}
// store v (+ add_in[off] when a residual is folded into the mixer's output)
__device__ __forceinline__ void st1a(void* base, const void* add_in, size_t off, int bf, float v) {
    if (add_in != nullptr) v += ld1(add_in, off, bf);
    st1(base, off, bf, v);
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
                                                      int bf, const void* __restrict__ add_in) {
    extern __shared__ __attribute__((aligned(16))) float lds_f32[];
    float* lds = lds_f32;
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
                st1a(y, add_in, base + (size_t)M * D + k, bf, d);
                if (mm != M) st1a(y, add_in, base + (size_t)mm * D + k, bf, s);
                if (k != 0) {
                    st1a(y, add_in, base + (size_t)M * D + mk, bf, s);
                    if (mm != M) st1a(y, add_in, base + (size_t)mm * D + mk, bf, d);
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
        st1a(y, add_in, base + (size_t)M * D + Dh, bf, P);
        if (mm != M) st1a(y, add_in, base + (size_t)mm * D + Dh, bf, P);
    }
}

// ---------------- v2: bf16, dim = 512 -- the benchmark shape (B, 65, 512) ------------------------------------
// Two 512-thread workgroups per CU (LDS 72 KB each, everything bf16 in LDS, fp32 in registers):
//   A  coalesced 16-byte loads -> LDS rows (row stride 1024 + 64 B)
//   B  33 in-place radix-8 Stockham FFTs (two real rows per FFT), 64 threads each, twiddles in registers
//   C  token-axis DFT + spectrum unpacking as ONE pair of matrix products on the MFMA pipe:
//      Y1 = W1 . Z, Y2 = W2 . Z  (W: 32 x 80 bf16 per table, Z: the 66 planar FFT rows, fragments by
//      ds_read_b64_tr_b16 straight from the row-major tile), rows m >= 32 on the VALU with fp32 weights
//   D  y[m,k] = Y1[m,k] + Y2[m,D-k],  y[N-m,k] = Y1[m,D-k] + Y2[m,k] from a bf16 LDS stage, 16-byte stores.
// Formulation and bf16 error budget are checked on the host (tests/cpu_harness/fnet_v2_test.cpp).
// Diagnostic stamps (tools/fnet_stamps.hip defines SPV_FNET_STAMPS and provides the buffer); compiled out of the library.
#ifdef SPV_FNET_STAMPS
__device__ unsigned long long* g_fnet_stamps = nullptr;
#define V2_STAMP(i)                                                                                  \
    do {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if (g_fnet_stamps && (threadIdx.x & 63) == 0)                                                \
            g_fnet_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    } while (0)
#define V2_STAMP_RT(i)                                                                               \
    do {                                                                                             \
        if (g_fnet_stamps && (threadIdx.x & 63) == 0)                                                \
            g_fnet_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define V2_STAMP(i)
#define V2_STAMP_RT(i)
#endif
constexpr int V2D = 512;
constexpr int V2RS = V2D * 2 + 64;   // LDS row stride in bytes: 64 B of padding per row = a quarter of the 256-byte bank window, so that
                                     // the four rows (and the row pair two rows further) a transposing read touches cover disjoint banks
__device__ __forceinline__ int v2_off(int idx) { return 2 * idx; }  // byte offset of element idx in a row
constexpr int V2KS = 5;              // MFMA k-steps: 80 >= 2 * ceil(79 / 2) planar rows
constexpr int V2XR = 8;              // at most 8 rows m in [32, 40) go through the VALU path
typedef __attribute__((ext_vector_type(4))) short v2s16x4;
typedef __attribute__((ext_vector_type(8))) short v2s16x8;

__global__ __launch_bounds__(256) void fnet_v2_table_kernel(bf16_t* __restrict__ frag, float* __restrict__ extra, int N) {
    const int nh1 = N / 2 + 1;
    auto weight = [&](int t, int m, int j) -> float {
        if (m >= nh1 || j >= 2 * ((N + 1) / 2)) return 0.0f;
        const int n1 = j & ~1, n2 = n1 + 1;
        float c1 = 0.f, s1 = 0.f, c2 = 0.f, s2 = 0.f;
        if (n1 < N) sincospif(2.0f * (float)(((long long)m * n1) % N) / (float)N, &s1, &c1);
        if (n2 < N) sincospif(2.0f * (float)(((long long)m * n2) % N) / (float)N, &s2, &c2);
        float w1r, w1i, w2r, w2i;
        fnet_v2_weights(c1, s1, c2, s2, w1r, w1i, w2r, w2i);
        return t == 0 ? ((j & 1) ? w1i : w1r) : ((j & 1) ? w2i : w2r);
    };
    const int nfrag = 2 * V2KS * 64 * 8;
    const int nfragx = V2KS * 64 * 8;
    bf16_t* fragx = reinterpret_cast<bf16_t*>(extra);  // [ks][lane][8] bf16 in the (former fp32) extras region, same byte size
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nfrag + nfragx; i += gridDim.x * blockDim.x) {
        if (i < nfrag) {
            // fragment order of v_mfma_f32_32x32x16_bf16 (the A operand of W . Z == the B operand of Z^T . W^T): lane l holds
            // W[m = l & 31][j = 16 ks + 8 (l >> 5) + e]
            const int e = i & 7, l = (i >> 3) & 63, ks = (i >> 9) % V2KS, t = i / (V2KS * 512);
            frag[i] = f2bf(weight(t, l & 31, 16 * ks + 8 * (l >> 5) + e));
        } else {
            // third accumulator: column 0 carries W1[32], column 1 W2[32], the other 30 columns are zero
            const int q = i - nfrag;
            const int e = q & 7, l = (q >> 3) & 63, ks = q >> 9, col = l & 31;
            fragx[q] = f2bf(col < 2 ? weight(col, 32, 16 * ks + 8 * (l >> 5) + e) : 0.0f);
        }
    }
}

__device__ __forceinline__ float v2_ld(const bf16_t* p) { return bf2f(*p); }

// ---- phase B on the MFMA pipe: the 512-point FFT of a row pair z = x[2f] + i x[2f+1] as TWO matrix products --------------
// Index split n = 16 a + b (a < 32, b < 16), k = c + 32 d (c < 32, d < 16), w_n = exp(-2 pi i / n):
//     T[c,b]  = sum_a w_32^(a c) z[16 a + b]          stage 1: a 32-point DFT for every b        (64 x 64 real matrix B1)
//     T'[c,b] = T[c,b] w_512^(b c)                    twiddle: lane-local on the accumulators
//     Z[c+32d] = sum_b w_16^(b d) T'[c,b]             stage 2: a 16-point DFT for every c        (32 x 32 real matrix B2)
// Both products are taken in the transposed form (data = A operand, DFT matrix = B operand held in registers), so that
//   * stage 1 reads its fragments straight from the natural row layout with the transposing LDS read (rows of the [k][m] view
//     are the 16-element groups a of a tile row; 4 consecutive groups are 128 contiguous bytes),
//   * its accumulators hold, per lane, one c and runs of 4 consecutive b: T' leaves as 8-byte LDS stores into a [c][(part,b)]
//     image (64 B per c, 16-byte chunks XOR-swizzled by (c >> 2) & 3: conflict-free b128 reads by stage 2),
//   * stage 2's accumulators hold, per lane, one (part, d) and runs of 4 consecutive c = 4 consecutive frequencies k: the
//     spectrum goes back to the tile rows (Re -> row 2f, Im -> row 2f+1, natural order) as 8-byte stores.
// Everything is in place and WAVE-LOCAL: a wave owns whole row pairs (two per 32-row MFMA block), reads all fragments of a block
// before it overwrites those rows (one wave's LDS instructions execute in order), so phase B needs no workgroup barrier.
// 202 MFMAs per sample replace 3 x 33 radix-8 VALU passes (18 k of the kernel's 54 k cycles per sample in round 1).
// The DFT matrices are the same for every sample, so their bf16 rounding is a COHERENT error of the operator (0.2 % per entry): it
// does not average out over the batch the way the data's rounding does, and doubled the error of the deepest parameter gradients
// (rel-L2 1.1e-2 -> 2.3e-2 on the embedding's frequency weights at bs 512).  Each matrix is therefore kept as hi + lo bf16 parts
// (error 2^-17) and every product takes two MFMAs.
constexpr int V3_B1_FLOATS = 8 * 64 * 8 / 2;   // [part: hi, lo][trig: cos, sin][a half] fragments of 2 pi a c / 32 (x 64 lanes x 8 bf16)
constexpr int V3_B2_FLOATS = 4 * 64 * 8 / 2;   // [part: hi, lo][k-step]
constexpr int V3_FLOATS = V3_B1_FLOATS + V3_B2_FLOATS;

__global__ __launch_bounds__(256) void fnet_v3_table_kernel(bf16_t* __restrict__ b1, bf16_t* __restrict__ b2) {
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    // B-operand fragment order of v_mfma_f32_32x32x16_bf16: lane l holds B[k = 8 (l >> 5) + e][n = l & 31]
    // stage 1: w^(ac) (zr + i zi), w = cos - i sin: Re = cos zr + sin zi, Im = -sin zr + cos zi.  Four fragments are kept
    // ([trig: cos, sin][a half]); the -sin ones are the sin ones with the sign bits flipped.
    for (int i = i0; i < 8 * 64 * 8; i += stride) {
        const int e = i & 7, l = (i >> 3) & 63, ah = (i >> 9) & 1, trig = (i >> 10) & 1, lo = i >> 11;
        const int a = 16 * ah + 8 * (l >> 5) + e, c = l & 31;
        float sn, cs;
        sincospif(2.0f * (float)((a * c) & 31) / 32.0f, &sn, &cs);
        const float v = trig == 0 ? cs : sn;
        const bf16_t hi = f2bf(v);
        b1[i] = lo ? f2bf(v - bf2f(hi)) : hi;
    }
    for (int i = i0; i < 4 * 64 * 8; i += stride) {
        const int e = i & 7, l = (i >> 3) & 63, ks = (i >> 9) & 1, lo = i >> 10;
        const int part_in = ks, bb = 8 * (l >> 5) + e, part_out = (l & 31) >> 4, d = l & 15;
        float sn, cs;
        sincospif(2.0f * (float)((bb * d) & 15) / 16.0f, &sn, &cs);
        const float v = part_out == 0 ? (part_in == 0 ? cs : sn) : (part_in == 0 ? -sn : cs);
        const bf16_t hi = f2bf(v);
        b2[i] = lo ? f2bf(v - bf2f(hi)) : hi;
    }
}

__device__ __forceinline__ unsigned v2_pack(float lo, float hi) { return pack_bf16x2(lo, hi); }

struct V3Tab {          // stage 1: DFT-32 fragments + the twiddles applied to its accumulators
    uint4 b1[2][2][2];  // [hi / lo][cos / sin][a half]
    float twr[8], twi[8];
};
struct V3Tab2 {         // stage 2: DFT-16 fragments; asked for only once stage 1 is through (the two sets never live together)
    uint4 b2[2][2];     // [hi / lo][k-step]
};
__device__ __forceinline__ bf16x8 v3_frag(uint4 v, bool negate) {
    if (negate) v = make_uint4(v.x ^ 0x80008000u, v.y ^ 0x80008000u, v.z ^ 0x80008000u, v.w ^ 0x80008000u);
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ void v3_load_tables(V3Tab& t, const float* __restrict__ v3, int lane) {
    const uint4* b1 = reinterpret_cast<const uint4*>(v3);
#pragma unroll
    for (int lo = 0; lo < 2; ++lo) {
#pragma unroll
        for (int trig = 0; trig < 2; ++trig)
#pragma unroll
            for (int ah = 0; ah < 2; ++ah) t.b1[lo][trig][ah] = b1[((lo * 2 + trig) * 2 + ah) * 64 + lane];
    }
    // twiddle w_512^(b c) = cos - i sin of 2 pi b c / 512 for this lane's c = lane & 31 and its 8 values of b (accumulator register
    // j of a row pair -> b = (j & 3) + 8 (j >> 2) + 4 (lane >> 5)); v_sin / v_cos take revolutions
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int bb = (j & 3) + 8 * (j >> 2) + 4 * h;
        const float rev = (float)(bb * c) * (1.0f / 512.0f);
        t.twr[j] = __builtin_amdgcn_cosf(rev);
        t.twi[j] = -__builtin_amdgcn_sinf(rev);
    }
}

__device__ __forceinline__ void v3_load_tables2(V3Tab2& t, const float* __restrict__ v3, int lane) {
    const uint4* b2 = reinterpret_cast<const uint4*>(v3 + V3_B1_FLOATS);
#pragma unroll
    for (int lo = 0; lo < 2; ++lo)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) t.b2[lo][ks] = b2[(lo * 2 + ks) * 64 + lane];
}

// byte address of T'[c][k .. k+3] (k = 16 part + b, a multiple of 4) of row pair f inside its own two tile rows
__device__ __forceinline__ int v3_tp_addr(int f, int c, int k) {
    return (2 * f + (c >> 4)) * V2RS + (c & 15) * 64 + (((k >> 3) ^ ((c >> 2) & 3)) << 4) + ((k & 7) << 1);
}

// one 32-row block = row pairs f0 and f0 + 1 (`two` = the second one exists); the calling wave owns tile rows 2 f0 .. 2 f0 + 3
__device__ __forceinline__ void v3_stage1_block(unsigned char* lds, int f0, bool two, const V3Tab& t, int lane) {
    using lds_tr = v2s16x4 __attribute__((address_space(3)))*;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int fr = two ? (g & 1) : 0, kh = g >> 1;  // an absent second pair re-reads the first one's rows; its results are dropped
    f32x16 accR, accI;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accR[r] = 0.0f; accI[r] = 0.0f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int a0 = 16 * (ks & 1) + 8 * kh;
        const unsigned char* row = lds + (2 * (f0 + fr) + (ks >> 1)) * V2RS;
        const v2s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(row + v2_off(16 * (a0 + q) + 4 * pp)));
        const v2s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(row + v2_off(16 * (a0 + 4 + q) + 4 * pp)));
        const v2s16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const bf16x8 afr = __builtin_bit_cast(bf16x8, av);
        // k-step ks = 2 part_in + a half:  (re -> re) cos, (re -> im) -sin, (im -> re) sin, (im -> im) cos; hi + lo parts of each
        const int pin = ks >> 1, ah = ks & 1;
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) {
            accR = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, v3_frag(t.b1[lo][pin][ah], false), accR, 0, 0, 0);
            accI = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, v3_frag(t.b1[lo][1 - pin][ah], pin == 0), accI, 0, 0, 0);
        }
    }
    // twiddle + T' -> LDS.  lane: c = lane & 31, h = lane >> 5; register r: row m = (r & 3) + 8 (r >> 2) + 4 h of the block
    // = row pair r >> 3, b = (r & 3) + 8 ((r >> 2) & 1) + 4 h
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int fl = 0; fl < 2; ++fl) {
        if (fl == 1 && !two) break;
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            float re[4], im[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float tr = accR[8 * fl + 4 * grp + j], ti = accI[8 * fl + 4 * grp + j];
                const float wr = t.twr[4 * grp + j], wi = t.twi[4 * grp + j];
                re[j] = tr * wr - ti * wi;
                im[j] = tr * wi + ti * wr;
            }
            const int b0 = 8 * grp + 4 * h;
            *reinterpret_cast<uint2*>(lds + v3_tp_addr(f0 + fl, c, b0)) = make_uint2(v2_pack(re[0], re[1]), v2_pack(re[2], re[3]));
            *reinterpret_cast<uint2*>(lds + v3_tp_addr(f0 + fl, c, 16 + b0)) = make_uint2(v2_pack(im[0], im[1]), v2_pack(im[2], im[3]));
        }
    }
}

__device__ __forceinline__ void v3_stage2_block(unsigned char* lds, int f0, bool two, const V3Tab2& t, int lane) {
    const int c = lane & 31, h = lane >> 5;
    // stage 2 per row pair: T'[c][k = (part, b)] (16 bytes per lane and k-step) is the B operand here and the 16-point DFT matrix the
    // A operand, so that a lane owns ONE c = one residue of the frequency k = c + 32 d and its registers run over (part, d): the
    // 32 lanes of a register write 32 CONSECUTIVE frequencies (64 contiguous bytes of a tile row).  (In the transposed form a lane
    // owned one d and the 16 lanes of an 8-byte store hit two bank positions: 8-way conflicts, 58 % of all LDS cycles of the kernel.)
#pragma unroll
    for (int fl = 0; fl < 2; ++fl) {
        if (fl == 1 && !two) break;
        const int f = f0 + fl;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint4 tv = *reinterpret_cast<const uint4*>(lds + v3_tp_addr(f, c, 16 * ks + 8 * h));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, t.b2[0][ks]), __builtin_bit_cast(bf16x8, tv), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, t.b2[1][ks]), __builtin_bit_cast(bf16x8, tv), acc, 0, 0, 0);
        }
        // register r: row m = (r & 3) + 8 (r >> 2) + 4 h of the product = (part = m >> 4, d = m & 15); column = c; frequency k = c + 32 d.
        // Neighbouring lanes hold neighbouring frequencies: even lanes take over the odd neighbour's value of registers 0-7 (Re), odd
        // lanes the even neighbour's of registers 8-15 (Im), so every lane leaves with 8 packed pairs: 8 four-byte stores, not 16 two-byte
        bf16_t* orow = reinterpret_cast<bf16_t*>(lds + (2 * f) * V2RS);
        const bool odd = lane & 1;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            // quad_perm [1,0,3,2]: the value of the lane's neighbour
            const float nre = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[r]), 0xB1, 0xF, 0xF, true));
            const float nim = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[8 + r]), 0xB1, 0xF, 0xF, true));
            const int d = (r & 3) + 8 * ((r >> 2) & 1) + 4 * h;
            // even lane: Re pair (own c, neighbour c + 1) -> row 2 f; odd lane: Im pair (neighbour c - 1, own c) -> row 2 f + 1
            const unsigned pr = odd ? pack_bf16x2(nim, acc[8 + r]) : pack_bf16x2(acc[r], nre);
            unsigned char* dst = reinterpret_cast<unsigned char*>(orow) + (odd ? V2RS : 0) + v2_off(32 * d + (c & ~1));
            *reinterpret_cast<unsigned*>(dst) = pr;
        }
    }
}

// LayerNorm-1 of the encoder layer fused into the mixer (reference spectre.py:66: norm1(mix(x)) + x):
//   FUSE = 1 (forward): phase D normalises each finished row (a row = 64 consecutive chunks = one wave: statistics by wave
//            reductions), adds the residual and writes x1; the pre-norm row is written too (the backward needs it).
//   FUSE = 2 (backward): phase A turns the incoming gradient into the gradient of the pre-norm mixer output on the fly
//            (LayerNorm backward per row) while filling LDS; the column sums (dgamma, dbeta) leave as one slab per
//            workgroup.  Saves the add+LayerNorm kernels and one read + one write of a (B, N, D) tensor each way.
struct FnetLn {
    const float* gamma;
    const float* beta;
    float* mean;            // [batch * N]  (written by FUSE 1, read by FUSE 2)
    float* rstd;
    const bf16_t* res;      // FUSE 1: residual input (the mixer's own input x)
    bf16_t* prenorm;        // FUSE 1: pre-norm mixer output, saved for the backward
    const bf16_t* m_in;     // FUSE 2: that saved tensor
    float* partials;        // FUSE 2: [batch][2][D] column sums of this workgroup
};
constexpr float V2_LN_EPS = 1e-5f;

constexpr int V2IT = 9;                  // row-chunk iterations of the 512 threads over 65 rows x 64 chunks of 16 B
constexpr int V2YRS = V2D * 2 + 16;      // row stride of the Y stage (phase C -> D): 260 dwords = 4 mod 32 banks between rows

template <int FUSE>
__global__ __launch_bounds__(512, 4) void fnet_mfma_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                           const uint4* __restrict__ wfrag, const uint4* __restrict__ wfragx,
                                                           const float* __restrict__ v3tab, int N, int stagger,
                                                           const bf16_t* __restrict__ add_in, FnetLn ln) {
    extern __shared__ __attribute__((aligned(16))) float lds_f32[];
    unsigned char* lds = reinterpret_cast<unsigned char*>(lds_f32);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t base = (size_t)blockIdx.x * N * V2D;
    const int NF = (N + 1) >> 1, JR = 2 * NF, nh1 = N / 2 + 1;

    V2_STAMP_RT(10);
    // Phase stagger: with batch = 2 x CUs every CU gets exactly two workgroups and, launched together, they would all
    // load, then all compute, then all store -- HBM idle while the VALUs work and vice versa.  The second half of the
    // grid (the second workgroup of each CU under in-order dispatch; a performance assumption only) sleeps through the
    // first half's load phase, so its loads overlap their FFTs and its FFTs overlap their stores.
    if (stagger > 0) {
        // which workgroups wait: the second half of the grid (bit 8 clear), or -- bit 8 set -- the workgroups in an ODD threadgroup
        // slot of their CU (HW_REG_HW_ID bits 19:16, TG_ID): with two resident workgroups per CU exactly one of each pair,
        // whatever the dispatcher's block -> CU placement (a performance heuristic only: any outcome is correct)
        const bool second = (stagger & 256) ? (__builtin_amdgcn_s_getreg((3 << 11) | (16 << 6) | 4) & 1) != 0 : blockIdx.x >= (gridDim.x >> 1);
        if (second)
            for (int i = 0; i < (stagger & 255); ++i) __builtin_amdgcn_s_sleep(127);
    }
    V2_STAMP(0);
    // the DFT-matrix fragments of phase B: requested first, so that they arrive under the tile's own loads
    V3Tab tab;
    if (FUSE != 2) v3_load_tables(tab, v3tab, lane);  // (the backward's staging loop needs the registers: it asks afterwards)
    // ---- A: x -> LDS (bf16), pad row of an odd N zeroed.  Thread tid owns chunk tid & 63 of rows (tid >> 6) + 8 i: all of an
    // iteration batch's loads are issued before the first use (a dependent load per loop trip cost a memory round trip each)
    const int ch = tid & 63;
    if (FUSE == 2) {
        // x is the gradient wrt x1 = LN(m) + x0: d m = rstd (dy g - mean(dy g) - xhat mean(dy g xhat)); one row per wave step
        float gam[8], accg[8], accb[8];
        {
            const float4 g0 = *reinterpret_cast<const float4*>(ln.gamma + ch * 8), g1 = *reinterpret_cast<const float4*>(ln.gamma + ch * 8 + 4);
            gam[0] = g0.x; gam[1] = g0.y; gam[2] = g0.z; gam[3] = g0.w; gam[4] = g1.x; gam[5] = g1.y; gam[6] = g1.z; gam[7] = g1.w;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { accg[u] = 0.0f; accb[u] = 0.0f; }
#pragma unroll
        for (int i0 = 0; i0 < V2IT; i0 += 3) {
            uint4 dvv[3], mvv[3];
            float mn[3], rs[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int row = min(wave + 8 * (i0 + i), N - 1);  // rows past N: a valid address, the value is not used
                // (uniform sample base + a 32-bit lane offset: the 64-bit per-lane address arithmetic of `ptr + base + row * V2D` was
                // ~130 of the kernel's ~2 200 instructions, and the kernel is instruction-issue bound)
                const unsigned off = (unsigned)row * V2D + (unsigned)ch * 8;
                dvv[i] = *reinterpret_cast<const uint4*>(x + base + off);
                mvv[i] = *reinterpret_cast<const uint4*>(ln.m_in + base + off);
                mn[i] = (ln.mean + (size_t)blockIdx.x * N)[(unsigned)row];
                rs[i] = (ln.rstd + (size_t)blockIdx.x * N)[(unsigned)row];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int row = wave + 8 * (i0 + i);
                if (row >= N) continue;
                const uint4 dv = dvv[i], mv = mvv[i];
                const float mean = mn[i], rstd = rs[i];
                const unsigned dw[4] = {dv.x, dv.y, dv.z, dv.w}, mw[4] = {mv.x, mv.y, mv.z, mv.w};
                float xh[8], t[8];
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float d0 = __uint_as_float(dw[u] << 16), d1 = __uint_as_float(dw[u] & 0xffff0000u);
                    xh[2 * u] = (__uint_as_float(mw[u] << 16) - mean) * rstd;
                    xh[2 * u + 1] = (__uint_as_float(mw[u] & 0xffff0000u) - mean) * rstd;
                    accg[2 * u] += d0 * xh[2 * u]; accg[2 * u + 1] += d1 * xh[2 * u + 1];
                    accb[2 * u] += d0; accb[2 * u + 1] += d1;
                    t[2 * u] = d0 * gam[2 * u]; t[2 * u + 1] = d1 * gam[2 * u + 1];
                    s1 += t[2 * u] + t[2 * u + 1];
                    s2 += t[2 * u] * xh[2 * u] + t[2 * u + 1] * xh[2 * u + 1];
                }
                const float m1 = wave_sum(s1) * (1.0f / V2D), m2 = wave_sum(s2) * (1.0f / V2D);
                unsigned o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    o[u] = pack_bf16x2(rstd * (t[2 * u] - m1 - xh[2 * u] * m2), rstd * (t[2 * u + 1] - m1 - xh[2 * u + 1] * m2));
                *reinterpret_cast<uint4*>(lds + row * V2RS + v2_off(ch * 8)) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
        v3_load_tables(tab, v3tab, lane);  // arrive under the column-sum rounds below
        // column sums: the 8 waves hold the same 512 columns; meet in the 4 KiB behind the tile, in wave order
        float* red = reinterpret_cast<float*>(lds + (size_t)std::max(2 * ((N + 1) / 2), 2 * (N / 2 + 1)) * V2RS);
        for (int w = 0; w < 8; ++w) {
            if (wave == w) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    red[ch * 8 + u] = (w == 0 ? 0.0f : red[ch * 8 + u]) + accg[u];
                    red[V2D + ch * 8 + u] = (w == 0 ? 0.0f : red[V2D + ch * 8 + u]) + accb[u];
                }
            }
            __syncthreads();
        }
        for (int c = tid; c < 2 * V2D; c += 512) ln.partials[(size_t)blockIdx.x * 2 * V2D + c] = red[c];
    } else {
        uint4 xv[V2IT];
#pragma unroll
        for (int i = 0; i < V2IT; ++i) {
            const int row = min(wave + 8 * i, N - 1);  // rows past N: a valid address, the value is not stored
            xv[i] = *reinterpret_cast<const uint4*>(x + base + ((unsigned)row * V2D + (unsigned)ch * 8));
        }
#pragma unroll
        for (int i = 0; i < V2IT; ++i) {
            const int row = wave + 8 * i;
            if (row < N) *reinterpret_cast<uint4*>(lds + row * V2RS + v2_off(ch * 8)) = xv[i];
        }
    }
    if (N & 1)
        for (int c = tid; c < 64; c += 512) *reinterpret_cast<uint4*>(lds + N * V2RS + v2_off(c * 8)) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    V2_STAMP(1);

    // ---- B: row-pair FFTs on the MFMA pipe, two row pairs per block, blocks dealt round-robin to the waves (wave-local)
    {
        const int nblk = (NF + 1) >> 1;
        for (int blk = wave; blk < nblk; blk += 8) v3_stage1_block(lds, 2 * blk, 2 * blk + 1 < NF, tab, lane);
        V3Tab2 tab2;
        v3_load_tables2(tab2, v3tab, lane);
        for (int blk = wave; blk < nblk; blk += 8) v3_stage2_block(lds, 2 * blk, 2 * blk + 1 < NF, tab2, lane);
    }
    V2_STAMP(2);
    __syncthreads();  // all FFT rows are final before any wave reads them as MFMA operands
    V2_STAMP(3);

    // ---- C: Y1 = W1 . Z, Y2 = W2 . Z on the MFMA pipe, TRANSPOSED (the tile is the A operand, W^T the B operand): this wave owns
    // columns [64 wave, 64 wave + 64); an accumulator then holds, per lane, one output row m = lane & 31 and runs of 4 consecutive
    // columns, which leave as 8-byte LDS stores.  Row m = 32 (the 33rd of N = 65) rides in ONE more accumulator for both column
    // blocks: its B fragment carries W1[32], W2[32] in columns 0, 1 for block 0 and -- the same registers moved two lanes up -- in
    // columns 2, 3 for block 1 (round 1: 80 LDS reads + 160 FMAs per thread on the VALU).
    f32x16 acc[2][2], accx;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][0][r] = 0.0f; acc[0][1][r] = 0.0f; acc[1][0][r] = 0.0f; acc[1][1][r] = 0.0f; accx[r] = 0.0f; }
    const bool extra = nh1 > 32;
    {
        using lds_ptr = v2s16x4 __attribute__((address_space(3)))*;
        const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
        const int colb = v2_off(wave * 64 + 16 * (g & 1) + 4 * pp);  // byte offset of this lane's 4 columns inside a row
#pragma unroll
        for (int ks = 0; ks < V2KS; ++ks) {
            const bf16x8 w1 = __builtin_bit_cast(bf16x8, wfrag[(0 * V2KS + ks) * 64 + lane]);
            const bf16x8 w2 = __builtin_bit_cast(bf16x8, wfrag[(1 * V2KS + ks) * 64 + lane]);
            bf16x8 wx[2];
            {
                const uint4 wa = wfragx[ks * 64 + lane];
                wx[0] = __builtin_bit_cast(bf16x8, wa);
                uint4 wb;  // row_shr:2 with zero fill: lanes 2, 3 (34, 35) take over lanes 0, 1 (32, 33); everything else is zero either way
                wb.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)wa.x, 0x112, 0xF, 0xF, true);
                wb.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)wa.y, 0x112, 0xF, 0xF, true);
                wb.z = (unsigned)__builtin_amdgcn_update_dpp(0, (int)wa.z, 0x112, 0xF, 0xF, true);
                wb.w = (unsigned)__builtin_amdgcn_update_dpp(0, (int)wa.w, 0x112, 0xF, 0xF, true);
                wx[1] = __builtin_bit_cast(bf16x8, wb);
            }
            const int j_lo = min(16 * ks + 8 * (g >> 1) + q, JR - 1);      // rows past JR carry zero weights: clamp
            const int j_hi = min(16 * ks + 8 * (g >> 1) + q + 4, JR - 1);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const v2s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lds + j_lo * V2RS + colb + cb * 64));
                const v2s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lds + j_hi * V2RS + colb + cb * 64));
                const v2s16x8 bv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const bf16x8 zf = __builtin_bit_cast(bf16x8, bv);
                acc[cb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf, w1, acc[cb][0], 0, 0, 0);
                acc[cb][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf, w2, acc[cb][1], 0, 0, 0);
                accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf, wx[cb], accx, 0, 0, 0);  // (zero weights when there is no row 32)
            }
        }
    }
    V2_STAMP(4);
    V2_STAMP(5);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every read of Z is done: its LDS becomes the Y stage (Y1 rows [0,nh1), Y2 rows [nh1, 2 nh1))
    V2_STAMP(6);

    {
        // accumulator register r of lane (h = lane >> 5, m = lane & 31): column (r & 3) + 8 (r >> 2) + 4 h of the 32-column block
        const int h = lane >> 5, m = lane & 31;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int colbyte = (wave * 64 + cb * 32 + 4 * h) * 2;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (m < nh1) {
                    unsigned char* yr = lds + (t * nh1 + m) * V2YRS + colbyte;
#pragma unroll
                    for (int grp = 0; grp < 4; ++grp)
                        *reinterpret_cast<uint2*>(yr + 16 * grp) =
                            make_uint2(v2_pack(acc[cb][t][4 * grp], acc[cb][t][4 * grp + 1]), v2_pack(acc[cb][t][4 * grp + 2], acc[cb][t][4 * grp + 3]));
                }
            }
        }
        if (extra && m < 4) {  // columns 0, 1 of the fifth accumulator = Y1[32], Y2[32] of column block 0; columns 2, 3 of block 1
            unsigned char* yr = lds + ((m & 1) * nh1 + 32) * V2YRS + (wave * 64 + (m >> 1) * 32 + 4 * h) * 2;
#pragma unroll
            for (int grp = 0; grp < 4; ++grp)
                *reinterpret_cast<uint2*>(yr + 16 * grp) =
                    make_uint2(v2_pack(accx[4 * grp], accx[4 * grp + 1]), v2_pack(accx[4 * grp + 2], accx[4 * grp + 3]));
        }
    }
    V2_STAMP(7);
    // the residual rows of phase D (forward: the mixer's own input; backward / plain: add_in) are requested here, all at once, as
    // soon as the accumulators are dead (asked for while they were live, the loads pushed the kernel over its 128 registers: 26-48
    // spilled VGPRs = 54 MB of scratch traffic per launch).  The barrier below waits for LDS only (s_waitcnt lgkmcnt + s_barrier,
    // not __syncthreads, whose fence would also wait for these loads); a load per loop trip was a memory round trip per row
    const bf16_t* resp = FUSE == 1 ? ln.res : add_in;
    // opaque copies of the sample's offset and this lane's chunk: the forward's residual IS the mixer input, so hipcc otherwise keeps
    // phase A's nine 64-bit row addresses alive across phases B and C for these loads and the stores of phase D -- 26 spilled VGPRs,
    // 54 MB of scratch traffic per launch (FETCH/WRITE_SIZE: 184 MB against 102 MB algorithmic)
    size_t base_d = base;
    int ch_d = ch, wave_d = wave;
    asm volatile("" : "+s"(base_d), "+v"(ch_d), "+v"(wave_d));
    uint4 rv[V2IT];
#pragma unroll
    for (int i = 0; i < V2IT; ++i) rv[i] = make_uint4(0, 0, 0, 0);  // bf16 zeros (plain mixer call: nothing to add)
    if (resp != nullptr) {
#pragma unroll
        for (int i = 0; i < V2IT; ++i) {
            const int r = min(wave_d + 8 * i, N - 1);
            rv[i] = *reinterpret_cast<const uint4*>(resp + base_d + ((unsigned)r * V2D + (unsigned)ch_d * 8));
        }
    }
    float gg[8], bb[8];
    if (FUSE == 1) {
        const int k0 = ch_d * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(ln.gamma + k0), g1 = *reinterpret_cast<const float4*>(ln.gamma + k0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(ln.beta + k0), b1 = *reinterpret_cast<const float4*>(ln.beta + k0 + 4);
        gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w; gg[4] = g1.x; gg[5] = g1.y; gg[6] = g1.z; gg[7] = g1.w;
        bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    V2_STAMP(8);

    // ---- D: combine mirrored halves, 8 consecutive frequencies per lane, 16-byte stores; row r = wave + 8 i is this wave's
#pragma unroll
    for (int i = 0; i < V2IT; ++i) {
        const int r = wave_d + 8 * i, k0 = ch_d * 8;
        if (r >= N) continue;
        const bool low = 2 * r <= N;           // r <= N / 2
        const int m = low ? r : N - r;
        const bf16_t* fwd = reinterpret_cast<const bf16_t*>(lds + ((low ? 0 : 1) * nh1 + m) * V2YRS);  // read at k
        const bf16_t* mir = reinterpret_cast<const bf16_t*>(lds + ((low ? 1 : 0) * nh1 + m) * V2YRS);  // read at D - k
        const uint4 fv = *reinterpret_cast<const uint4*>(fwd + k0);
        const unsigned fw[4] = {fv.x, fv.y, fv.z, fv.w};
        // the mirror partners of k0 .. k0 + 7 are D - k0, D - k0 - 1, ..., D - k0 - 7: element 0 of chunk (64 - ch) & 63, then elements
        // 7 .. 1 of chunk 63 - ch -- one 16-byte and one 2-byte LDS read instead of eight 2-byte ones with their index arithmetic
        const uint4 mv = *reinterpret_cast<const uint4*>(mir + 8 * (63 - ch_d));
        const float mf[8] = {v2_ld(mir + ((V2D - k0) & (V2D - 1))), __uint_as_float(mv.w & 0xffff0000u), __uint_as_float(mv.w << 16),
                             __uint_as_float(mv.z & 0xffff0000u), __uint_as_float(mv.z << 16), __uint_as_float(mv.y & 0xffff0000u),
                             __uint_as_float(mv.y << 16), __uint_as_float(mv.x & 0xffff0000u)};
        const unsigned aw[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
        unsigned o[4];
        if (FUSE == 1) {
            // the row (64 chunks) is this wave's: LayerNorm over the bf16-rounded mixer output (exactly what the unfused
            // add+LayerNorm kernel would read back), then + residual
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a0 = __uint_as_float(fw[u] << 16) + mf[2 * u];
                const float a1 = __uint_as_float(fw[u] & 0xffff0000u) + mf[2 * u + 1];
                o[u] = pack_bf16x2(a0, a1);
            }
            *reinterpret_cast<uint4*>(ln.prenorm + base_d + ((unsigned)r * V2D + (unsigned)k0)) = make_uint4(o[0], o[1], o[2], o[3]);
            float v[8];
            float sm = 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[2 * u] = __uint_as_float(o[u] << 16);
                v[2 * u + 1] = __uint_as_float(o[u] & 0xffff0000u);
                sm += v[2 * u] + v[2 * u + 1];
            }
            const float mean = wave_sum(sm) * (1.0f / V2D);
            float sq = 0.0f;
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float d = v[u] - mean; sq += d * d; }
            const float rstd = rsqrtf(wave_sum(sq) * (1.0f / V2D) + V2_LN_EPS);
            if (lane == 0) { (ln.mean + (size_t)blockIdx.x * N)[(unsigned)r] = mean; (ln.rstd + (size_t)blockIdx.x * N)[(unsigned)r] = rstd; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float e0 = (v[2 * u] - mean) * rstd * gg[2 * u] + bb[2 * u] + __uint_as_float(aw[u] << 16);
                const float e1 = (v[2 * u + 1] - mean) * rstd * gg[2 * u + 1] + bb[2 * u + 1] + __uint_as_float(aw[u] & 0xffff0000u);
                o[u] = pack_bf16x2(e0, e1);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a0 = __uint_as_float(fw[u] << 16) + mf[2 * u] + __uint_as_float(aw[u] << 16);
                const float a1 = __uint_as_float(fw[u] & 0xffff0000u) + mf[2 * u + 1] + __uint_as_float(aw[u] & 0xffff0000u);
                o[u] = pack_bf16x2(a0, a1);
            }
        }
        *reinterpret_cast<uint4*>(y + base_d + ((unsigned)r * V2D + (unsigned)k0)) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    V2_STAMP(9);
    V2_STAMP_RT(11);
}

// dgamma / dbeta of the fused backward: out[p][c] = sum over the per-sample slabs [batch][2][V2D], in the order every fold of this
// library uses (spv_common.h fold_partials_block), so that a deferred fold (spv_reduce_multi) gives the same bits
__global__ __launch_bounds__(FOLD_COLS * FOLD_ROWS) void fnet_ln_fold_kernel(const float* __restrict__ partials, float* __restrict__ dgamma,
                                                                            float* __restrict__ dbeta, int parts) {
    const FoldJob j{partials, {dgamma, dbeta, nullptr, nullptr, nullptr}, parts, 2, V2D};
    fold_partials_block(j, blockIdx.x, threadIdx.x);
}

// ---------------- generic fallback (any tokens, dim): two direct-DFT kernels through an fp32 workspace
// stage 1: ws[row][k] = (A, B) = (sum_d x cos(2 pi k d/D), sum_d x sin(..)), one workgroup per row
__global__ __launch_bounds__(256) void fnet_generic_stage1(const void* __restrict__ x, float2* __restrict__ ws, int D, int bf) {
    extern __shared__ __attribute__((aligned(16))) float lds_f32[];
    float* lds = lds_f32;
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
__global__ __launch_bounds__(256) void fnet_generic_stage2(const float2* __restrict__ ws, void* __restrict__ y, int N, int D, int bf,
                                                           const void* __restrict__ add_in) {
    extern __shared__ __attribute__((aligned(16))) float lds_f32[];
    float* lds = lds_f32;
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
        st1a(y, add_in, base + (size_t)m * D + k, bf, acc);
    }
}

// rfft(x).real and its adjoint; one workgroup per row
__global__ __launch_bounds__(256) void rfft_real_kernel(const void* __restrict__ x, void* __restrict__ y, int D, int transpose, int bf) {
    extern __shared__ __attribute__((aligned(16))) float lds_f32[];
    float* lds = lds_f32;
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
                                                         int D, int axis_tokens, int len, int inverse, int bf, int zero_mode) {
    const float r = 0.70710678118654752440f;
    const float odd_scale = zero_mode ? r : 1.0f;
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
            else if (p < la) v = at(len - 1) * odd_scale;   // the unpaired last element: copied (pass-through) or paired with a zero
            else { const int q = p - la; v = (at(2 * q) - at(2 * q + 1)) * r; }
        } else {
            if ((len & 1) && p == len - 1) v = at(la - 1) * odd_scale;
            else {
                const int q = p >> 1;
                const float a = at(q), dd = at(la + q);
                v = ((p & 1) ? a - dd : a + dd) * r;
            }
        }
        st1(dst, e, bf, v);
    }
}

// The same level along the embedding axis for bf16 rows whose transformed length is a multiple of 16: a thread owns 8 consecutive
// elements (one 16-byte access; forward: four pairs -> four averages + four differences, two 8-byte stores; inverse the other way
// round; positions >= len are copied).  The element-per-thread kernel above does three integer divisions and two 2-byte loads per
// 2-byte store: 51 us for the 68 MB of a Small layer (0.17 of HBM peak).
template <bool INV>
__global__ __launch_bounds__(256) void haar_dim_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows, int D, int len) {
    const float r = 0.70710678118654752440f;
    const int cpr = D >> 3, la = len >> 1;   // 8-element chunks per row; len is even: la = lh = len / 2
    const int64_t total = rows * cpr;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / cpr;
        const int c = (int)(e - row * cpr);
        const bf16_t* sr = src + row * D;
        bf16_t* dr = dst + row * D;
        if (c * 8 >= len) {
            *reinterpret_cast<uint4*>(dr + c * 8) = *reinterpret_cast<const uint4*>(sr + c * 8);
        } else if (!INV) {
            const uint4 t = *reinterpret_cast<const uint4*>(sr + c * 8);
            const unsigned w[4] = {t.x, t.y, t.z, t.w};
            float av[4], dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float x0 = __uint_as_float(w[u] << 16), x1 = __uint_as_float(w[u] & 0xffff0000u);
                av[u] = (x0 + x1) * r;
                dv[u] = (x0 - x1) * r;
            }
            uint2 oa, od;
            oa.x = pack_bf16x2(av[0], av[1]); oa.y = pack_bf16x2(av[2], av[3]);
            od.x = pack_bf16x2(dv[0], dv[1]); od.y = pack_bf16x2(dv[2], dv[3]);
            *reinterpret_cast<uint2*>(dr + c * 4) = oa;
            *reinterpret_cast<uint2*>(dr + la + c * 4) = od;
        } else {
            const uint2 ta = *reinterpret_cast<const uint2*>(sr + c * 4), td = *reinterpret_cast<const uint2*>(sr + la + c * 4);
            const unsigned wa[2] = {ta.x, ta.y}, wd[2] = {td.x, td.y};
            float o[8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float a0 = __uint_as_float(wa[u] << 16), a1 = __uint_as_float(wa[u] & 0xffff0000u);
                const float d0 = __uint_as_float(wd[u] << 16), d1 = __uint_as_float(wd[u] & 0xffff0000u);
                o[4 * u + 0] = (a0 + d0) * r; o[4 * u + 1] = (a0 - d0) * r;
                o[4 * u + 2] = (a1 + d1) * r; o[4 * u + 3] = (a1 - d1) * r;
            }
            uint4 t;
            t.x = pack_bf16x2(o[0], o[1]); t.y = pack_bf16x2(o[2], o[3]); t.z = pack_bf16x2(o[4], o[5]); t.w = pack_bf16x2(o[6], o[7]);
            *reinterpret_cast<uint4*>(dr + c * 8) = t;
        }
    }
}

}  // namespace

// table buffer: [v1 token twiddles: (tokens+1)*2*FNET_TWS floats][v2 MFMA weight fragments: 2*5*64*8 bf16][v2 fp32 rows m >= 32]
static int64_t v2_frag_off(int tokens) { return (int64_t)(tokens + 1) * 2 * FNET_TWS; }
static int64_t v2_extra_off(int tokens) { return v2_frag_off(tokens) + 2 * V2KS * 64 * 8 / 2; }
static int64_t v2_tw_off(int tokens) { return v2_extra_off(tokens) + V2XR * 2 * 16 * V2KS; }
extern "C" int64_t spv_fnet_twiddle_floats(int tokens) { return v2_tw_off(tokens) + V3_FLOATS; }

extern "C" int spv_fnet_make_twiddle(float* tw, int tokens, void* stream) {
    SPV_CHECK(tokens > 0, "spv_fnet_make_twiddle: tokens=%d", tokens);
    hipLaunchKernelGGL(fnet_twiddle_kernel, dim3(cdiv((tokens + 1) * 2 * FNET_TWS, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), tw, tokens);
    SPV_LAUNCH_CHECK("spv_fnet_make_twiddle");
    hipLaunchKernelGGL(fnet_v2_table_kernel, dim3(32), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<bf16_t*>(tw + v2_frag_off(tokens)), tw + v2_extra_off(tokens), tokens);
    SPV_LAUNCH_CHECK("spv_fnet_make_twiddle(v2)");
    float* v3 = tw + v2_tw_off(tokens);
    hipLaunchKernelGGL(fnet_v3_table_kernel, dim3(16), dim3(256), 0, static_cast<hipStream_t>(stream), reinterpret_cast<bf16_t*>(v3),
                       reinterpret_cast<bf16_t*>(v3 + V3_B1_FLOATS));
    SPV_LAUNCH_CHECK("spv_fnet_make_twiddle(v3)");
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

extern "C" int spv_fnet_mix(const void* x, void* y, const void* add_in, const float* twiddle, int batch, int tokens, int dim,
                            int dtype, float* workspace, void* stream) {
    SPV_CHECK(batch > 0 && tokens > 0 && dim > 0, "spv_fnet_mix: empty");
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_fnet_mix: bad dtype %d", dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int bf = dtype == SPV_BF16;
    static const bool no_v2 = SPV_LAB_SET("SPV_FNET_NO_V2");  // tuning / A-B aid
    if (bf && dim == V2D && tokens >= 2 && tokens <= 65 && !no_v2) {
        SPV_CHECK(twiddle != nullptr, "spv_fnet_mix: twiddle table required");
        static const int stagger_env = SPV_LAB_INT("SPV_FNET_STAGGER", -1);
        const int v2_stagger = stagger_env >= 0 ? stagger_env : (batch >= 512 ? 1 : 0);  // x 8128 cycles (~3.5 us)
        const int rows = std::max(2 * ((tokens + 1) / 2), 2 * (tokens / 2 + 1));
        const size_t lds = (size_t)rows * V2RS;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fnet_mfma_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        SPV_COUNT_PATH(SPV_PATH_FNET_MFMA);
    hipLaunchKernelGGL(fnet_mfma_kernel<0>, dim3(batch), dim3(512), lds, st, static_cast<const bf16_t*>(x), static_cast<bf16_t*>(y),
                           reinterpret_cast<const uint4*>(twiddle + v2_frag_off(tokens)), reinterpret_cast<const uint4*>(twiddle + v2_extra_off(tokens)),
                           twiddle + v2_tw_off(tokens), tokens, v2_stagger, static_cast<const bf16_t*>(add_in), FnetLn{});
        SPV_LAUNCH_CHECK("spv_fnet_mix(v2)");
        return 0;
    }
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
            hipLaunchKernelGGL((fnet_lds_kernel<MHV>), dim3(batch), dim3(FT), lds, st, x, y, twiddle, tokens, dim, log2tpf, plan, bf, add_in); \
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
                       reinterpret_cast<const float2*>(workspace), y, tokens, dim, bf, add_in);
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
    SPV_CHECK(inverse >= 0 && inverse <= 3, "spv_haar_dwt: inverse=%d (bit 0: adjoint, bit 1: 'zero' extension of odd lengths)", inverse);
    const int zero_mode = (inverse >> 1) & 1;
    inverse &= 1;
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
        static const bool generic = SPV_LAB_SET("SPV_HAAR_GENERIC");   // A/B switch
        if (!generic && axis == 2 && dtype == SPV_BF16 && dim % 8 == 0 && lens[l] % 16 == 0 &&
            (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
            const int64_t rows = (int64_t)batch * tokens;
            const int g2 = (int)std::min<int64_t>((rows * (dim / 8) + 255) / 256, 8192);
            if (inverse) hipLaunchKernelGGL((haar_dim_bf16_kernel<true>), dim3(g2), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, rows, dim, lens[l]);
            else hipLaunchKernelGGL((haar_dim_bf16_kernel<false>), dim3(g2), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, rows, dim, lens[l]);
            SPV_LAUNCH_CHECK("spv_haar_dwt(dim, bf16)");
            src = dst;
            continue;
        }
        hipLaunchKernelGGL(haar_level_kernel, dim3(grid), dim3(256), 0, st, src, dst, total, tokens, dim, axis == 1, lens[l], inverse,
                           dtype == SPV_BF16, zero_mode);
        SPV_LAUNCH_CHECK("spv_haar_dwt");
        src = dst;
    }
    return 0;
}


// ---- mixer + LayerNorm-1 + residual as one kernel each way (bf16, dim 512, tokens <= 65: the shapes of fnet_mfma_kernel)
extern "C" int spv_fnet_ln_supported(int tokens, int dim, int dtype) {
    static const bool off = SPV_LAB_SET("SPV_FNET_NO_V2") || SPV_LAB_SET("SPV_FNET_NO_FUSE");
    return (!off && dtype == SPV_BF16 && dim == V2D && tokens >= 2 && tokens <= 65) ? 1 : 0;
}

static int fnet_ln_stagger(int batch) {
    static const int env = SPV_LAB_INT("SPV_FNET_LN_STAGGER", 0);  // x 8128 cycles (+ 256: by threadgroup slot instead of grid half); tuning aid
    return batch >= 512 ? env : 0;
}
static size_t fnet_v2_lds(int tokens) { return (size_t)std::max(2 * ((tokens + 1) / 2), 2 * (tokens / 2 + 1)) * V2RS; }

extern "C" int spv_fnet_ln_fwd(const void* x, void* prenorm, void* out, const float* gamma, const float* beta, float* mean, float* rstd,
                               const float* twiddle, int batch, int tokens, int dim, int dtype, void* stream) {
    SPV_CHECK(batch > 0 && spv_fnet_ln_supported(tokens, dim, dtype), "spv_fnet_ln_fwd: unsupported shape %d x %d x %d / dtype %d", batch,
              tokens, dim, dtype);
    SPV_CHECK(twiddle && gamma && beta && mean && rstd && prenorm, "spv_fnet_ln_fwd: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    FnetLn ln{gamma, beta, mean, rstd, static_cast<const bf16_t*>(x), static_cast<bf16_t*>(prenorm), nullptr, nullptr};
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fnet_mfma_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    SPV_COUNT_PATH(SPV_PATH_FNET_MFMA);
    hipLaunchKernelGGL(fnet_mfma_kernel<1>, dim3(batch), dim3(512), fnet_v2_lds(tokens), st, static_cast<const bf16_t*>(x),
                       static_cast<bf16_t*>(out), reinterpret_cast<const uint4*>(twiddle + v2_frag_off(tokens)), reinterpret_cast<const uint4*>(twiddle + v2_extra_off(tokens)),
                       twiddle + v2_tw_off(tokens), tokens, fnet_ln_stagger(batch), static_cast<const bf16_t*>(nullptr), ln);
    SPV_LAUNCH_CHECK("spv_fnet_ln_fwd");
    return 0;
}

extern "C" int spv_fnet_ln_bwd(const void* dout, const void* prenorm, const float* mean, const float* rstd, const float* gamma, void* dx,
                               float* dgamma, float* dbeta, float* partials, const float* twiddle, int batch, int tokens, int dim,
                               int dtype, void* stream) {
    SPV_CHECK(batch > 0 && spv_fnet_ln_supported(tokens, dim, dtype), "spv_fnet_ln_bwd: unsupported shape %d x %d x %d / dtype %d", batch,
              tokens, dim, dtype);
    SPV_CHECK(twiddle && gamma && mean && rstd && prenorm && partials && ((dgamma != nullptr) == (dbeta != nullptr)), "spv_fnet_ln_bwd: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    FnetLn ln{gamma, nullptr, const_cast<float*>(mean), const_cast<float*>(rstd), nullptr, nullptr, static_cast<const bf16_t*>(prenorm), partials};
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fnet_mfma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // input = dout (LayerNorm backward applied while it is staged); + dout again at the end: the residual path
    SPV_COUNT_PATH(SPV_PATH_FNET_MFMA);
    hipLaunchKernelGGL(fnet_mfma_kernel<2>, dim3(batch), dim3(512), fnet_v2_lds(tokens) + 2 * V2D * sizeof(float), st,
                       static_cast<const bf16_t*>(dout), static_cast<bf16_t*>(dx), reinterpret_cast<const uint4*>(twiddle + v2_frag_off(tokens)),
                       reinterpret_cast<const uint4*>(twiddle + v2_extra_off(tokens)), twiddle + v2_tw_off(tokens), tokens, fnet_ln_stagger(batch), static_cast<const bf16_t*>(dout), ln);
    SPV_LAUNCH_CHECK("spv_fnet_ln_bwd");
    // dgamma == NULL: the caller folds the partials itself (deferred: spv_reduce_multi kind 1 with parts = batch, nsum = 2, n = dim)
    if (dgamma != nullptr) hipLaunchKernelGGL(fnet_ln_fold_kernel, dim3(2 * V2D / FOLD_COLS), dim3(FOLD_COLS * FOLD_ROWS), 0, st, partials, dgamma, dbeta, batch);
    SPV_LAUNCH_CHECK("spv_fnet_ln_bwd(fold)");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Row 0 only of  x1 = LayerNorm1(Re(F_N x F_D)) + x  (the LAST layer of a stack whose consumer reads the CLS row: hip_ops.
// LAST_LAYER_CLS_ONLY).  Token frequency 0 is the plain sum over tokens, so
//   m0[j] = Re(F_N x F_D)[0, j] = sum_d s[d] cos(2 pi j d / D),  s[d] = sum_n x[n, d]  =  Re(FFT_D(s))[j]:
// one pass over the sample (66 KB) and ONE D-point FFT instead of the 2-D transform; the backward is dx[n, :] = Re(FFT_D(dm0)) for
// every n (+ the residual's gradient in row 0): one 66 KB write.  One workgroup of D / 2 threads per sample, thread t owns columns
// 2t and 2t + 1; the FFT is a radix-2 decimation-in-time pass over a bit-reversed LDS image, fp32 throughout.
namespace {
template <int D> __device__ __forceinline__ int bitrev(int v) {
    constexpr int LG = D == 256 ? 8 : D == 512 ? 9 : 10;
    return (int)(__builtin_bitreverse32((unsigned)v) >> (32 - LG));
}
// in: re[] holds the real input in bit-reversed order, im[] zero; out: natural order.  tw: cos / sin of 2 pi k / D, k < D / 2
template <int D> __device__ __forceinline__ void fft_real_lds(float* re, float* im, const float* twc, const float* tws, int t) {
#pragma unroll 1
    for (int len = 2; len <= D; len <<= 1) {
        const int half = len >> 1, k = t & (half - 1), i0 = ((t / half) * len) + k, i1 = i0 + half;
        const int ti = k * (D / len);
        const float c = twc[ti], sn = -tws[ti];   // exp(-2 pi i k / len)
        __syncthreads();
        const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
        const float tr = br * c - bi * sn, tim = br * sn + bi * c;
        re[i0] = ar + tr; im[i0] = ai + tim;
        re[i1] = ar - tr; im[i1] = ai - tim;
    }
    __syncthreads();
}
template <int D> __device__ __forceinline__ float block_sum(float v, float* red, int t) {
    v = wave_sum(v);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    float a = 0.0f;
#pragma unroll
    for (int w = 0; w < D / 128; ++w) a += red[w];
    return a;
}

template <typename T, int D>
__global__ __launch_bounds__(D / 2) void fnet_cls_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             T* __restrict__ out, float* __restrict__ m0, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int N) {
    __shared__ float re[D], im[D], twc[D / 2], tws[D / 2], red[8];
    const int b = blockIdx.x, t = threadIdx.x;
    const T* xs = x + (size_t)b * N * D + 2 * t;
    const float x00 = io<T>::ld(xs), x01 = io<T>::ld(xs + 1);
    // token sum with 16-byte loads: EPV columns per lane, RG row groups of D / EPV lanes; partial sums joined through LDS (im[] and
    // re[] are free until the FFT image is written)
    constexpr int EPV = 16 / (int)sizeof(T), LPR = D / EPV, RG = (D / 2) / LPR;
    {
        const int rg = t / LPR, c0 = (t % LPR) * EPV;
        float acc[EPV];
#pragma unroll
        for (int u = 0; u < EPV; ++u) acc[u] = 0.0f;
        const T* xr = x + (size_t)b * N * D + c0;
#pragma unroll 4
        for (int n = rg; n < N; n += RG) {
            if constexpr (sizeof(T) == 2) {
                const uint4 q = *reinterpret_cast<const uint4*>(xr + (size_t)n * D);
                acc[0] += __uint_as_float(q.x << 16); acc[1] += __uint_as_float(q.x & 0xffff0000u);
                acc[2] += __uint_as_float(q.y << 16); acc[3] += __uint_as_float(q.y & 0xffff0000u);
                acc[4] += __uint_as_float(q.z << 16); acc[5] += __uint_as_float(q.z & 0xffff0000u);
                acc[6] += __uint_as_float(q.w << 16); acc[7] += __uint_as_float(q.w & 0xffff0000u);
            } else {
                const float4 q = *reinterpret_cast<const float4*>(xr + (size_t)n * D);
                acc[0] += q.x; acc[1] += q.y; acc[2] += q.z; acc[3] += q.w;
            }
        }
        float* part = rg < 1 ? re : (rg < 2 ? im : nullptr);
        static_assert(RG <= 4, "row groups");
        // RG = 2 (fp32): groups in re, im.  RG = 4 (bf16): groups 0, 1 in re / im, groups 2, 3 added in a second round
        if (RG == 2 || rg < 2) {
#pragma unroll
            for (int u = 0; u < EPV; ++u) part[c0 + u] = acc[u];
        }
        __syncthreads();
        if (RG == 4 && rg >= 2) {
            float* p2 = rg == 2 ? re : im;
#pragma unroll
            for (int u = 0; u < EPV; ++u) p2[c0 + u] += acc[u];
        }
        __syncthreads();
    }
    const float s0 = re[2 * t] + im[2 * t], s1 = re[2 * t + 1] + im[2 * t + 1];
    __syncthreads();
    {
        float sn, c;
        sincospif(2.0f * (float)t / (float)D, &sn, &c);
        twc[t] = c; tws[t] = sn;
    }
    re[bitrev<D>(2 * t)] = s0; re[bitrev<D>(2 * t + 1)] = s1;
    im[2 * t] = 0.0f; im[2 * t + 1] = 0.0f;
    __syncthreads();   // (hipcc hoists the loop's twiddle reads above the loop's own first barrier: the table must be complete here)
    fft_real_lds<D>(re, im, twc, tws, t);
    const float a0 = re[2 * t], a1 = re[2 * t + 1];
    const float mu = block_sum<D>(a0 + a1, red, t) * (1.0f / D);
    const float d0 = a0 - mu, d1 = a1 - mu;
    const float var = block_sum<D>(d0 * d0 + d1 * d1, red, t) * (1.0f / D);
    const float rs = rsqrtf(var + 1e-5f);
    m0[(size_t)b * D + 2 * t] = a0; m0[(size_t)b * D + 2 * t + 1] = a1;
    if (t == 0) { mean[b] = mu; rstd[b] = rs; }
    io<T>::st(out + (size_t)b * D + 2 * t, d0 * rs * gamma[2 * t] + beta[2 * t] + x00);
    io<T>::st(out + (size_t)b * D + 2 * t + 1, d1 * rs * gamma[2 * t + 1] + beta[2 * t + 1] + x01);
}

template <typename T, int D>
__global__ __launch_bounds__(D / 2) void fnet_cls_bwd_kernel(const T* __restrict__ g1, const float* __restrict__ m0, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma, T* __restrict__ dx,
                                                             float* __restrict__ partials, int N) {
    __shared__ float re[D], im[D], twc[D / 2], tws[D / 2], red[8];
    const int b = blockIdx.x, t = threadIdx.x;
    const float mu = mean[b], rs = rstd[b];
    const float g0 = io<T>::ld(g1 + (size_t)b * D + 2 * t), gq = io<T>::ld(g1 + (size_t)b * D + 2 * t + 1);
    const float h0 = (m0[(size_t)b * D + 2 * t] - mu) * rs, h1 = (m0[(size_t)b * D + 2 * t + 1] - mu) * rs;
    float* pp = partials + (size_t)b * 2 * D;   // [2][D]: dgamma, dbeta contributions of this sample (folded over the batch by a fold job)
    pp[2 * t] = g0 * h0; pp[2 * t + 1] = gq * h1;
    pp[D + 2 * t] = g0; pp[D + 2 * t + 1] = gq;
    const float q0 = g0 * gamma[2 * t], q1 = gq * gamma[2 * t + 1];
    const float sa = block_sum<D>(q0 + q1, red, t) * (1.0f / D);
    const float sb = block_sum<D>(q0 * h0 + q1 * h1, red, t) * (1.0f / D);
    const float e0 = rs * (q0 - sa - h0 * sb), e1 = rs * (q1 - sa - h1 * sb);
    {
        float sn, c;
        sincospif(2.0f * (float)t / (float)D, &sn, &c);
        twc[t] = c; tws[t] = sn;
    }
    re[bitrev<D>(2 * t)] = e0; re[bitrev<D>(2 * t + 1)] = e1;
    im[2 * t] = 0.0f; im[2 * t + 1] = 0.0f;
    __syncthreads();   // (hipcc hoists the loop's twiddle reads above the loop's own first barrier: the table must be complete here)
    fft_real_lds<D>(re, im, twc, tws, t);
    // every token row gets the same spectrum (row 0 + the residual's gradient): 16-byte stores, EPV columns per lane, RG row groups
    constexpr int EPV = 16 / (int)sizeof(T), LPR = D / EPV, RG = (D / 2) / LPR;
    const int rg = t / LPR, c0 = (t % LPR) * EPV;
    float v[EPV];
#pragma unroll
    for (int u = 0; u < EPV; ++u) v[u] = re[c0 + u];
    T* drow = dx + (size_t)b * N * D + c0;
    if (rg == 0) {
        float r0[EPV];
#pragma unroll
        for (int u = 0; u < EPV; ++u) r0[u] = v[u] + io<T>::ld(g1 + (size_t)b * D + c0 + u);
        if constexpr (sizeof(T) == 2) {
            uint4 q;
            q.x = pack_bf16x2(r0[0], r0[1]); q.y = pack_bf16x2(r0[2], r0[3]); q.z = pack_bf16x2(r0[4], r0[5]); q.w = pack_bf16x2(r0[6], r0[7]);
            *reinterpret_cast<uint4*>(drow) = q;
        } else {
            *reinterpret_cast<float4*>(drow) = make_float4(r0[0], r0[1], r0[2], r0[3]);
        }
    }
    if constexpr (sizeof(T) == 2) {
        uint4 q;
        q.x = pack_bf16x2(v[0], v[1]); q.y = pack_bf16x2(v[2], v[3]); q.z = pack_bf16x2(v[4], v[5]); q.w = pack_bf16x2(v[6], v[7]);
#pragma unroll 4
        for (int n = rg == 0 ? RG : rg; n < N; n += RG) *reinterpret_cast<uint4*>(drow + (size_t)n * D) = q;
    } else {
        const float4 q = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll 4
        for (int n = rg == 0 ? RG : rg; n < N; n += RG) *reinterpret_cast<float4*>(drow + (size_t)n * D) = q;
    }
}
}  // namespace

extern "C" int spv_fnet_cls_supported(int tokens, int dim, int dtype) {
    static const bool off = SPV_LAB_SET("SPV_FNET_NO_CLS");
    return (!off && (dtype == SPV_BF16 || dtype == SPV_F32) && (dim == 256 || dim == 512 || dim == 1024) && tokens >= 1) ? 1 : 0;
}

extern "C" int spv_fnet_cls_fwd(const void* x, const float* gamma, const float* beta, void* out, float* m0, float* mean, float* rstd, int batch,
                                int tokens, int dim, int dtype, void* stream) {
    SPV_CHECK(spv_fnet_cls_supported(tokens, dim, dtype), "spv_fnet_cls_fwd: unsupported tokens=%d dim=%d dtype=%d", tokens, dim, dtype);
    SPV_CHECK(batch > 0, "spv_fnet_cls_fwd: empty batch");
    hipStream_t st = static_cast<hipStream_t>(stream);
#define SPV_CLS_F(TT, DD) hipLaunchKernelGGL((fnet_cls_fwd_kernel<TT, DD>), dim3(batch), dim3(DD / 2), 0, st, (const TT*)x, gamma, beta, (TT*)out, m0, mean, rstd, tokens)
    if (dtype == SPV_BF16) { if (dim == 256) SPV_CLS_F(bf16_t, 256); else if (dim == 512) SPV_CLS_F(bf16_t, 512); else SPV_CLS_F(bf16_t, 1024); }
    else { if (dim == 256) SPV_CLS_F(float, 256); else if (dim == 512) SPV_CLS_F(float, 512); else SPV_CLS_F(float, 1024); }
#undef SPV_CLS_F
    SPV_LAUNCH_CHECK("spv_fnet_cls_fwd");
    return 0;
}

extern "C" int spv_fnet_cls_bwd(const void* g1, const float* m0, const float* mean, const float* rstd, const float* gamma, void* dx,
                                float* partials, int batch, int tokens, int dim, int dtype, void* stream) {
    SPV_CHECK(spv_fnet_cls_supported(tokens, dim, dtype), "spv_fnet_cls_bwd: unsupported tokens=%d dim=%d dtype=%d", tokens, dim, dtype);
    SPV_CHECK(batch > 0, "spv_fnet_cls_bwd: empty batch");
    hipStream_t st = static_cast<hipStream_t>(stream);
#define SPV_CLS_B(TT, DD) hipLaunchKernelGGL((fnet_cls_bwd_kernel<TT, DD>), dim3(batch), dim3(DD / 2), 0, st, (const TT*)g1, m0, mean, rstd, gamma, (TT*)dx, partials, tokens)
    if (dtype == SPV_BF16) { if (dim == 256) SPV_CLS_B(bf16_t, 256); else if (dim == 512) SPV_CLS_B(bf16_t, 512); else SPV_CLS_B(bf16_t, 1024); }
    else { if (dim == 256) SPV_CLS_B(float, 256); else if (dim == 512) SPV_CLS_B(float, 512); else SPV_CLS_B(float, 1024); }
#undef SPV_CLS_B
    SPV_LAUNCH_CHECK("spv_fnet_cls_bwd");
    return 0;
}
