// spv_gemm.hip -- C[M,N] = A[M,K] . B[N,K]^T (+bias)(+C), MFMA on gfx950.
//
// Both operands are K-contiguous ("NT"), which is the only layout the module mirror ever issues:
//   forward   h  = x . W^T          A = x [rows,K],   B = W   [N,K]
//   data grad dx = dh . W           A = dh [rows,N],  B = W^T [K,N]   (W^T kept beside the bf16 shadow of W)
//   weight grad dW = dh^T . x       A = dh^T [N,rows], B = x^T [K,rows], split-K over rows
//
// Tile: 128 x 128 x (128 bytes of K) per 256-thread workgroup; 4 waves as 2 x 2, each wave owns a
// 64 x 64 block = 2 x 2 MFMA 32x32 accumulators (64 fp32 registers).  LDS rows are 128 B of data + 16 B
// pad: 16 consecutive rows then cover all 64 banks exactly once for ds_read_b128 (conflict free).
// Global -> registers -> LDS staging with the next tile's loads issued before the MFMAs of the current
// one (one barrier pair per K step).  bf16: v_mfma_f32_32x32x16_bf16; fp32: v_mfma_f32_32x32x2_f32
// (exact fp32 fma chain -- the parity path).
#define SPV_USES_SEED
#include "spv_common.h"

#include <stdlib.h>

#include <algorithm>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128 + 16;  // LDS bytes per tile row
constexpr int KBYTES = 128;     // bytes of K per tile row

template <typename T> struct KT { static constexpr int BK = KBYTES / sizeof(T); static constexpr int CH = 16 / sizeof(T); };

template <typename TO> __device__ __forceinline__ void store_out(TO* p, float v);
template <> __device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_out<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <typename TO> __device__ __forceinline__ float load_out(const TO* p);
template <> __device__ __forceinline__ float load_out<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_out<bf16_t>(const bf16_t* p) { return bf2f(*p); }

// XCD-aware, bijective remap of the linear workgroup id (guide T1): workgroups that share an XCD
// (id % 8) get a contiguous chunk of tiles, so neighbouring tiles re-use operand panels in that XCD's L2.
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = id & 7;
    int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (id >> 3);
}

// tile id -> (tm, tn).  Few N tiles (the layer GEMMs, N <= 1024): row-major, the N tiles of one M panel are neighbours and
// the A panel is fetched once per XCD.  Many N tiles (the MHPermutMix data gradient, N = 8192: the 8 MiB weight does not
// fit an XCD's 4 MiB L2 and was re-fetched for every one of the 260 M panels -- 2.2 GB of L2 misses per launch,
// FETCH_SIZE): groups of 8 M panels are walked column by column, so ~100 co-running tiles touch 8 A panels and a dozen B
// panels (2.5 MiB) at a time: 0.33 GB of L2 misses instead of 2.2 GB.  (The launch time did not move -- the misses were
// Infinity Cache hits -- but the fabric carries 7x less beside the RCCL traffic.)
__device__ __forceinline__ void tile_coords(int tile, int tiles_n, int tiles_mn, int& tm, int& tn) {
    constexpr int GM = 8;
    if (tiles_n <= 8) { tm = tile / tiles_n; tn = tile % tiles_n; return; }
    const int tiles_m = tiles_mn / tiles_n;
    const int per_group = GM * tiles_n;
    const int grp = tile / per_group, in = tile % per_group;
    const int m_first = grp * GM;
    const int gm = min(GM, tiles_m - m_first);  // the last group may be short
    tm = m_first + in % gm;
    tn = in / gm;
}

// Epilogue shared by the NT and TN kernels: the accumulators go through a wave-private fp32 LDS stage
// ([32 rows][64 + 4 pad]) so that every lane leaves with 8 CONSECUTIVE columns of one row: 16-byte (bf16) /
// 2 x 16-byte (fp32) global stores, 8 rows x 128 B per wave instruction, instead of 64 two-byte stores per lane
// (store-issue bound).  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
template <typename TO>
__device__ __forceinline__ void store_acc_tile(f32x16 (*acc)[2], unsigned char* smem, const float* __restrict__ bias,
                                               TO* __restrict__ C, float* __restrict__ ws, int M, int N, int ldc, int accumulate,
                                               int m0, int n0, int split, int rg, int gs, int roff,
                                               const float* __restrict__ bias2d, const void* __restrict__ bc = nullptr,
                                               int bc_pw = 0, int bc_bf = 0, int wrow = -1, int wcol = -1, float drop_p = 0.0f,
                                               uint64_t drop_seed = 0) {
    // acc: two rows of two 32x32 accumulators = this wave's 64 x 64 block at tile rows wrow.. (default (wave >> 1) * 64)
    // and tile columns wcol.. (default (wave & 1) * 64)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fh = lane >> 5;
    if (wrow < 0) wrow = (wave >> 1) * 64;
    if (wcol < 0) wcol = (wave & 1) * 64;
    float* stage = reinterpret_cast<float*>(smem + wave * 9216);  // 9216 B per wave >= 32 * 68 * 4
    constexpr int SLD = 68;
    const bool vec_c = ((size_t)ldc * sizeof(TO)) % 16 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    const bool vec_ws = (N & 3) == 0;
    // the replay-time seed word is ONE global load per tile, issued here (inside the chunk loop it was a dependent L2 round trip in
    // front of every chunk's hash: 7.5 of the embedding GEMM's 25 us)
    const uint64_t seed_live = drop_p > 0.0f ? live_seed(drop_seed) : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stage[((r & 3) + 8 * (r >> 2) + 4 * fh) * SLD + j * 32 + (lane & 31)] = acc[i][j][r];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int qd = lane + 64 * t;
            const int lr = qd >> 3, c8 = qd & 7;
            const int row = m0 + wrow + i * 32 + lr;
            const int col0 = n0 + wcol + c8 * 8;
            if (row >= M || col0 >= N) continue;
            float v[8];
            {
                const float4 lo = *reinterpret_cast<const float4*>(stage + lr * SLD + c8 * 8);
                const float4 hi = *reinterpret_cast<const float4*>(stage + lr * SLD + c8 * 8 + 4);
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            }
            const int nvalid = min(8, N - col0);
            if (ws != nullptr) {
                float* wp = ws + ((size_t)split * M + row) * N + col0;
                if (nvalid == 8 && vec_ws) {
                    *reinterpret_cast<float4*>(wp) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(wp + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                    for (int u = 0; u < nvalid; ++u) wp[u] = v[u];
                }
                continue;
            }
            int orow = row;
            const float* b2 = nullptr;
            if (rg > 0) {  // grouped output rows (patch rows -> token rows, skipping each image's CLS row)
                orow = (row / rg) * gs + roff + (row % rg);
                if (bias2d) b2 = bias2d + (size_t)(row % rg) * N + col0;
            }
            TO* cp = C + (size_t)orow * ldc + col0;
            if (bc != nullptr) {
                // + P[row][col / pw] / pw: the transposed exact-window pooling of the SpectreLinear skip (backward of
                // layers.py:93,101) broadcast straight into the data gradient instead of a rows x K buffer written by the
                // tail kernel and re-read here
                const float sc = 1.0f / (float)bc_pw;
                const int ldp = N / bc_pw;
                if ((bc_pw & 7) == 0) {  // the lane's 8 columns lie in one window: one load, one divide
                    const size_t o = (size_t)row * ldp + col0 / bc_pw;
                    const float t = (bc_bf ? bf2f(static_cast<const bf16_t*>(bc)[o]) : static_cast<const float*>(bc)[o]) * sc;
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] += t;
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (u < nvalid) {
                            const size_t o = (size_t)row * ldp + (col0 + u) / bc_pw;
                            v[u] += (bc_bf ? bf2f(static_cast<const bf16_t*>(bc)[o]) : static_cast<const float*>(bc)[o]) * sc;
                        }
                    }
                }
            }
            if (nvalid == 8 && vec_c) {
                if (bias != nullptr) {
                    const float4 b0 = *reinterpret_cast<const float4*>(bias + col0), b1 = *reinterpret_cast<const float4*>(bias + col0 + 4);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                }
                if (b2 != nullptr) {
                    if (vec_ws && (reinterpret_cast<uintptr_t>(bias2d) & 15) == 0) {   // two 16-byte loads (eight 4-byte ones: 34 -> us of the embedding GEMM)
                        const float4 p0 = *reinterpret_cast<const float4*>(b2), p1 = *reinterpret_cast<const float4*>(b2 + 4);
                        v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w; v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
                    } else {
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] += b2[u];
                    }
                }
                if (drop_p > 0.0f) {
                    // the nn.Dropout that follows the projection (spectre.py:156), on the flat index of the contiguous output: the
                    // mask spv_dropout / spv_embed_bwd derive from the same seed
                    const size_t fi = (size_t)orow * ldc + col0;
                    const unsigned key = dropout_row_key(seed_live, (uint64_t)fi >> 12);
                    const unsigned c0 = (unsigned)(fi & 4095);
                    const float inv_keep = 1.0f / (1.0f - drop_p);
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] *= dropout_scale(key, c0 + u, drop_p, inv_keep);
                }
                if constexpr (sizeof(TO) == 2) {
                    if (accumulate) {
                        const uint4 old = *reinterpret_cast<const uint4*>(cp);
                        v[0] += __uint_as_float(old.x << 16); v[1] += __uint_as_float(old.x & 0xffff0000u);
                        v[2] += __uint_as_float(old.y << 16); v[3] += __uint_as_float(old.y & 0xffff0000u);
                        v[4] += __uint_as_float(old.z << 16); v[5] += __uint_as_float(old.z & 0xffff0000u);
                        v[6] += __uint_as_float(old.w << 16); v[7] += __uint_as_float(old.w & 0xffff0000u);
                    }
                    uint4 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    o.z = pack_bf16x2(v[4], v[5]);
                    o.w = pack_bf16x2(v[6], v[7]);
                    *reinterpret_cast<uint4*>(cp) = o;
                } else {
                    float4* c4 = reinterpret_cast<float4*>(cp);
                    if (accumulate) {
                        const float4 o0 = c4[0], o1 = c4[1];
                        v[0] += o0.x; v[1] += o0.y; v[2] += o0.z; v[3] += o0.w; v[4] += o1.x; v[5] += o1.y; v[6] += o1.z; v[7] += o1.w;
                    }
                    c4[0] = make_float4(v[0], v[1], v[2], v[3]);
                    c4[1] = make_float4(v[4], v[5], v[6], v[7]);
                }
            } else {
                for (int u = 0; u < nvalid; ++u) {
                    float x = v[u];
                    if (bias != nullptr) x += bias[col0 + u];
                    if (b2 != nullptr) x += b2[u];
                    if (drop_p > 0.0f) {
                        const size_t fi = (size_t)orow * ldc + col0 + u;
                        x *= dropout_scale(dropout_row_key(seed_live, (uint64_t)fi >> 12), (unsigned)(fi & 4095), drop_p, 1.0f / (1.0f - drop_p));
                    }
                    if (accumulate) x += load_out<TO>(cp + u);
                    store_out<TO>(cp + u, x);
                }
            }
        }
    }
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const T* __restrict__ A, const T* __restrict__ B,
                                                      const float* __restrict__ bias, TO* __restrict__ C,
                                                      float* __restrict__ ws, int M, int N, int K, int lda,
                                                      int ldb, int ldc, int k_per_split, int accumulate,
                                                      int tiles_n, int tiles_mn, int nsplit, int rg, int gs, int roff,
                                                      const float* __restrict__ bias2d, const void* __restrict__ bc, int bc_pw,
                                                      int bc_bf, float drop_p, uint64_t drop_seed) {
    constexpr int BK = KT<T>::BK;
    constexpr int CH = KT<T>::CH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BM * ROWB];
    unsigned char* sA = smem;
    unsigned char* sB = smem + BM * ROWB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // one linear id over (split, tile), XCD-remapped as a whole: each XCD gets whole K-slices, so the tiles that
    // share a slice's operand rows hit in that XCD's L2 (per-slice round-robin made every XCD fetch every slice:
    // 3.4x the algorithmic HBM reads on the weight-gradient GEMM, FETCH_SIZE)
    const int lin = xcd_remap(blockIdx.x, tiles_mn * nsplit);
    const int split = lin / tiles_mn;
    const int tile = lin % tiles_mn;
    // consecutive tile ids sweep the N tiles of one M panel -> the A panel is fetched once per XCD
    int tm, tn;
    tile_coords(tile, tiles_n, tiles_mn, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);

    // staging map: 128 rows x 8 chunks of 16 B per operand; thread -> (row = tid/8 + 32 i, chunk = tid%8)
    const int srow = tid >> 3, sch = tid & 7;
    uint4 ra[4], rb[4];
    const T* aptr[4];
    const T* bptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int gm = min(m0 + srow + 32 * i, M - 1);
        int gn = min(n0 + srow + 32 * i, N - 1);
        aptr[i] = A + (size_t)gm * lda + sch * CH;
        bptr[i] = B + (size_t)gn * ldb + sch * CH;
    }
    auto load_tile = [&](int k0) {
        const bool ok = (k0 + sch * CH) < kend;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = ok ? *reinterpret_cast<const uint4*>(aptr[i] + k0) : make_uint4(0, 0, 0, 0);
            rb[i] = ok ? *reinterpret_cast<const uint4*>(bptr[i] + k0) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4*>(sA + (srow + 32 * i) * ROWB + sch * 16) = ra[i];
            *reinterpret_cast<uint4*>(sB + (srow + 32 * i) * ROWB + sch * 16) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int frow = lane & 31, fh = lane >> 5;
    const unsigned char* fa0 = sA + (wm * 64 + frow) * ROWB;
    const unsigned char* fb0 = sB + (wn * 64 + frow) * ROWB;

    load_tile(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        store_tile();
        __syncthreads();
        if (k0 + BK < kend) load_tile(k0 + BK);
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 a[2], b[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    a[f] = *reinterpret_cast<const bf16x8*>(fa0 + f * 32 * ROWB + ks * 32 + fh * 16);
                    b[f] = *reinterpret_cast<const bf16x8*>(fb0 + f * 32 * ROWB + ks * 32 + fh * 16);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // fp32: lane half fh owns k = 16 fh + s (any bijection works as long as A and B agree)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 a[2], b[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    a[f] = *reinterpret_cast<const float4*>(fa0 + f * 32 * ROWB + fh * 64 + q * 16);
                    b[f] = *reinterpret_cast<const float4*>(fb0 + f * 32 * ROWB + fh * 64 + q * 16);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }

    store_acc_tile<TO>(acc, smem, bias, C, ws, M, N, ldc, accumulate, m0, n0, split, rg, gs, roff, bias2d, bc, bc_pw, bc_bf, -1, -1, drop_p, drop_seed);
}

// ---------------------------------------------------------------------------------------------------------
// NT contraction, direct-to-LDS variant (bf16, K and the split-K slices multiples of 64): the same 128 x 128 x 64
// tile, but the operands are staged with global_load_lds_dwordx4 (no staging VGPRs, one 1-KiB LDS burst per wave
// instruction = 8 tile rows) into TWO LDS buffers, so the loads of K-step t+1 fly while the MFMAs of step t run
// and one barrier per step is enough.  LDS-DMA writes are lane-linear, so the tile rows cannot be padded; bank
// conflicts are removed by an XOR swizzle of the 16-byte chunk index with (row >> 1) & 7, applied on the per-lane
// SOURCE address and again on the fragment read (guide rule 21): the 16 rows a ds_read_b128 lane group touches then
// fall on 8 chunk positions x 2 row parities = all 64 banks.
constexpr int GBK = 64;  // split-K slices of the direct-to-LDS kernel are multiples of this many bf16 elements

// KB = bytes of K per tile row per stage: 128 (64 bf16, 2 x 32 KiB stages, 2 workgroups per CU: long-K problems) or
// 64 (32 bf16, 2 x 16 KiB stages, 4 workgroups per CU: the skinny K = 512..768 layer GEMMs, which are bound by load
// latency per K-step rather than by MFMA issue and want more independent tiles in flight per CU).
template <typename TO, int KB, int NST>
__global__ __launch_bounds__(256) void gemm_nt_glds_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                           const float* __restrict__ bias, TO* __restrict__ C,
                                                           float* __restrict__ ws, int M, int N, int K, int lda, int ldb,
                                                           int ldc, int k_per_split, int accumulate, int tiles_n,
                                                           int tiles_mn, int nsplit, int rg, int gs, int roff,
                                                           const float* __restrict__ bias2d, const void* __restrict__ bc,
                                                           int bc_pw, int bc_bf, float drop_p, uint64_t drop_seed) {
    constexpr int CPR = KB / 16;              // 16-byte chunks per tile row
    constexpr int RPI = 64 / CPR;             // tile rows written by one wave instruction (1 KiB)
    constexpr int IPW = (BM / RPI) / 4;       // DMA instructions per wave per operand per stage
    constexpr int STAGE = (BM + BN) * KB;     // bytes per stage
    constexpr int KE = KB / 2;                // bf16 elements of K per stage
    constexpr int SM = NST * STAGE > 36864 ? NST * STAGE : 36864;  // the epilogue stage needs 36 KiB
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // one linear id over (split, tile), XCD-remapped as a whole: each XCD gets whole K-slices, so the tiles that
    // share a slice's operand rows hit in that XCD's L2 (per-slice round-robin made every XCD fetch every slice:
    // 3.4x the algorithmic HBM reads on the weight-gradient GEMM, FETCH_SIZE)
    const int lin = xcd_remap(blockIdx.x, tiles_mn * nsplit);
    const int split = lin / tiles_mn;
    const int tile = lin % tiles_mn;
    int tm, tn;
    tile_coords(tile, tiles_n, tiles_mn, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    auto swizzle = [](int R) { return CPR == 8 ? ((R >> 1) & 7) : ((R >> 2) & 3); };

    // per-lane source pointers of this wave's IPW + IPW DMA instructions per stage
    const bf16_t* asrc[IPW];
    const bf16_t* bsrc[IPW];
#pragma unroll
    for (int t = 0; t < IPW; ++t) {
        const int R = RPI * (wave * IPW + t) + lane / CPR;     // tile row written by this lane
        const int c = (lane % CPR) ^ swizzle(R);                // logical 16-byte chunk that belongs at this LDS position
        asrc[t] = A + (size_t)min(m0 + R, M - 1) * lda + c * 8;
        bsrc[t] = B + (size_t)min(n0 + R, N - 1) * ldb + c * 8;
    }
    auto stage = [&](int buf, int k0) {
        unsigned char* sa = smem + buf * STAGE + wave * IPW * 1024;  // this wave's rows of the A tile
        unsigned char* sb = sa + BM * KB;
#pragma unroll
        for (int t = 0; t < IPW; ++t) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[t] + k0),
                                             (__attribute__((address_space(3))) void*)(sa + t * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[t] + k0),
                                             (__attribute__((address_space(3))) void*)(sb + t * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int frow = lane & 31, fh = lane >> 5, swz = swizzle(frow);  // 64-row / 32-row offsets do not change the swizzle
    const int fa_off = (wm * 64 + frow) * KB, fb_off = BM * KB + (wn * 64 + frow) * KB;

    // NST stages in an LDS ring, prefetch distance NST - 1.  With three stages the wait is counted: loads, LDS-DMA and
    // stores retire in issue order, so leaving the youngest stage's 2 * IPW pieces outstanding means the stage about to
    // be read has landed (in-kernel stamps on the 33280 x 768 x 512 layer GEMM: 250-280 cycles at this wait per K step
    // instead of 760-850 with two stages and vmcnt(0)).
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (kbeg + s * KE < kend) stage(s, kbeg + s * KE);
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += KE) {
        if (NST >= 3 && k0 + KE < kend) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of stage `buf` has landed
        __builtin_amdgcn_s_barrier();                          // everyone's has; everyone is done reading the stage before
        if (k0 + (NST - 1) * KE < kend) stage((buf + NST - 1) % NST, k0 + (NST - 1) * KE);
        const unsigned char* sa = smem + buf * STAGE + fa_off;
        const unsigned char* sb = smem + buf * STAGE + fb_off;
#pragma unroll
        for (int ks = 0; ks < KB / 32; ++ks) {
            const int ch = ((ks * 2 + fh) ^ swz) * 16;
            bf16x8 a[2], b[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                a[f] = *reinterpret_cast<const bf16x8*>(sa + f * 32 * KB + ch);
                b[f] = *reinterpret_cast<const bf16x8*>(sb + f * 32 * KB + ch);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
    store_acc_tile<TO>(acc, smem, bias, C, ws, M, N, ldc, accumulate, m0, n0, split, rg, gs, roff, bias2d, bc, bc_pw, bc_bf, -1, -1, drop_p, drop_seed);
}

// ---------------------------------------------------------------------------------------------------------
// NT contraction, strip kernel (bf16 in, bf16 out, N a multiple of 256, K of 128): the layer GEMMs of the training step
// (33280 x 768 x 512 and 33280 x 512 x 768).  On the 128 x 128 kernels those are 1560 / 1040 tiles on 768 workgroup
// slots (2.03 / 1.35 dispatch rounds, the last one nearly empty), and a 128 x 128 x 64 K-step moves as many bytes
// L2 -> LDS as the CU can take in the time of its MFMAs (64 B/clk against 4096 flop/clk).  Here ONE 8-wave workgroup
// per CU owns a 256-column strip of C and an uneven run of 32-row blocks (the host hands out ceil/floor shares, so the
// grid is <= 256 workgroups with the same work +- one block):
//   * a sub-tile is 2*MB row blocks x 256 columns (waves as 2 x 4, wave tile 32*MB x 64) plus an optional extra block
//     whose 32 x 256 strip is shared out as one 32 x 32 accumulator per wave -- the +-1 block of the uneven shares
//     costs every wave the same;
//   * operands are staged by LDS-DMA in 64-deep K-tiles, two buffers, one piece of K-tile t+1 issued behind every MFMA
//     pair of K-tile t (bursts stall all 8 waves on the 64 B/clk L1 path);
//   * the pipeline runs on across sub-tiles: the next sub-tile's first two K-tiles are in flight before the epilogue's
//     stores are issued, and the waits behind an epilogue are COUNTED (vmcnt retires loads, LDS-DMA and stores in issue
//     order), so they cover the DMA but not the younger stores;
//   * C leaves through a wave-private fp32 LDS stage as 16-byte write-through (sc1) stores: plain stores leave the
//     lines dirty in L2 and the end-of-kernel write-back is then serial (1-1.5 us per launch).
// Measured (tools/gemm_big_lab.hip, bit-identical output): 31.5 vs 46 us at 33280 x 768 x 512, 29.5 vs 45 us at
// 33280 x 512 x 768.
__device__ __forceinline__ void vmwait_upto(int n) {
    switch (n) {
#define SPV_VMW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        SPV_VMW(1) SPV_VMW(2) SPV_VMW(3) SPV_VMW(4) SPV_VMW(5) SPV_VMW(6) SPV_VMW(7) SPV_VMW(8) SPV_VMW(9) SPV_VMW(10)
        SPV_VMW(11) SPV_VMW(12) SPV_VMW(13) SPV_VMW(14) SPV_VMW(15) SPV_VMW(16) SPV_VMW(17) SPV_VMW(18) SPV_VMW(19) SPV_VMW(20)
        SPV_VMW(21) SPV_VMW(22) SPV_VMW(23) SPV_VMW(24) SPV_VMW(25) SPV_VMW(26) SPV_VMW(27) SPV_VMW(28) SPV_VMW(29) SPV_VMW(30)
#undef SPV_VMW
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// EPI: 0 = store, 1 = accumulate (C += ...), 2 = + bc[row][col / bc_pw] / bc_pw -- the transposed exact-window pooling of the
// SpectreLinear skip broadcast into the data gradient (MHPermutMix linear, N = 8192: 762 us on the 128 x 128 kernel, 14.7 % of peak).
// The old C values (EPI 1) / the pooled value of a lane's 8 columns (EPI 2) are fetched two half-blocks ahead of their store.
template <int MB, int EPIM>
__global__ __launch_bounds__(512) void gemm_nt_strip_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ C, int M, int K,
                                                            int lda, int ldb, int ldc, int nstrips, int base, int rem, int nwg,
                                                            const bf16_t* __restrict__ bc, int bc_pw, int bc_ld) {
    constexpr bool ACC = EPIM == 1;
    constexpr bool BCM = EPIM == 2 || EPIM == 3;
    constexpr bool BC2 = EPIM == 3;   // pooling windows that are not multiples of 8 wide (Base: 12): a lane's 8 columns span two windows
    const float bc_scale = BCM ? 1.0f / (float)bc_pw : 0.0f;
    constexpr int TM = 64 * MB;                       // rows of a sub-tile without the extra block
    constexpr int STAGE = (TM + 32 + 256) * 128;      // one K-tile: A rows, extra A rows, B rows, 128 B (64 bf16) each
    constexpr bool PRE1 = MB <= 3;                    // LDS left for an epilogue stage beside the two K-tile buffers
    constexpr int EPI = 8 * 4352;                     // 8 waves x [16 rows][68] floats
    constexpr int EPI_OFF = PRE1 ? 2 * STAGE : STAGE; // MB = 4: the epilogue stages through K-tile buffer 1
    constexpr int BIAS_OFF = PRE1 ? 2 * STAGE + EPI : 2 * STAGE;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[BIAS_OFF + 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: LDS-DMA bases and wave offsets stay in SGPRs
    const int wm = wave >> 2, wn = wave & 3;
    // the strips of one row group sit on one XCD and run together: its A rows are fetched into that L2 once
    const int id = xcd_remap(blockIdx.x, nwg);
    int g = id / nstrips, strip = id % nstrips;
    // MANY strips (the MHPermutMix data gradient, N = 8192: 32): a whole row group per XCD means 32 different B strips of 256 x K each
    // on its 32 CUs -- 8 MB at K = 512 through a 4 MB L2, re-read for every sub-tile: PMC 1.14 GB fetched per launch against 42 MB of
    // operands.  So an XCD takes 8 strips x (its workgroups / 8) row groups instead: 2 MB of B that stays, A fetched by four XCDs.
    {
        const int per_xcd = nwg >> 3, groups = nwg / nstrips;
        constexpr int S = 8;
        if (nstrips >= 16 && (nwg & 7) == 0 && nstrips % S == 0 && per_xcd % S == 0 && groups % (per_xcd / S) == 0) {
            const int G = per_xcd / S, SS = nstrips / S;
            const int x = id / per_xcd, local = id % per_xcd;
            strip = (x % SS) * S + local % S;
            g = (x / SS) * G + local / S;
        }
    }
    int blk0 = g * base + min(g, rem);
    int cnt = base + (g < rem ? 1 : 0);
    const int n0 = strip * 256;
    const int nkt = K / 64;

    // DMA sources as 32-bit byte offsets from the (uniform) operand bases: the base + K offset stay in SGPRs, one VGPR per
    // piece instead of a 64-bit address pair (the host checks that both operands span < 4 GiB)
    uint32_t bsrc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int R = 8 * (wave * 4 + t) + (lane >> 3);
        const int c = (lane & 7) ^ ((R >> 1) & 7);    // the swizzle of gemm_nt_glds_kernel<., 128, .>
        bsrc[t] = (uint32_t)(((size_t)(n0 + R) * ldb + c * 8) * 2);
    }
    const int frow = lane & 31, fh = lane >> 5, swz = (frow >> 1) & 7;
    const int fa_off = (wm * 32 * MB + frow) * 128;
    const int fax_off = (TM + frow) * 128;
    const int fb_off = (TM + 32 + wn * 64 + frow) * 128;
    const int fbx_off = (TM + 32 + wave * 32 + frow) * 128;
    float* sbias = reinterpret_cast<float*>(smem + BIAS_OFF);  // first read behind the K loop's barriers
    if (tid < 256) sbias[tid] = bias ? bias[n0 + tid] : 0.0f;
    const float* bv = sbias + wn * 64 + (lane & 7) * 8;
    const float* bvx = sbias + wave * 32 + (lane & 3) * 8;

    uint32_t asrc[MB];
    uint32_t axsrc;
    int s_m0, s_mlim, s_take;
    bool s_extra;
    auto setup = [&]() {  // descriptors + DMA source pointers of the sub-tile that starts at block blk0
        s_take = (cnt % (2 * MB) != 0 && cnt >= 2 * MB + 1) ? 2 * MB + 1 : min(2 * MB, cnt);
        s_extra = s_take == 2 * MB + 1;
        s_m0 = blk0 * 32;
        s_mlim = min(M, (blk0 + s_take) * 32);
#pragma unroll
        for (int t = 0; t < MB; ++t) {
            const int R = 8 * (wave * MB + t) + (lane >> 3);
            const int c = (lane & 7) ^ ((R >> 1) & 7);
            asrc[t] = (uint32_t)(((size_t)min(s_m0 + R, M - 1) * lda + c * 8) * 2);
        }
        const int R = TM + 8 * (wave & 3) + (lane >> 3);
        const int c = (lane & 7) ^ ((R >> 1) & 7);
        axsrc = (uint32_t)(((size_t)min(s_m0 + R, M - 1) * lda + c * 8) * 2);
        blk0 += s_take;
        cnt -= s_take;
    };
    // one 1-KiB LDS-DMA piece of a K-tile (compile-time index: MB A pieces, 4 B pieces, the extra block's piece)
    auto piece = [&](auto ptag, int buf, int k0) __attribute__((always_inline)) {
        constexpr int P = decltype(ptag)::value;
        const unsigned char* abase = reinterpret_cast<const unsigned char*>(A + k0);
        const unsigned char* bbase = reinterpret_cast<const unsigned char*>(B + k0);
        if constexpr (P < MB)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(abase + asrc[P < MB ? P : 0]),
                                             (__attribute__((address_space(3))) void*)(smem + buf * STAGE + (wave * MB + P) * 1024), 16, 0, 0);
        else if constexpr (P < MB + 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bbase + bsrc[P >= MB && P < MB + 4 ? P - MB : 0]),
                                             (__attribute__((address_space(3))) void*)(smem + buf * STAGE + (TM + 32) * 128 + (wave * 4 + P - MB) * 1024), 16, 0, 0);
        else if constexpr (P == MB + 4) {
            if (s_extra && wave < 4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(abase + axsrc),
                                                 (__attribute__((address_space(3))) void*)(smem + buf * STAGE + TM * 128 + wave * 1024), 16, 0, 0);
        }
    };
    auto stage = [&](int buf, int k0) {
        piece(std::integral_constant<int, 0>{}, buf, k0);
        piece(std::integral_constant<int, 1>{}, buf, k0);
        piece(std::integral_constant<int, 2>{}, buf, k0);
        piece(std::integral_constant<int, 3>{}, buf, k0);
        piece(std::integral_constant<int, 4>{}, buf, k0);
        piece(std::integral_constant<int, 5>{}, buf, k0);
        piece(std::integral_constant<int, 6>{}, buf, k0);
        piece(std::integral_constant<int, 7>{}, buf, k0);
        piece(std::integral_constant<int, 8>{}, buf, k0);
    };

    setup();
    stage(0, 0);
    stage(1, 64);          // both buffers are free at the start: the second K-tile's latency overlaps the first one's
    int pend = 0;          // upper bound of the epilogue stores issued after the newest staged K-tile
    bool kt1 = true;       // the second K-tile of the current sub-tile is already staged (here, or ahead of the previous epilogue)
    while (true) {
        const int c_m0 = s_m0, c_mlim = s_mlim, c_take = s_take;
        const bool extra = s_extra;
        const bool more = cnt > 0;
        const int pieces = MB + 4 + ((extra && wave < 4) ? 1 : 0);
        f32x16 acc[MB][2], accx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accx[r] = 0.0f;
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                acc[i][0][r] = 0.0f;
                acc[i][1][r] = 0.0f;
            }
        }
        // accumulate: the old values of C.  One workgroup per CU has nothing else to run while a load is in flight, so they
        // are requested two half-blocks ahead of their use (4 registers per store).  Rows past the sub-tile's end are
        // clamped (masked on the store).  (Touch loads that pull the lines towards the L2 during the K loop, or a deeper
        // window, measured the same or worse: the read-modify-write costs its bytes -- 51 MB more per launch in a phase in
        // which every CU is storing -- not its latency.)
        int e_m0 = 0;  // the sub-tile's first row again, set behind the K loop
        uint4 oldv[2 * MB][2], oldx[2];
        auto fetch_old = [&](auto hbtag) __attribute__((always_inline)) {
            constexpr int hb = decltype(hbtag)::value;
            if constexpr (ACC && hb < 2 * MB) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int q = lane + 64 * it, lr = q >> 3, c8 = q & 7;
                    const int row = min(e_m0 + wm * 32 * MB + hb * 16 + lr, M - 1);
                    oldv[hb][it] = *reinterpret_cast<const uint4*>(C + (size_t)row * ldc + n0 + wn * 64 + c8 * 8);
                }
            } else if constexpr (ACC && hb < 2 * MB + 2) {
                if (extra) {
                    const int row = min(e_m0 + TM + 16 * (hb - 2 * MB) + (lane >> 2), M - 1);
                    oldx[hb - 2 * MB] = *reinterpret_cast<const uint4*>(C + (size_t)row * ldc + n0 + wave * 32 + (lane & 3) * 8);
                }
            } else if constexpr (BCM && hb < 2 * MB) {
#pragma unroll
                for (int it = 0; it < 2; ++it) {  // one pooled value per lane and store: its 8 columns lie in one window (bc_pw % 8 == 0)
                    const int q = lane + 64 * it, lr = q >> 3, c8 = q & 7;
                    const int row = min(e_m0 + wm * 32 * MB + hb * 16 + lr, M - 1);
                    const int cb = n0 + wn * 64 + c8 * 8;
                    oldv[hb][it].x = bc[(size_t)row * bc_ld + cb / bc_pw];
                    if constexpr (BC2) {
                        oldv[hb][it].y = bc[(size_t)row * bc_ld + (cb + 7) / bc_pw];
                        oldv[hb][it].z = (unsigned)((cb / bc_pw + 1) * bc_pw - cb);   // columns of the 8 that belong to the first window
                    }
                }
            } else if constexpr (BCM && hb < 2 * MB + 2) {
                if (extra) {
                    const int row = min(e_m0 + TM + 16 * (hb - 2 * MB) + (lane >> 2), M - 1);
                    const int cb = n0 + wave * 32 + (lane & 3) * 8;
                    oldx[hb - 2 * MB].x = bc[(size_t)row * bc_ld + cb / bc_pw];
                    if constexpr (BC2) {
                        oldx[hb - 2 * MB].y = bc[(size_t)row * bc_ld + (cb + 7) / bc_pw];
                        oldx[hb - 2 * MB].z = (unsigned)((cb / bc_pw + 1) * bc_pw - cb);
                    }
                }
            }
        };
        for (int t = 0; t < nkt; ++t) {
            // K-tile t has landed: this wave's pieces by the counted wait, everyone's by the barrier, which also says that
            // every wave is done reading the other buffer
            if (t == 0) vmwait_upto(kt1 ? pend + pieces : pend);
            else if (t == 1 && kt1) vmwait_upto(pend);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            bool dma = false;
            int dbuf = 0, dk0 = 0;
            if (t + 1 < nkt) {
                if (!(t == 0 && kt1)) { dma = true; dbuf = (t + 1) & 1; dk0 = (t + 1) * 64; }
            } else if (more) {
                setup();       // the descriptors now describe the NEXT sub-tile; its first K-tile goes to buffer 0
                dma = true;
            }
            const unsigned char* sp = smem + (t & 1) * STAGE;
            // (Requesting the fragments of k-step ks + 1 before the MFMAs of k-step ks -- two register sets, 0-9 spilled VGPRs at MB = 4 --
            // measured bit-identical and not a microsecond faster: the LDS latency is already hidden by the SIMD's second wave.  With every
            // A row reading row 0 (SPV_STRIP_LDA0, A always in cache) the K = 8192 shape goes 297 -> 249 us, 37.6 -> 44.9 % of peak: HBM
            // latency against ONE K-tile of prefetch is a fifth of the loop's time, the rest is the CU's own L1 -> LDS -> MFMA chain.)
            auto kstep = [&](auto kstag) __attribute__((always_inline)) {
                constexpr int ks = decltype(kstag)::value;
                const int ch = ((ks * 2 + fh) ^ swz) * 16;
                bf16x8 a[MB], b[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) b[f] = *reinterpret_cast<const bf16x8*>(sp + fb_off + f * 32 * 128 + ch);
#pragma unroll
                for (int f = 0; f < MB; ++f) a[f] = *reinterpret_cast<const bf16x8*>(sp + fa_off + f * 32 * 128 + ch);
                // one DMA piece of the next K-tile behind every MFMA pair (pinned): issued three at a time behind a whole k-step, the
                // pieces of all eight waves reached the 64 B/clk L1 path together and every wave stood 250-400 clocks in issue
                // (in-kernel stamps: k-steps with pieces 620-840 clocks, the one without 390)
                auto pair = [&](auto itag) __attribute__((always_inline)) {
                    constexpr int i = decltype(itag)::value;
                    if constexpr (i < MB) {
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[0], acc[i][0], 0, 0, 0);
                        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[1], acc[i][1], 0, 0, 0);
                        constexpr int P = ks * MB + i;
                        if constexpr (P < MB + 5) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (dma) piece(std::integral_constant<int, P>{}, dbuf, dk0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                };
                pair(std::integral_constant<int, 0>{});
                pair(std::integral_constant<int, 1>{});
                pair(std::integral_constant<int, 2>{});
                pair(std::integral_constant<int, 3>{});
                if (extra) {
                    const bf16x8 ax = *reinterpret_cast<const bf16x8*>(sp + fax_off + ch);
                    const bf16x8 bx = *reinterpret_cast<const bf16x8*>(sp + fbx_off + ch);
                    accx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax, bx, accx, 0, 0, 0);
                }
            };
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});
            kstep(std::integral_constant<int, 2>{});
            kstep(std::integral_constant<int, 3>{});
        }
        e_m0 = c_m0;                    // opaque copy made behind the K loop: hipcc otherwise computes the epilogue's row
        asm volatile("" : "+s"(e_m0));  // addresses before the loop and carries them through it in registers (spills)
        kt1 = false;
        if (PRE1 && more) {
            __builtin_amdgcn_s_barrier();   // every wave is done with buffer 1
            stage(1, 64);
            kt1 = true;
        }
        fetch_old(std::integral_constant<int, 0>{});
        fetch_old(std::integral_constant<int, 1>{});
        if (!PRE1) __syncthreads();         // the epilogue stages through buffer 1
        pend = 0;
        {
            float* st = reinterpret_cast<float*>(smem + EPI_OFF + wave * 4352);
            constexpr int SLD = 68;
            const bool exact = c_take >= 2 * MB && c_mlim == c_m0 + c_take * 32;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            auto pack_store = [&](const float* v, const float* bb, const uint4& old, int row, int col) __attribute__((always_inline)) {
                bf16_t* dst = C + (size_t)row * ldc + col;
                float w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = v[u] + bb[u];
                if constexpr (BC2) {
                    const float t0 = __uint_as_float(old.x << 16) * bc_scale, t1 = __uint_as_float(old.y << 16) * bc_scale;
                    const int first = (int)old.z;
#pragma unroll
                    for (int u = 0; u < 8; ++u) w[u] += u < first ? t0 : t1;
                } else if constexpr (BCM) {
                    const float t = __uint_as_float(old.x << 16) * bc_scale;
#pragma unroll
                    for (int u = 0; u < 8; ++u) w[u] += t;
                }
                if constexpr (ACC) {
                    w[0] += __uint_as_float(old.x << 16); w[1] += __uint_as_float(old.x & 0xffff0000u);
                    w[2] += __uint_as_float(old.y << 16); w[3] += __uint_as_float(old.y & 0xffff0000u);
                    w[4] += __uint_as_float(old.z << 16); w[5] += __uint_as_float(old.z & 0xffff0000u);
                    w[6] += __uint_as_float(old.w << 16); w[7] += __uint_as_float(old.w & 0xffff0000u);
                }
                u32x4 o;
                o.x = pack_bf16x2(w[0], w[1]);
                o.y = pack_bf16x2(w[2], w[3]);
                o.z = pack_bf16x2(w[4], w[5]);
                o.w = pack_bf16x2(w[6], w[7]);
                // s_nop 1: the hardware reads a 16-byte store's data registers a state after issue; hipcc pads nothing inside an asm string
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(o) : "memory");
            };
            auto emit64 = [&](int row0, bool masked, const uint4 (&old)[2]) __attribute__((always_inline)) {  // 16 rows x 64 columns: 2 stores per lane
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int q = lane + 64 * it, lr = q >> 3, c8 = q & 7;
                    const float4 lo = *reinterpret_cast<const float4*>(st + lr * SLD + c8 * 8);
                    const float4 hi = *reinterpret_cast<const float4*>(st + lr * SLD + c8 * 8 + 4);
                    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    if (!masked || row0 + lr < c_mlim) pack_store(v, bv, old[it], row0 + lr, n0 + wn * 64 + c8 * 8);
                }
            };
            auto emit32 = [&](int row0, bool masked, const uint4& old) __attribute__((always_inline)) {  // 16 rows x 32 columns: 1 store per lane
                const int lr = lane >> 2, c8 = lane & 3;
                const float4 lo = *reinterpret_cast<const float4*>(st + lr * SLD + c8 * 8);
                const float4 hi = *reinterpret_cast<const float4*>(st + lr * SLD + c8 * 8 + 4);
                const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (!masked || row0 + lr < c_mlim) pack_store(v, bvx, old, row0 + lr, n0 + wave * 32 + c8 * 8);
            };
            auto half = [&](auto hbtag, auto masked_tag) __attribute__((always_inline)) {
                constexpr int hb = decltype(hbtag)::value;
                constexpr bool MASKED = decltype(masked_tag)::value;
                if constexpr (hb < 2 * MB) {
                    constexpr int i = hb >> 1, h = hb & 1;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 8; ++r)
                            st[((r & 3) + 8 * (r >> 2) + 4 * fh) * SLD + j * 32 + (lane & 31)] = acc[i][j][8 * h + r];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    emit64(e_m0 + wm * 32 * MB + hb * 16, MASKED, oldv[hb]);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                } else if constexpr (hb < 2 * MB + 2) {
                    if (extra) {
                        constexpr int h = hb - 2 * MB;
#pragma unroll
                        for (int r = 0; r < 8; ++r) st[((r & 3) + 8 * (r >> 2) + 4 * fh) * SLD + (lane & 31)] = accx[8 * h + r];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        emit32(e_m0 + TM + 16 * h, MASKED, oldx[h]);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    }
                }
                fetch_old(std::integral_constant<int, hb + 2>{});
            };
            auto run = [&](auto mt) __attribute__((always_inline)) {
                half(std::integral_constant<int, 0>{}, mt);
                half(std::integral_constant<int, 1>{}, mt);
                half(std::integral_constant<int, 2>{}, mt);
                half(std::integral_constant<int, 3>{}, mt);
                half(std::integral_constant<int, 4>{}, mt);
                half(std::integral_constant<int, 5>{}, mt);
                half(std::integral_constant<int, 6>{}, mt);
                half(std::integral_constant<int, 7>{}, mt);
                half(std::integral_constant<int, 8>{}, mt);
                half(std::integral_constant<int, 9>{}, mt);
            };
            if (exact) {
                // every lane issues exactly 4 MB (+2) stores: the count the next waits leave outstanding.  (With
                // accumulate the old values' loads are younger than the staged DMA too, but they have returned by the
                // time their stores issue, so the bound holds.)
                run(std::false_type{});
                pend = 4 * MB + (extra ? 2 : 0);
            } else {
                run(std::true_type{});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if (!more) break;
    }
}

// the uneven shares of the strip kernel: `groups` row groups of base (+1 for the first rem) 32-row blocks; picks the
// wave-row height MB whose sub-tiles (2 MB or 2 MB + 1 blocks) cover base and base + 1 without padding
// CUs the strip kernel leaves alone (spv_set_reserved_cus): a strip workgroup needs a whole CU (2 x 256 VGPRs per SIMD, 155 KiB of
// LDS), so on a CU that holds a resident RCCL channel it cannot start and would run as a second dispatch round.  16 reserved CUs cost
// nothing at the layer shapes (the largest share stays 9 / 13 row blocks).
static int g_reserved_cus = 0;

inline bool strip_plan(int M, int N, int K, int& MB, int& nstrips, int& groups, int& base, int& rem) {
    static const int enabled = SPV_LAB_INT("SPV_GEMM_STRIP", 1);
    if (!enabled || N % 256 != 0 || N > 16384 || K % 128 != 0 || K < 128 || M < 8192) return false;   // (Base width: the MHPermutMix data gradient has N = 9216)
    nstrips = N / 256;
    const int nblk = cdiv(M, 32);
    groups = (256 - g_reserved_cus) / nstrips;
    if (groups < 1) return false;
    if (groups > nblk) groups = nblk;
    base = nblk / groups;
    rem = nblk % groups;
    auto padded = [&](int mb, int c) {  // MFMA block slots a run of c blocks occupies
        int slots = 0;
        while (c > 0) {
            const int take = (c % (2 * mb) != 0 && c >= 2 * mb + 1) ? 2 * mb + 1 : std::min(2 * mb, c);
            slots += take >= 2 * mb ? take : 2 * mb;
            c -= take;
        }
        return slots;
    };
    int best = 0, best_slots = 1 << 30;
    const int worst = base + (rem ? 1 : 0);
    for (int mb = 4; mb >= 2; --mb) {
        const int s = padded(mb, worst);
        if (s < best_slots) { best_slots = s; best = mb; }
    }
    if (best == 0 || best_slots * 100 > worst * 115) return false;  // more than 15 % padding: the 128 x 128 kernels do better
    MB = best;
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// Few-rows NT GEMM (M <= 2048: the CLS-only last layer's feed-forward half, 512 rows): C[M,N] (+)= A[M,K] . B[N,K]^T (+ bias).
// Such a problem is 24 tiles of 128 x 128: the tile kernel needs split-K to reach more than 24 CUs, i.e. a slab workspace and a
// reduce launch -- 9.5 + 5.5 us for 0.4 GFLOP, both at the launch floor.  Here ONE 32 x 32 output tile per 4-wave workgroup (384
// workgroups at 512 x 768), the K range dealt over the four waves and joined in LDS, and no operand staging at all: both operands
// are K-contiguous, so the 16 bytes a lane feeds one v_mfma_f32_32x32x16_bf16 with are 16 contiguous bytes of one row -- a plain
// global_load_dwordx4 from the L2-resident operands, every load of a wave's K share in flight before its first MFMA.
template <typename TO>
__global__ __launch_bounds__(256) void gemm_nt_rows_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, const float* __restrict__ bias,
                                                           TO* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc, int accumulate,
                                                           int tiles_n) {
    __shared__ float red[4][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * 32, n0 = tn * 32;
    const int ksteps = K >> 4;
    const bf16_t* ap = A + (size_t)min(m0 + (lane & 31), M - 1) * lda + (lane >> 5) * 8;   // clamped rows: loads are unconditional
    const bf16_t* bp = B + (size_t)min(n0 + (lane & 31), N - 1) * ldb + (lane >> 5) * 8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    constexpr int U = 8;
    for (int ks = wave; ks < ksteps; ks += 4 * U) {
        uint4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(ks + 4 * u, ksteps - 1);
            a[u] = *reinterpret_cast<const uint4*>(ap + (size_t)k * 16);
            b[u] = *reinterpret_cast<const uint4*>(bp + (size_t)k * 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in = ks + 4 * u < ksteps;   // (a select on the loaded value, not a branch around the load)
            const uint4 av = in ? a[u] : make_uint4(0, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b[u]), acc, 0, 0, 0);
        }
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][lane & 31] = acc[r];
    __syncthreads();
    const int row = tid >> 3, c0 = (tid & 7) * 4;
    const int gr = m0 + row;
    if (gr >= M) return;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = red[0][row][c0 + j] + red[1][row][c0 + j] + red[2][row][c0 + j] + red[3][row][c0 + j];
    TO* cp = C + (size_t)gr * ldc + n0 + c0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int gc = n0 + c0 + j;
        if (gc < N) {
            if (bias != nullptr) v[j] += bias[gc];
            if (accumulate) v[j] += load_out<TO>(cp + j);
        }
    }
    if (n0 + c0 + 3 < N && (ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) {
        if constexpr (sizeof(TO) == 2) {
            uint2 pk;
            pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
            pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(cp) = pk;
        } else {
            *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (n0 + c0 + j < N) store_out<TO>(cp + j, v[j]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// TN contraction: C[M,N] = sum_k A[k][m] * B[k][n]  with A [K, lda>=M] and B [K, ldb>=N] row-major (bf16).
// This is the weight gradient dW = dh^T . x taken straight from the row-major activations: no transposed copies.
// Tiles are staged in LDS exactly as they lie in memory ([k][m] rows of 256 B + 64 B pad, coalesced 16-byte loads);
// the MFMA fragments (8 consecutive k for one m / one n per lane) come out of ds_read_b64_tr_b16, the gfx950
// transposing LDS read: per 16-lane group it reads a 4 (k) x 16 (m) block and hands lane i column i.  With the
// 320-byte row stride the four k rows of a group land in four disjoint 16-bank ranges (conflict free).
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef unsigned tn_u32x4 __attribute__((ext_vector_type(4)));
#ifdef SPV_LAB
__device__ int g_tn_ablate = 0;   // set by the host from SPV_TN_ABLATE before a launch (lab build only)
#endif
constexpr int TROWB = 256 + 64;  // LDS bytes per k row (128 bf16 + pad)
constexpr int TBK = 64;          // k rows per stage (16 MFMAs per wave between barriers)

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* base) {
    // base: this lane's address for rows k0..k0+3; the second read covers rows k0+4..k0+7
    using lds_ptr = s16x4 __attribute__((address_space(3)))*;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + 4 * TROWB));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// one 128 x 128 tile of one K-slice (the body of gemm_tn_kernel and of the batched form below)
template <typename TO, int DEPTH>
__device__ __forceinline__ void tn_tile_body(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, TO* __restrict__ C,
                                             float* __restrict__ ws, int M, int N, int K, int lda, int ldb, int ldc, int k_per_split,
                                             int accumulate, int tiles_n, int tile, int split, int tiles_m = 0) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TBK * TROWB];  // 40 960 B (>= the epilogue's 36 864 B)
    unsigned char* sA = smem;
    unsigned char* sB = smem + TBK * TROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // tile order.  Few column tiles (the layer gradients, 6 x 4): row-major.  Many more column tiles than row tiles (the MHPermutMix
    // gradient [512, 8192, 33280]: 4 x 64): the row tiles of one column are NEIGHBOURS, so the four workgroups that read the same
    // 8.5 MB panel of the gathered matrix run on one XCD at the same time and three of them hit in its L2 -- row-major put them 64 tiles
    // apart, i.e. on different XCDs, and the 545 MB matrix was fetched once per row tile (PMC: 2.25 GB per launch against 0.6 GB).
    const bool m_fast = tiles_m > 0 && tiles_n >= 4 * tiles_m;
    const int tm = m_fast ? tile % tiles_m : tile / tiles_n, tn = m_fast ? tile / tiles_m : tile % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);

    // staging map: TBK k rows x 16 chunks of 16 B per operand -> TBK / 16 chunks per thread per operand
    const int srow = tid >> 4, sch = tid & 15;  // rows srow + 16 i
    const bool a_ok = (m0 + sch * 8) < M, b_ok = (n0 + sch * 8) < N;  // M, N multiples of 8: a chunk is all in or all out
    const bool edge = m0 + BM > M || n0 + BN > N;                     // uniform: some lane's chunk lies outside the matrix
    // Three register sets: the loads of K-tiles t+1 .. t+3 are in flight while tile t is multiplied.  With the weight
    // gradient's 288 workgroups (~1 per CU, one wave per SIMD) a K-tile is 16 MFMAs = 0.27 us of work per wave, far less
    // than an HBM round trip: with one tile in flight the kernel ran at one K-tile per load latency (43 tiles x ~0.95 us).
    // The loads of the pipelined loop are UNCONDITIONAL (see tn_wide_body: predicated loads cost the counted waits): a lane whose
    // chunk lies outside the matrix reads column 0 instead and zeroes the value before the LDS store; a prefetch past the slice's
    // last full tile is clamped to that tile; a partial last tile takes the predicated loads behind the loop.
    typedef tn_u32x4 RegTile[TBK / 16];
    RegTile ra0, rb0, ra1, rb1, ra2, rb2;
    uint32_t aoff[TBK / 16], boff[TBK / 16];
#pragma unroll
    for (int i = 0; i < TBK / 16; ++i) {
        aoff[i] = (uint32_t)(((size_t)(srow + 16 * i) * lda + (a_ok ? m0 + sch * 8 : 0)) * 2);
        boff[i] = (uint32_t)(((size_t)(srow + 16 * i) * ldb + (b_ok ? n0 + sch * 8 : 0)) * 2);
    }
    const int nfull = (kend - kbeg) / TBK, krem = (kend - kbeg) % TBK;
    auto load_full = [&](int t, RegTile& ra, RegTile& rb) __attribute__((always_inline)) {
        const int k0 = kbeg + min(t, nfull - 1) * TBK;   // uniform
        const unsigned char* ab = reinterpret_cast<const unsigned char*>(A + (size_t)k0 * lda);
        const unsigned char* bb = reinterpret_cast<const unsigned char*>(B + (size_t)k0 * ldb);
#pragma unroll
        for (int i = 0; i < TBK / 16; ++i) {
            ra[i] = *reinterpret_cast<const tn_u32x4*>(ab + aoff[i]);
            rb[i] = *reinterpret_cast<const tn_u32x4*>(bb + boff[i]);
        }
    };
    auto load_partial = [&](int k0, RegTile& ra, RegTile& rb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TBK / 16; ++i) {
            const int k = k0 + srow + 16 * i;
            const bool kin = k < kend;
            ra[i] = (kin && a_ok) ? *reinterpret_cast<const tn_u32x4*>(A + (size_t)k * lda + m0 + sch * 8) : tn_u32x4{0u, 0u, 0u, 0u};
            rb[i] = (kin && b_ok) ? *reinterpret_cast<const tn_u32x4*>(B + (size_t)k * ldb + n0 + sch * 8) : tn_u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto store_tile = [&](RegTile& ra, RegTile& rb) __attribute__((always_inline)) {
        if (edge) {
#pragma unroll
            for (int i = 0; i < TBK / 16; ++i) {
                if (!a_ok) ra[i] = tn_u32x4{0u, 0u, 0u, 0u};
                if (!b_ok) rb[i] = tn_u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int i = 0; i < TBK / 16; ++i) {
            *reinterpret_cast<tn_u32x4*>(sA + (srow + 16 * i) * TROWB + sch * 16) = ra[i];
            *reinterpret_cast<tn_u32x4*>(sB + (srow + 16 * i) * TROWB + sch * 16) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // transposing-read lane geometry: 16-lane group g = lane >> 4 covers columns 16 (g & 1) + [0,16) of the 32-wide
    // MFMA block and k half h = g >> 1; inside the group lane 4 q + p supplies row q, columns 4 p .. 4 p + 3
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int frag_off = (8 * (g >> 1) + q) * TROWB + (16 * (g & 1) + 4 * pp) * 2;
    const unsigned char* fa0 = sA + frag_off + (wm * 64) * 2;
    const unsigned char* fb0 = sB + frag_off + (wn * 64) * 2;
    auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < TBK / 16; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                a[f] = tr_frag(fa0 + ks * 16 * TROWB + f * 64);
                b[f] = tr_frag(fb0 + ks * 16 * TROWB + f * 64);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
#ifdef SPV_LAB
    const int abl = g_tn_ablate;   // lab: 1 = no LDS stores, 2 = no global loads after the prologue, 4 = no MFMAs / fragment reads (wrong results)
#else
    constexpr int abl = 0;
#endif
    // one K-tile: its registers go to LDS, the set is refilled with full tile `next` (DEPTH ahead), then the MFMAs
    auto step = [&](int next, RegTile& ra, RegTile& rb) __attribute__((always_inline)) {
        if (!(abl & 1)) store_tile(ra, rb);
        __syncthreads();
        if (!(abl & 2)) load_full(next, ra, rb);
        if (!(abl & 4)) multiply();
        __syncthreads();
    };
    auto step_last = [&](RegTile& ra, RegTile& rb) __attribute__((always_inline)) {   // no refill
        if (!(abl & 1)) store_tile(ra, rb);
        __syncthreads();
        if (!(abl & 4)) multiply();
        __syncthreads();
    };

    if constexpr (DEPTH == 3) {
        if (nfull > 0) {
            load_full(0, ra0, rb0);
            load_full(1, ra1, rb1);
            load_full(2, ra2, rb2);
            int t = 0;
            for (; t + 3 <= nfull; t += 3) {
                step(t + 3, ra0, rb0);
                step(t + 4, ra1, rb1);
                step(t + 5, ra2, rb2);
            }
            if (t < nfull) {
                step_last(ra0, rb0);
                if (t + 1 < nfull) step_last(ra1, rb1);
            }
        }
    } else {   // one K-tile in flight
        for (int t = 0; t < nfull; ++t) {
            load_full(t, ra0, rb0);
            step_last(ra0, rb0);
        }
    }
    if (krem > 0) {
        load_partial(kbeg + nfull * TBK, ra0, rb0);
        step_last(ra0, rb0);
    }
    store_acc_tile<TO>(acc, smem, nullptr, C, ws, M, N, ldc, accumulate, m0, n0, split, 0, 0, 0, nullptr);
}

template <typename TO, int DEPTH>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                      TO* __restrict__ C, float* __restrict__ ws, int M, int N, int K, int lda,
                                                      int ldb, int ldc, int k_per_split, int accumulate, int tiles_n,
                                                      int tiles_mn, int nsplit) {
    // one linear id over (split, tile), XCD-remapped as a whole: each XCD gets whole K-slices, so the tiles that
    // share a slice's operand rows hit in that XCD's L2 (per-slice round-robin made every XCD fetch every slice:
    // 3.4x the algorithmic HBM reads on the weight-gradient GEMM, FETCH_SIZE)
    const int lin = xcd_remap(blockIdx.x, tiles_mn * nsplit);
    tn_tile_body<TO, DEPTH>(A, B, C, ws, M, N, K, lda, ldb, ldc, k_per_split, accumulate, tiles_n, lin % tiles_mn, lin / tiles_mn, tiles_mn / tiles_n);
}

// Several weight gradients with the same K (the rows of the batch) in ONE launch: the layer weight gradients are 24 tiles each, so
// alone each needs ~21 K-slices to fill the chip (a 25-K-tile loop per workgroup, 33 MB of partial sums written and re-read);
// eight of them together fill it with 5 slices of 104 K-tiles.  Isolated: 8 x 53 us -> 307 us (emulated as one 768 x 4096 x 33280
// problem).  Only for gradients nobody reads before the backward pass ends (hip_ops._held_wgrads).
constexpr int TNB_MAX = 8;
struct TnBatch {
    int nprob;
    int nlong;                    // problems [0, nlong) reduce over the call's K rows in nsplit slices; the rest over their own (fewer) rows, unsplit
    int first_tile[TNB_MAX + 1];
    long long ws_off[TNB_MAX];   // floats into the workspace: nsplit slabs of M x N per LONG problem, problem after problem
    struct P {
        const bf16_t* A;
        const bf16_t* B;
        float* C;
        int M, N, lda, ldb, ldc, tiles_n, K;
    } p[TNB_MAX];
};
// which (problem, tile, K-slice) this workgroup computes: the long problems' tiles slice after slice (XCD-remapped as a whole, see
// gemm_tn_kernel), then the short problems' tiles (the CLS-only last layer's 512-row gradients: one short workgroup each instead of a
// launch + a reduce of their own behind the big one)
struct TnWork {
    int j, tile, split, K, k_per_split;
    float* ws;
};
__device__ __forceinline__ TnWork tn_batch_work(const TnBatch& tb, float* ws, int K, int k_per_split, int nsplit) {
    // The short tiles take the FIRST workgroup ids (padded to a multiple of 8 so that the long part keeps its XCD mapping; the pad
    // workgroups leave at once): a short workgroup is a tenth of a long one, and dispatched first it only delays the CUs it lands on
    // by that much -- appended behind 504 long workgroups (two full rounds at one per CU) the 24 short ones pushed 16 long workgroups
    // into a third round (+35 us per launch).
    const int long_tiles = tb.first_tile[tb.nlong];
    const int short_tiles = tb.first_tile[tb.nprob] - long_tiles;
    const int short_pad = (short_tiles + 7) & ~7;
    int split = 0, t;
    if ((int)blockIdx.x < short_pad) {
        if ((int)blockIdx.x >= short_tiles) return {-1, 0, 0, 0, 0, nullptr};
        t = long_tiles + (int)blockIdx.x;
    } else {
        const int lin = xcd_remap(blockIdx.x - short_pad, long_tiles * nsplit);
        split = lin / long_tiles;
        t = lin % long_tiles;
    }
    int j = 0;
    while (j + 1 < tb.nprob && t >= tb.first_tile[j + 1]) ++j;   // workgroup-uniform
    const bool is_long = j < tb.nlong;
    return {j, t - tb.first_tile[j], split, is_long ? K : tb.p[j].K, is_long ? k_per_split : tb.p[j].K, is_long ? ws + tb.ws_off[j] : nullptr};
}
template <int DEPTH>
__global__ __launch_bounds__(256) void gemm_tn_batch_kernel(TnBatch tb, float* __restrict__ ws, int K, int k_per_split, int nsplit) {
    const TnWork w = tn_batch_work(tb, ws, K, k_per_split, nsplit);
    if (w.j < 0) return;
    const TnBatch::P& q = tb.p[w.j];
    tn_tile_body<float, DEPTH>(q.A, q.B, q.C, w.ws, q.M, q.N, w.K, q.lda, q.ldb, q.ldc, w.k_per_split, 0, q.tiles_n, w.tile, w.split);
}

#ifdef SPV_LAB   // measured 5 % slower than the one-buffer kernel: lab build only
// The same 128 x 128 tile on TWO LDS buffers and ONE barrier per K-tile.  tn_tile_body above stores a K-tile, waits at a barrier,
// multiplies, waits at a second barrier: inside a workgroup nothing overlaps, and its ablations put the sum of
// exposed loads, LDS stores and MFMAs at the kernel's time.  Here the registers of K-tile t + 1 go to the OTHER buffer between the
// k-steps of tile t (the store issue hides under that tile's MFMAs), the set is refilled with tile t + 4, and one barrier both
// publishes tile t + 1 and frees tile t's buffer.  81 920 B of LDS: two workgroups per CU, as many as the 192 registers allow anyway.
template <typename TO>
__device__ __forceinline__ void tn_tile_body_db(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, TO* __restrict__ C,
                                                float* __restrict__ ws, int M, int N, int K, int lda, int ldb, int ldc, int k_per_split,
                                                int accumulate, int tiles_n, int tile, int split, int tiles_m = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tdb_smem[];   // 2 x 40 960 B
    constexpr int BUF = 2 * TBK * TROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const bool m_fast = tiles_m > 0 && tiles_n >= 4 * tiles_m;
    const int tm = m_fast ? tile % tiles_m : tile / tiles_n, tn = m_fast ? tile / tiles_m : tile % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    const int srow = tid >> 4, sch = tid & 15;
    const bool a_ok = (m0 + sch * 8) < M, b_ok = (n0 + sch * 8) < N;
    typedef uint4 RegTile[TBK / 16];
    RegTile ra0, rb0, ra1, rb1, ra2, rb2;
    auto load_tile = [&](int k0, RegTile& ra, RegTile& rb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TBK / 16; ++i) {
            const int k = k0 + srow + 16 * i;
            const bool kin = k < kend;
            ra[i] = (kin && a_ok) ? *reinterpret_cast<const uint4*>(A + (size_t)k * lda + m0 + sch * 8) : make_uint4(0, 0, 0, 0);
            rb[i] = (kin && b_ok) ? *reinterpret_cast<const uint4*>(B + (size_t)k * ldb + n0 + sch * 8) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_half = [&](unsigned char* buf, const RegTile& ra, const RegTile& rb, int half) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 2 * half; i < 2 * half + 2; ++i) {
            *reinterpret_cast<uint4*>(buf + (srow + 16 * i) * TROWB + sch * 16) = ra[i];
            *reinterpret_cast<uint4*>(buf + TBK * TROWB + (srow + 16 * i) * TROWB + sch * 16) = rb[i];
        }
    };
    static_assert(TBK / 16 == 4, "two halves of two row groups");
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const int frag_off = (8 * (g >> 1) + q) * TROWB + (16 * (g & 1) + 4 * pp) * 2;
    const int fa_off = frag_off + (wm * 64) * 2, fb_off = TBK * TROWB + frag_off + (wn * 64) * 2;
    // one K-tile: multiply from `cur`; between its k-steps the registers of the NEXT tile go to `nxt`; then that set is refilled
    auto step = [&](int k0, const unsigned char* cur, unsigned char* nxt, RegTile& ra, RegTile& rb) __attribute__((always_inline)) {
        const bool more = k0 + TBK < kend;       // (ra, rb) hold tile k0 + TBK
#pragma unroll
        for (int ks = 0; ks < TBK / 16; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                a[f] = tr_frag(cur + fa_off + ks * 16 * TROWB + f * 64);
                b[f] = tr_frag(cur + fb_off + ks * 16 * TROWB + f * 64);
            }
            if (ks < 2 && more) store_half(nxt, ra, rb, ks);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (k0 + 4 * TBK < kend) load_tile(k0 + 4 * TBK, ra, rb);
        __syncthreads();
    };
    load_tile(kbeg, ra0, rb0);
    if (kbeg + TBK < kend) load_tile(kbeg + TBK, ra1, rb1);
    if (kbeg + 2 * TBK < kend) load_tile(kbeg + 2 * TBK, ra2, rb2);
    store_half(tdb_smem, ra0, rb0, 0);
    store_half(tdb_smem, ra0, rb0, 1);
    if (kbeg + 3 * TBK < kend) load_tile(kbeg + 3 * TBK, ra0, rb0);
    __syncthreads();
    // tile t is multiplied from buffer t & 1; the register sets rotate with period 3: (ra1, ra2, ra0) hold tiles t + 1 for t = 0, 1, 2
    int t = 0;
    for (int k0 = kbeg; k0 < kend; k0 += 3 * TBK, t += 3) {
        step(k0, tdb_smem + (t & 1) * BUF, tdb_smem + ((t + 1) & 1) * BUF, ra1, rb1);
        if (k0 + TBK < kend) step(k0 + TBK, tdb_smem + ((t + 1) & 1) * BUF, tdb_smem + (t & 1) * BUF, ra2, rb2);
        if (k0 + 2 * TBK < kend) step(k0 + 2 * TBK, tdb_smem + (t & 1) * BUF, tdb_smem + ((t + 1) & 1) * BUF, ra0, rb0);
    }
    store_acc_tile<TO>(acc, tdb_smem, nullptr, C, ws, M, N, ldc, accumulate, m0, n0, split, 0, 0, 0, nullptr);
}

__global__ __launch_bounds__(256) void gemm_tn_batch_db_kernel(TnBatch tb, float* __restrict__ ws, int K, int k_per_split, int nsplit) {
    const TnWork w = tn_batch_work(tb, ws, K, k_per_split, nsplit);
    if (w.j < 0) return;
    const TnBatch::P& q = tb.p[w.j];
    tn_tile_body_db<float>(q.A, q.B, q.C, w.ws, q.M, q.N, w.K, q.lda, q.ldb, q.ldc, w.k_per_split, 0, q.tiles_n, w.tile, w.split);
}

#endif  // SPV_LAB

// ---------------------------------------------------------------------------------------------------------
// TN contraction, WIDE tile: 256 (m) x 128 (n) per 8-wave workgroup (waves 4 x 2, 64 x 64 each), otherwise gemm_tn_kernel's
// structure (register-staged tiles three K-tiles deep, one LDS buffer, transposing fragment reads).  The weight gradients of the
// encoder layers (768 x 512 x 33 280) are bound by the L2 -> LDS path, not by MFMA issue: a 128 x 128 tile moves 32 KiB per
// 2.1 MFLOP K-tile (400 MB per launch through a path that sustains ~14 TB/s chip-wide), the 256 x 128 tile 48 KiB per 4.2 MFLOP
// (300 MB) at the same number of split-K slabs (12 tiles x 21 slices = one workgroup per CU).
constexpr int TWM = 256;              // tile rows (m)
constexpr int TWROWA = TWM * 2 + 64;  // LDS bytes per k row of the A tile: 576 = 64 mod 256, the four k rows of a read group land in disjoint banks

template <typename TO, bool DB>
__device__ __forceinline__ void tn_wide_body(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, TO* __restrict__ C,
                                             float* __restrict__ ws, int M, int N, int K, int lda, int ldb, int ldc,
                                             int k_per_split, int accumulate, int tiles_n, int tile, int split) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tw_smem[];  // TW_SMEM / TW_SMEM_DB bytes
    unsigned char* sA = tw_smem;
    unsigned char* sB = tw_smem + TBK * TWROWA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = (tile / tiles_n) * TWM, n0 = (tile % tiles_n) * BN;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);

    // staging map: A = 64 k rows x 32 chunks of 16 B -> 4 per thread (rows arow + 16 i); B = 64 x 16 chunks -> 2 per thread.
    // Addresses = a uniform K-tile base (SGPRs) + a per-lane 32-bit offset inside the tile.  The loads of the pipelined loop are
    // UNCONDITIONAL straight-line code: with every load in its own `k < kend ? load : 0` branch (rounds 1-3) the compiler could not
    // count the loads in flight and put `s_waitcnt vmcnt(0)` in front of every LDS store -- the "three K-tiles in flight" were one
    // multiply phase of prefetch, and the kernel ran at load latency + LDS stores + MFMAs added up (ISA + ablations).
    // A prefetch past the slice's last full tile is clamped to that tile (in bounds, never consumed); a partial last tile (K-slices
    // that are not multiples of 64 rows) takes the predicated loads, once, behind the loop.
    const int arow = tid >> 5, ach = tid & 31, brow = tid >> 4, bch = tid & 15;
    typedef tn_u32x4 RegA[4];   // native vectors: a tile of HIP uint4 structs that is only copied in and out stays an alloca (scratch)
    typedef tn_u32x4 RegB[2];
    RegA a0, a1, a2;
    RegB b0, b1, b2;
    uint32_t aoff[4], boff[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) aoff[i] = (uint32_t)(((size_t)(arow + 16 * i) * lda + ach * 8) * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) boff[i] = (uint32_t)(((size_t)(brow + 32 * i) * ldb + bch * 8) * 2);
    const int nfull = (kend - kbeg) / TBK, krem = (kend - kbeg) % TBK;
    auto load_full = [&](int t, RegA& ra, RegB& rb) __attribute__((always_inline)) {
        const int k0 = kbeg + min(t, nfull - 1) * TBK;   // uniform
        const unsigned char* ab = reinterpret_cast<const unsigned char*>(A + (size_t)k0 * lda + m0);
        const unsigned char* bb = reinterpret_cast<const unsigned char*>(B + (size_t)k0 * ldb + n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const tn_u32x4*>(ab + aoff[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = *reinterpret_cast<const tn_u32x4*>(bb + boff[i]);
    };
    auto load_partial = [&](int k0, RegA& ra, RegB& rb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + arow + 16 * i;
            ra[i] = k < kend ? *reinterpret_cast<const tn_u32x4*>(A + (size_t)k * lda + m0 + ach * 8) : tn_u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + brow + 32 * i;
            rb[i] = k < kend ? *reinterpret_cast<const tn_u32x4*>(B + (size_t)k * ldb + n0 + bch * 8) : tn_u32x4{0u, 0u, 0u, 0u};
        }
    };
    // the six 16-byte LDS stores of a register tile; part ks of 4 (2 + 2 + 1 + 1 stores) or all of them (ks < 0)
    auto store_part = [&](int boff_lds, const RegA& ra, const RegB& rb, int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ks < 0 || ks == (i >> 1)) *reinterpret_cast<tn_u32x4*>(sA + boff_lds + (arow + 16 * i) * TWROWA + ach * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (ks < 0 || ks == 2 + i) *reinterpret_cast<tn_u32x4*>(sB + boff_lds + (brow + 32 * i) * TROWB + bch * 16) = rb[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const unsigned char* fa0 = sA + (8 * (g >> 1) + q) * TWROWA + (16 * (g & 1) + 4 * pp) * 2 + (wm * 64) * 2;
    const unsigned char* fb0 = sB + (8 * (g >> 1) + q) * TROWB + (16 * (g & 1) + 4 * pp) * 2 + (wn * 64) * 2;
    using lds_ptr = s16x4 __attribute__((address_space(3)))*;
    auto fragA = [&](const unsigned char* p) __attribute__((always_inline)) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * TWROWA));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
#ifdef SPV_LAB
    const int abl = g_tn_ablate;   // lab: 1 = no LDS stores, 2 = no global loads after the prologue, 4 = no MFMAs / fragment reads (wrong results)
#else
    constexpr int abl = 0;
#endif
    // the 16 MFMAs of the K-tile in buffer `cur`; between its k-steps (STORE) the register tile (ra, rb) goes to buffer `nxt`
    auto multiply = [&](int cur, auto store_tag, int nxt, const RegA& ra, const RegB& rb) __attribute__((always_inline)) {
        constexpr bool STORE = decltype(store_tag)::value;
#pragma unroll
        for (int ks = 0; ks < TBK / 16; ++ks) {
            bf16x8 a[2], b[2];
            if (!(abl & 4)) {
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    a[f] = fragA(fa0 + cur + ks * 16 * TWROWA + f * 64);
                    b[f] = tr_frag(fb0 + cur + ks * 16 * TROWB + f * 64);
                }
            }
            if constexpr (STORE) {
                if (!(abl & 1)) store_part(nxt, ra, rb, ks);
            }
            if (!(abl & 4)) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    using yes = std::true_type;
    using no = std::false_type;

    if constexpr (DB) {
        // TWO LDS buffers, ONE barrier per K-tile: while tile t is multiplied out of buffer t & 1, the registers of tile t + 1 go to
        // the other buffer between the k-steps (the 13-cycle ds_write_b128 issues hide under the MFMAs), that register set is
        // refilled with tile t + 4, and the barrier both publishes tile t + 1 and frees tile t's buffer.  Register sets rotate with
        // period 3 (tile j lives in set j % 3), buffers with period 2.
        constexpr int WBUF = TBK * TWROWA + TBK * TROWB;
        if (nfull > 0) {
            load_full(0, a0, b0);
            load_full(1, a1, b1);
            load_full(2, a2, b2);
            store_part(0, a0, b0, -1);
            load_full(3, a0, b0);
            __syncthreads();
            auto step = [&](int t, RegA& ra, RegB& rb) __attribute__((always_inline)) {   // (ra, rb) hold tile t + 1
                multiply((t & 1) * WBUF, yes{}, ((t + 1) & 1) * WBUF, ra, rb);
                if (!(abl & 2)) load_full(t + 4, ra, rb);
                __syncthreads();
            };
            auto step_noload = [&](int t, const RegA& ra, const RegB& rb) __attribute__((always_inline)) {
                multiply((t & 1) * WBUF, yes{}, ((t + 1) & 1) * WBUF, ra, rb);
                __syncthreads();
            };
            int t = 0;
            for (; t + 4 <= nfull; t += 3) {
                step(t, a1, b1);
                step(t + 1, a2, b2);
                step(t + 2, a0, b0);
            }
            const int left = nfull - t;   // 1, 2 or 3 tiles: t in buffer t & 1, t + 1 in (a1, b1), t + 2 in (a2, b2)
            if (left >= 2) {
                step_noload(t, a1, b1);
                if (left == 3) step_noload(t + 1, a2, b2);
            }
            multiply(((nfull - 1) & 1) * WBUF, no{}, 0, a0, b0);
            __syncthreads();
        }
        if (krem > 0) {
            load_partial(kbeg + nfull * TBK, a0, b0);
            store_part(0, a0, b0, -1);
            __syncthreads();
            multiply(0, no{}, 0, a0, b0);
            __syncthreads();
        }
    } else {
        // one buffer, two barriers per K-tile: store, barrier, refill, multiply, barrier
        auto step = [&](int next, RegA& ra, RegB& rb) __attribute__((always_inline)) {
            if (!(abl & 1)) store_part(0, ra, rb, -1);
            __syncthreads();
            if (!(abl & 2)) load_full(next, ra, rb);
            multiply(0, no{}, 0, ra, rb);
            __syncthreads();
        };
        auto step_last = [&](const RegA& ra, const RegB& rb) __attribute__((always_inline)) {   // no refill
            if (!(abl & 1)) store_part(0, ra, rb, -1);
            __syncthreads();
            multiply(0, no{}, 0, ra, rb);
            __syncthreads();
        };
        if (nfull > 0) {
            load_full(0, a0, b0);
            load_full(1, a1, b1);
            load_full(2, a2, b2);
            int t = 0;
            for (; t + 3 <= nfull; t += 3) {
                step(t + 3, a0, b0);
                step(t + 4, a1, b1);
                step(t + 5, a2, b2);
            }
            if (t < nfull) {
                step_last(a0, b0);
                if (t + 1 < nfull) step_last(a1, b1);
            }
        }
        if (krem > 0) {
            load_partial(kbeg + nfull * TBK, a0, b0);
            step_last(a0, b0);
        }
    }
    store_acc_tile<TO>(acc, tw_smem, nullptr, C, ws, M, N, ldc, accumulate, m0, n0, split, 0, 0, 0, nullptr, nullptr, 0, 0, wm * 64, wn * 64);
}

constexpr int TW_SMEM = 8 * 9216;                                   // one K-tile buffer (57 344 B) < the epilogue's 8 x 9 216 B
constexpr int TW_SMEM_DB = 2 * (TBK * TWROWA + TBK * TROWB);       // two K-tile buffers: 114 688 B
template <typename TO, bool DB>
__global__ __launch_bounds__(512) void gemm_tn_wide_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, TO* __restrict__ C,
                                                           float* __restrict__ ws, int M, int N, int K, int lda, int ldb, int ldc,
                                                           int k_per_split, int accumulate, int tiles_n, int tiles_mn, int nsplit) {
    const int lin = xcd_remap(blockIdx.x, tiles_mn * nsplit);  // whole K-slices per XCD (see gemm_tn_kernel)
    tn_wide_body<TO, DB>(A, B, C, ws, M, N, K, lda, ldb, ldc, k_per_split, accumulate, tiles_n, lin % tiles_mn, lin / tiles_mn);
}

// ---------------------------------------------------------------------------------------------------------
// TN contraction, TALL tile: 512 (m) x 128 (n) per 8-wave workgroup (waves 4 x 2, 128 x 64 each: 128 accumulator registers), 32-row
// K-tiles, three register sets of operand rows, two LDS buffers, one barrier per K-tile.  For gradients with M = 512 and a long N -- the
// MHPermutMix linear's [512, 8192, 33 280] --, where the 128 x 128 kernel's four row tiles of a column block are four workgroups that
// each stream the same 8.5 MB panel of the gathered matrix and drift apart by more K-tiles than the L2 holds (PMC: 1.47 GB fetched per
// launch against 0.58 GB of operands: the kernel ran at 4.2 TB/s of HBM).  Here ONE workgroup owns all of M: the panel is read once.
constexpr int TTM = 512;
constexpr int TTK = 32;
constexpr int TTROWA = TTM * 2 + 64;   // 1088 = 64 mod 256: the four k rows of a transposing read land in disjoint banks
constexpr int TT_SMEM = 2 * (TTK * TTROWA + TTK * TROWB);   // 90 112 B (>= the epilogue's 8 x 9 216 B)
template <typename TO>
__global__ __launch_bounds__(512) void gemm_tn_tall_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, TO* __restrict__ C,
                                                           float* __restrict__ ws, int M, int N, int K, int lda, int ldb, int ldc,
                                                           int k_per_split, int accumulate, int tiles_n, int tiles_mn, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tt_smem[];
    constexpr int WBUF = TTK * TTROWA + TTK * TROWB;
    unsigned char* sA = tt_smem;
    unsigned char* sB = tt_smem + TTK * TTROWA;
    const int lin = xcd_remap(blockIdx.x, tiles_mn * nsplit);   // whole K-slices per XCD: the slice's A rows (dh) hit in that L2
    const int tile = lin % tiles_mn, split = lin / tiles_mn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = (tile / tiles_n) * TTM, n0 = (tile % tiles_n) * BN;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    // staging: A = 32 k rows x 64 chunks of 16 B -> 4 per thread (rows arow + 8 i); B = 32 x 16 chunks -> 1 per thread
    const int arow = tid >> 6, ach = tid & 63, brow = tid >> 4, bch = tid & 15;
    typedef tn_u32x4 RegA[4];
    RegA a0, a1, a2;
    tn_u32x4 b0, b1, b2;
    uint32_t aoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) aoff[i] = (uint32_t)(((size_t)(arow + 8 * i) * lda + ach * 8) * 2);
    const uint32_t boff = (uint32_t)(((size_t)brow * ldb + bch * 8) * 2);
    const int nfull = (kend - kbeg) / TTK, krem = (kend - kbeg) % TTK;
    auto load_full = [&](int t, RegA& ra, tn_u32x4& rb) __attribute__((always_inline)) {
        const int k0 = kbeg + min(t, nfull - 1) * TTK;   // uniform; a prefetch past the last full tile re-reads it (never consumed)
        const unsigned char* ab = reinterpret_cast<const unsigned char*>(A + (size_t)k0 * lda + m0);
        const unsigned char* bb = reinterpret_cast<const unsigned char*>(B + (size_t)k0 * ldb + n0);
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const tn_u32x4*>(ab + aoff[i]);
        rb = *reinterpret_cast<const tn_u32x4*>(bb + boff);
    };
    auto load_partial = [&](int k0, RegA& ra, tn_u32x4& rb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + arow + 8 * i;
            ra[i] = k < kend ? *reinterpret_cast<const tn_u32x4*>(A + (size_t)k * lda + m0 + ach * 8) : tn_u32x4{0u, 0u, 0u, 0u};
        }
        const int k = k0 + brow;
        rb = k < kend ? *reinterpret_cast<const tn_u32x4*>(B + (size_t)k * ldb + n0 + bch * 8) : tn_u32x4{0u, 0u, 0u, 0u};
    };
    // the five 16-byte LDS stores of a register tile: part ks of 2 (3 + 2 stores) or all of them (ks < 0)
    auto store_part = [&](int buf, const RegA& ra, const tn_u32x4& rb, int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ks < 0 || ks == (i >= 3 ? 1 : 0)) *reinterpret_cast<tn_u32x4*>(sA + buf + (arow + 8 * i) * TTROWA + ach * 16) = ra[i];
        if (ks < 0 || ks == 1) *reinterpret_cast<tn_u32x4*>(sB + buf + brow * TROWB + bch * 16) = rb;
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    const unsigned char* fa0 = sA + (8 * (g >> 1) + q) * TTROWA + (16 * (g & 1) + 4 * pp) * 2 + (wm * 128) * 2;
    const unsigned char* fb0 = sB + (8 * (g >> 1) + q) * TROWB + (16 * (g & 1) + 4 * pp) * 2 + (wn * 64) * 2;
    using lds_ptr = s16x4 __attribute__((address_space(3)))*;
    auto fragA = [&](const unsigned char* p) __attribute__((always_inline)) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * TTROWA));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    auto multiply = [&](int cur, auto store_tag, int nxt, const RegA& ra, const tn_u32x4& rb) __attribute__((always_inline)) {
        constexpr bool STORE = decltype(store_tag)::value;
#pragma unroll
        for (int ks = 0; ks < TTK / 16; ++ks) {
            bf16x8 a[4], b[2];
#pragma unroll
            for (int f = 0; f < 4; ++f) a[f] = fragA(fa0 + cur + ks * 16 * TTROWA + f * 64);
#pragma unroll
            for (int f = 0; f < 2; ++f) b[f] = tr_frag(fb0 + cur + ks * 16 * TROWB + f * 64);
            if constexpr (STORE) store_part(nxt, ra, rb, ks);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    using yes = std::true_type;
    using no = std::false_type;
    if (nfull > 0) {
        load_full(0, a0, b0);
        load_full(1, a1, b1);
        load_full(2, a2, b2);
        store_part(0, a0, b0, -1);
        load_full(3, a0, b0);
        __syncthreads();
        auto step = [&](int t, RegA& ra, tn_u32x4& rb) __attribute__((always_inline)) {   // (ra, rb) hold tile t + 1
            multiply((t & 1) * WBUF, yes{}, ((t + 1) & 1) * WBUF, ra, rb);
            load_full(t + 4, ra, rb);
            __syncthreads();
        };
        auto step_noload = [&](int t, const RegA& ra, const tn_u32x4& rb) __attribute__((always_inline)) {
            multiply((t & 1) * WBUF, yes{}, ((t + 1) & 1) * WBUF, ra, rb);
            __syncthreads();
        };
        int t = 0;
        for (; t + 4 <= nfull; t += 3) {
            step(t, a1, b1);
            step(t + 1, a2, b2);
            step(t + 2, a0, b0);
        }
        const int left = nfull - t;   // 1, 2 or 3 tiles: t in buffer t & 1, t + 1 in (a1, b1), t + 2 in (a2, b2)
        if (left >= 2) {
            step_noload(t, a1, b1);
            if (left == 3) step_noload(t + 1, a2, b2);
        }
        multiply(((nfull - 1) & 1) * WBUF, no{}, 0, a0, b0);
        __syncthreads();
    }
    if (krem > 0) {
        load_partial(kbeg + nfull * TTK, a0, b0);
        store_part(0, a0, b0, -1);
        __syncthreads();
        multiply(0, no{}, 0, a0, b0);
        __syncthreads();
    }
    // the wave's 128 x 64 block leaves as two 64 x 64 halves through the shared epilogue
    store_acc_tile<TO>(&acc[0], tt_smem, nullptr, C, ws, M, N, ldc, accumulate, m0, n0, split, 0, 0, 0, nullptr, nullptr, 0, 0, wm * 128, wn * 64);
    store_acc_tile<TO>(&acc[2], tt_smem, nullptr, C, ws, M, N, ldc, accumulate, m0, n0, split, 0, 0, 0, nullptr, nullptr, 0, 0, wm * 128 + 64, wn * 64);
}

// the batched launch on the 256 x 128 tile: every problem's M a multiple of 256, N of 128 (the encoder layers' 768 x 512 / 512 x 768
// weight gradients: 12 tiles each).  The batched 128 x 128 kernel moves 2.45 GB through the L2 -> LDS path for the six 33 280-row
// gradients of a step (144 us at the ~17 TB/s that path sustains chip-wide, against 75 us of MFMA issue); this tile moves 1.84 GB.
// tb.first_tile / tiles_n count WIDE tiles here.
template <bool DB>
__global__ __launch_bounds__(512) void gemm_tn_batch_wide_kernel(TnBatch tb, float* __restrict__ ws, int K, int k_per_split, int nsplit) {
    const TnWork w = tn_batch_work(tb, ws, K, k_per_split, nsplit);
    if (w.j < 0) return;
    const TnBatch::P& q = tb.p[w.j];
    tn_wide_body<float, DB>(q.A, q.B, q.C, w.ws, q.M, q.N, w.K, q.lda, q.ldb, q.ldc, w.k_per_split, 0, q.tiles_n, w.tile, w.split);
}

#ifdef SPV_LAB   // measured slower inside the training step: kept for the lab build only
// ---------------------------------------------------------------------------------------------------------
// TN contraction, LDS-DMA variant: the weight gradients of the encoder layers, dW[768 x 512] = dh^T . x over 33 280 rows
// (backward of layers.py:86).  Same output tile (128 x 128) as gemm_tn_kernel, but
//   * operands staged by global_load_lds_dwordx4 (no staging VGPRs, no ds_write) into a THREE-stage LDS ring of 64 k rows,
//     two stages in flight, ONE barrier per K-tile behind a counted vmcnt;
//   * 8 waves: two K-groups (k-steps {0,1} / {2,3} of every K-tile) x 2 x 2 wave tiles of 64 x 64, so every SIMD holds two
//     waves and one wave's transposing LDS reads run under the other's MFMAs; the two partial tiles meet in LDS at the end
//     (each wave finishes 32 of its 64 rows);
//   * DMA writes are lane-linear (a wave instruction = 4 k rows of 256 B), so rows cannot be padded: the 16-byte chunk index is
//     XOR-ed with (row & 3) << 2 on the per-lane SOURCE address and again on the fragment read -- the 4 k rows x 2 column blocks
//     that one half of a ds_read_b64_tr_b16 touches then cover 8 disjoint 32-byte bank ranges = all 64 banks.
// Needs M, N multiples of 128 and K, the split-K slices multiples of 64 (the layer shapes); anything else: gemm_tn_kernel.
constexpr int TDK = 64;                      // k rows per stage
constexpr int TD_STAGE = 2 * TDK * 256;      // A tile + B tile, 256 B per k row each
constexpr int TD_NST = 3;
constexpr int TD_SMEM = TD_NST * TD_STAGE;   // 96 KiB (the epilogue's 64 KiB exchange + 8 x 8.5 KiB stages reuse it)

// (Tried and dropped: a ninth wave touching the slice's operand lines six K-tiles ahead, on the theory that every K-tile's first
// request pays an HBM round trip -- no change, 56.5 vs 55.4 us: the loop is bound by DMA ISSUE on the 64 B/clk L1 path, below.)
template <typename TO>
__global__ __launch_bounds__(512) void gemm_tn_dma_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, TO* __restrict__ C,
                                                          float* __restrict__ ws, int M, int N, int K, int lda, int ldb, int ldc,
                                                          int k_per_split, int accumulate, int tiles_n, int tiles_mn, int nsplit) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char td_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int lin = xcd_remap(blockIdx.x, tiles_mn * nsplit);  // whole K-slices per XCD: a slice's operand rows hit in that L2
    const int split = lin / tiles_mn, tile = lin % tiles_mn;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);

    // DMA: per stage and operand 16 wave instructions of 4 k rows; this wave issues instructions 2 wave, 2 wave + 1
    const bf16_t* asrc[2];
    const bf16_t* bsrc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int R = 4 * (2 * wave + t) + (lane >> 4);          // k row of the stage written by this lane
        const int c = (lane & 15) ^ ((R & 3) << 2);              // logical 16-byte chunk that belongs at this LDS position
        asrc[t] = A + (size_t)R * lda + m0 + c * 8;
        bsrc[t] = B + (size_t)R * ldb + n0 + c * 8;
    }
    auto stage = [&](int buf, int k0) __attribute__((always_inline)) {
        unsigned char* sa = td_smem + buf * TD_STAGE + (2 * wave) * 1024;
        unsigned char* sb = sa + TDK * 256;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[t] + (size_t)k0 * lda),
                                             (__attribute__((address_space(3))) void*)(sa + t * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[t] + (size_t)k0 * ldb),
                                             (__attribute__((address_space(3))) void*)(sb + t * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // transposing-read geometry (as gemm_tn_kernel): group g = lane >> 4 covers columns 16 (g & 1) + [0,16) of a 32-wide MFMA
    // block and k half g >> 1; lane 4 q + p of the group supplies row q, columns 4 p .. 4 p + 3
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
    // byte offset inside a k row of MFMA block x (32 columns), swizzled for row & 3 == q: chunk = 4 x + 2 (g & 1) + (pp >> 1)
    auto col_off = [&](int x) { return (((4 * x + 2 * (g & 1) + (pp >> 1)) ^ (q << 2)) << 4) + ((pp & 1) << 3); };
    int offa[2], offb[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        offa[f] = (8 * (g >> 1) + q) * 256 + col_off(2 * wm + f);
        offb[f] = TDK * 256 + (8 * (g >> 1) + q) * 256 + col_off(2 * wn + f);
    }
    using lds_tr = s16x4 __attribute__((address_space(3)))*;
    auto frag = [&](const unsigned char* p) __attribute__((always_inline)) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(p + 4 * 256));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };

    if (kbeg < kend) stage(0, kbeg);
    if (kbeg + TDK < kend) stage(1, kbeg + TDK);
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += TDK) {
        // this wave's share of stage `buf` has landed (the younger stage's 4 pieces may still fly) ...
        if (k0 + TDK < kend) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // ... everyone's has, and everyone is done reading the stage refilled next
        // The four DMA pieces of K-tile t + 2 go out ONE behind every MFMA pair of K-tile t: issued together right after the barrier
        // the eight waves' 32 pieces (32 KiB) queue on the CU's 64 B/clk L1 path and every wave stands in issue for hundreds of
        // cycles before its first MFMA (measured: 1 800 cycles per K-tile against 512 of MFMA issue)
        const bool dma = k0 + 2 * TDK < kend;
        const int nbuf = buf == 0 ? 2 : buf - 1;
        unsigned char* da = td_smem + nbuf * TD_STAGE + (2 * wave) * 1024;
        unsigned char* db = da + TDK * 256;
        const size_t ka = (size_t)(k0 + 2 * TDK) * lda, kb = (size_t)(k0 + 2 * TDK) * ldb;
        const unsigned char* st = td_smem + buf * TD_STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int ks = 2 * kg + s2;
            bf16x8 a[2], b[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                a[f] = frag(st + ks * 16 * 256 + offa[f]);
                b[f] = frag(st + ks * 16 * 256 + offb[f]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_sched_barrier(0);
                if (dma) {
                    if (i == 0)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[s2] + ka),
                                                         (__attribute__((address_space(3))) void*)(da + s2 * 1024), 16, 0, 0);
                    else
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[s2] + kb),
                                                         (__attribute__((address_space(3))) void*)(db + s2 * 1024), 16, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        buf = buf == 2 ? 0 : buf + 1;
    }

    // ---- the two K-groups meet: wave (kg, wm, wn) finishes rows [32 kg, 32 kg + 32) of its 64 x 64 block, i.e. acc[kg][*] plus
    // the partner's acc[kg][*]; each wave hands over the half it does not finish (8 KiB per wave, lane-linear)
    __syncthreads();  // every wave is out of the K loop: the ring is free
    {
        float* xch = reinterpret_cast<float*>(td_smem) + (size_t)wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[(j * 16 + r) * 64 + lane] = kg == 0 ? acc[1][j][r] : acc[0][j][r];
    }
    __syncthreads();
    f32x16 fin[2];
    {
        const float* xch = reinterpret_cast<const float*>(td_smem) + (size_t)(wave ^ 4) * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) fin[j][r] = (kg == 0 ? acc[0][j][r] : acc[1][j][r]) + xch[(j * 16 + r) * 64 + lane];
    }
    __syncthreads();  // exchange area read: reuse it as the wave-private store stages
    {
        // 32 x 64 block of this wave -> fp32 LDS stage [32][68] -> every lane leaves with 8 consecutive columns of a row
        constexpr int SLD = 68;
        float* stg = reinterpret_cast<float*>(td_smem + wave * 9216);
        const int fh = lane >> 5;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * fh) * SLD + j * 32 + (lane & 31)] = fin[j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private stage: the wave's own writes are done before it reads
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int qd = lane + 64 * t;
            const int lr = qd >> 3, c8 = qd & 7;
            const int row = m0 + wm * 64 + kg * 32 + lr;
            const int col0 = n0 + wn * 64 + c8 * 8;
            const float4 lo = *reinterpret_cast<const float4*>(stg + lr * SLD + c8 * 8);
            const float4 hi = *reinterpret_cast<const float4*>(stg + lr * SLD + c8 * 8 + 4);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            if (ws != nullptr) {
                float4* wp = reinterpret_cast<float4*>(ws + ((size_t)split * M + row) * N + col0);
                wp[0] = lo;
                wp[1] = hi;
                continue;
            }
            TO* cp = C + (size_t)row * ldc + col0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (accumulate) v[u] += load_out<TO>(cp + u);
                store_out<TO>(cp + u, v[u]);
            }
        }
    }
}
#endif  // SPV_LAB

template <typename TO>
__device__ __forceinline__ void splitk_reduce_body(const float* __restrict__ ws, const float* __restrict__ bias, TO* __restrict__ C, int M, int N,
                                                   int ldc, int splits, int accumulate, size_t bid, size_t nblocks, int tid, int nthreads) {
    const size_t total = (size_t)M * N;
    if ((N & 3) == 0) {
        // four consecutive columns per thread, the slabs read four at a time with independent 16-byte loads (the scalar
        // loop was a chain of dependent 4-byte loads: 9.3 us for the weight gradient's 12 x 1.5 MB, twice its bytes' worth)
        const size_t quads = total >> 2;
        for (size_t qd = bid * nthreads + tid; qd < quads; qd += nblocks * nthreads) {
            const size_t e = qd << 2;
            const int row = (int)(e / N), col = (int)(e % N);
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            int sidx = 0;
            for (; sidx + 4 <= splits; sidx += 4) {
                const float4 a0 = *reinterpret_cast<const float4*>(ws + (size_t)(sidx + 0) * total + e);
                const float4 a1 = *reinterpret_cast<const float4*>(ws + (size_t)(sidx + 1) * total + e);
                const float4 a2 = *reinterpret_cast<const float4*>(ws + (size_t)(sidx + 2) * total + e);
                const float4 a3 = *reinterpret_cast<const float4*>(ws + (size_t)(sidx + 3) * total + e);
                // same association as the scalar loop: ((((v + a0) + a1) + a2) + a3)
                v.x = (((v.x + a0.x) + a1.x) + a2.x) + a3.x;
                v.y = (((v.y + a0.y) + a1.y) + a2.y) + a3.y;
                v.z = (((v.z + a0.z) + a1.z) + a2.z) + a3.z;
                v.w = (((v.w + a0.w) + a1.w) + a2.w) + a3.w;
            }
            for (; sidx < splits; ++sidx) {
                const float4 a0 = *reinterpret_cast<const float4*>(ws + (size_t)sidx * total + e);
                v.x += a0.x; v.y += a0.y; v.z += a0.z; v.w += a0.w;
            }
            float o[4] = {v.x, v.y, v.z, v.w};
            TO* cp = C + (size_t)row * ldc + col;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (bias) o[u] += bias[col + u];
                if (accumulate) o[u] += load_out<TO>(cp + u);
                store_out<TO>(cp + u, o[u]);
            }
        }
        return;
    }
    for (size_t e = bid * nthreads + tid; e < total; e += nblocks * nthreads) {
        const int row = (int)(e / N), col = (int)(e % N);
        float v = 0.0f;
        for (int s = 0; s < splits; ++s) v += ws[(size_t)s * total + e];
        if (bias) v += bias[col];
        TO* cp = C + (size_t)row * ldc + col;
        if (accumulate) v += load_out<TO>(cp);
        store_out<TO>(cp, v);
    }
}

template <typename TO>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                            TO* __restrict__ C, int M, int N, int ldc, int splits,
                                                            int accumulate) {
    splitk_reduce_body<TO>(ws, bias, C, M, N, ldc, splits, accumulate, blockIdx.x, gridDim.x, threadIdx.x, 256);
}

// Several folds in ONE launch (spv_fold_multi), and up to FJ_MAX folds riding in a split-K reduce (spv_gemm_tn_fold).  A 5-us launch
// between two large kernels costs the step ~20 us (eight per-layer fold launches removed: 157 us), so folds travel with launches that
// happen anyway.  (Keeping the split-K SUMS back until the end of the backward as well was measured and dropped: 2.09 -> 2.23 ms --
// eight 33 MB workspaces outlive their stay in the 256 MB MALL and the late sum reads them from HBM.)
constexpr int FJ_MAX = 16;
struct FoldJobs {
    int njobs;
    int first_block[FJ_MAX + 1];
    FoldJob job[FJ_MAX];
};
__device__ __forceinline__ void fold_jobs_block(const FoldJobs& fj, int bid, int tid) {
    int j = 0;
    while (j + 1 < fj.njobs && bid >= fj.first_block[j + 1]) ++j;   // workgroup-uniform
    fold_partials_block(fj.job[j], bid - fj.first_block[j], tid);
}
// the stand-alone fold launch: 256-thread workgroups with few registers, which the dispatcher places on CUs that a workgroup of the batched
// weight gradients occupies (8 waves x 184 VGPRs and 112 KB leave room for 4 more waves of up to 144) -- hip_ops.start_held_wgrads issues
// it on the main stream while that kernel runs on the side stream, instead of as extra workgroups of its reduce behind it
constexpr int FOLD_NARROW = 4;
__global__ __launch_bounds__(FOLD_COLS * FOLD_ROWS / FOLD_NARROW) void fold_multi_kernel(FoldJobs fj) {
    int j = 0;
    while (j + 1 < fj.njobs && (int)blockIdx.x >= fj.first_block[j + 1]) ++j;   // workgroup-uniform
    fold_partials_block<FOLD_NARROW>(fj.job[j], blockIdx.x - fj.first_block[j], threadIdx.x);
}

// the split-K reduce of a weight gradient with folds (the same layer's dgamma / dbeta / dbias partials, and any fold held back by an
// earlier node of the backward) as extra workgroups
template <typename TO>
__global__ __launch_bounds__(FOLD_COLS * FOLD_ROWS) void splitk_reduce_fold_kernel(const float* __restrict__ ws, TO* __restrict__ C, int M, int N,
                                                                                   int ldc, int splits, int accumulate, int reduce_blocks,
                                                                                   FoldJobs fj) {
    // the folds first: their workgroups are the long ones (a column of up to 1024 partial rows each), the sums' are many and short
    const int fold_blocks = fj.first_block[fj.njobs];
    if ((int)blockIdx.x < fold_blocks)
        fold_jobs_block(fj, (int)blockIdx.x, threadIdx.x);
    else if (threadIdx.x < 256)
        // a quarter of the block's threads each: the sum has 98 k four-column quads at the layer shapes -- as 1024-thread blocks that is
        // 96 workgroups for 256 CUs; as 256 live threads per block, 384
        splitk_reduce_body<TO>(ws, nullptr, C, M, N, ldc, splits, accumulate, blockIdx.x - fold_blocks, reduce_blocks, threadIdx.x, 256);
}

// the split-K sums of a batch (gemm_tn_batch_kernel) + folds, one launch: folds first, then reduce_blocks workgroups per problem
__global__ __launch_bounds__(FOLD_COLS * FOLD_ROWS) void splitk_reduce_batch_kernel(TnBatch tb, const float* __restrict__ ws, int splits,
                                                                                    int reduce_blocks, FoldJobs fj) {
    const int fold_blocks = fj.njobs > 0 ? fj.first_block[fj.njobs] : 0;
    if ((int)blockIdx.x < fold_blocks) {
        fold_jobs_block(fj, (int)blockIdx.x, threadIdx.x);
    } else {   // all 1024 threads sum (eight problems are 786 k four-column quads: 768 workgroups)
        const int b = (int)blockIdx.x - fold_blocks;
        const int j = b / reduce_blocks;   // (the launch has reduce_blocks workgroups per LONG problem: short ones stored their result)
        const TnBatch::P& q = tb.p[j];
        splitk_reduce_body<float>(ws + tb.ws_off[j], nullptr, q.C, q.M, q.N, q.ldc, splits, 0, b % reduce_blocks, reduce_blocks, threadIdx.x,
                                  FOLD_COLS * FOLD_ROWS);
    }
}

inline int kend_len(int K, int k_per_split) { return K < k_per_split ? K : k_per_split; }

// dropout of the grouped-rows epilogue (spv_gemm_nt_grouped_rows_drop sets them around its call; 0 everywhere else)
static thread_local float t_drop_p = 0.0f;
static thread_local uint64_t t_drop_seed = 0;

template <typename T, typename TO>
int launch_gemm(const void* A, const void* B, const float* bias, void* C, int M, int N, int K, int lda, int ldb,
                int ldc, int accumulate, int splits, void* workspace, hipStream_t st, int rg = 0, int gs = 0, int roff = 0,
                const float* bias2d = nullptr, const void* bc = nullptr, int bc_pw = 0, int bc_bf = 0) {
    constexpr int BK = KT<T>::BK;
    const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
    int k_per_split = K;
    if (splits > 1) {
        k_per_split = cdiv(cdiv(K, splits), BK) * BK;
        splits = cdiv(K, k_per_split);
    }
    float* ws = splits > 1 ? static_cast<float*>(workspace) : nullptr;
    dim3 grid(tiles_m * tiles_n * splits);
    if constexpr (sizeof(T) == 2) {
        static const bool no_rows = SPV_LAB_SET("SPV_GEMM_NO_ROWS");   // A/B switch
        // few rows: one 32 x 32 tile per workgroup straight from the L2-resident operands (no split-K workspace, no reduce launch)
        // (K <= 1536: a wave walks its K share in batches of 8 k-steps, one L2 round trip each -- at K = 8192 that chain is 40 us against
        // 15 + 5 for split-K + reduce; <= 1024 tiles: beyond that the 32 x 32 tiles re-read the operands 4x as often as 128 x 128 ones)
        if (!no_rows && splits == 1 && M <= 2048 && K <= 1536 && rg == 0 && bias2d == nullptr && bc == nullptr && t_drop_p == 0.0f && K % 16 == 0 &&
            lda % 8 == 0 && ldb % 8 == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 &&
            cdiv(M, 32) * cdiv(N, 32) >= 64 && cdiv(M, 32) * cdiv(N, 32) <= 1024) {
            const int tn32 = cdiv(N, 32);
            hipLaunchKernelGGL((gemm_nt_rows_kernel<TO>), dim3(cdiv(M, 32) * tn32), dim3(256), 0, st, static_cast<const bf16_t*>(A),
                               static_cast<const bf16_t*>(B), bias, static_cast<TO*>(C), M, N, K, lda, ldb, ldc, accumulate, tn32);
            SPV_LAUNCH_CHECK("spv_gemm_nt(rows)");
            SPV_COUNT_PATH(SPV_PATH_GEMM_ROWS);
            return 0;
        }
    }
    if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) {
        int mb, nstrips, groups, base, rem;
        const bool bc_ok = bc == nullptr || (bc_bf && !accumulate && bc_pw >= 8 && N % bc_pw == 0);   // (8 columns span <= 2 windows)
        const bool span32 = (size_t)M * lda * 2 < (1ull << 32) && (size_t)N * ldb * 2 < (1ull << 32);  // 32-bit DMA source offsets
        if (splits == 1 && rg == 0 && bias2d == nullptr && t_drop_p == 0.0f && bc_ok && span32 && ldc % 8 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
            (bias == nullptr || (reinterpret_cast<uintptr_t>(bias) & 3) == 0) && strip_plan(M, N, K, mb, nstrips, groups, base, rem)) {
            const int nwg = groups * nstrips;
#define SPV_STRIP(MBV, EPIV)                                                                                                \
    hipLaunchKernelGGL((gemm_nt_strip_kernel<MBV, EPIV>), dim3(nwg), dim3(512), 0, st, static_cast<const bf16_t*>(A),       \
                       static_cast<const bf16_t*>(B), bias, static_cast<bf16_t*>(C), M, K, lda, ldb, ldc, nstrips, base, rem, nwg, \
                       static_cast<const bf16_t*>(bc), bc_pw, bc ? N / bc_pw : 0)
            static const int acc_mb = SPV_LAB_INT("SPV_STRIP_ACC_MB", 0);  // tuning aid
            // diagnosis only (wrong results): every A row reads row 0, i.e. A always hits in L2 -- separates "HBM latency x one K-tile of
            // prefetch" from "L2 -> LDS rate" as the bound of the K loop
#ifdef SPV_LAB
            static const bool lda0 = SPV_LAB_SET("SPV_STRIP_LDA0");
            if (lda0) lda = 0;
#endif
            if (accumulate && acc_mb >= 2 && acc_mb <= 4) mb = acc_mb;
            SPV_COUNT_PATH(bc != nullptr ? SPV_PATH_GEMM_STRIP_POOL : accumulate ? SPV_PATH_GEMM_STRIP_ACC : SPV_PATH_GEMM_STRIP);
            if (bc != nullptr && bc_pw % 8 != 0) {   // two-window epilogue: its own instantiation (in the one-window kernel it cost 6 %)
                if (mb == 4) SPV_STRIP(4, 3);
                else if (mb == 3) SPV_STRIP(3, 3);
                else SPV_STRIP(2, 3);
            } else if (bc != nullptr) {
                if (mb == 4) SPV_STRIP(4, 2);
                else if (mb == 3) SPV_STRIP(3, 2);
                else SPV_STRIP(2, 2);
            } else if (accumulate) {
                if (mb == 4) SPV_STRIP(4, 1);
                else if (mb == 3) SPV_STRIP(3, 1);
                else SPV_STRIP(2, 1);
            } else {
                if (mb == 4) SPV_STRIP(4, 0);
                else if (mb == 3) SPV_STRIP(3, 0);
                else SPV_STRIP(2, 0);
            }
#undef SPV_STRIP
            SPV_LAUNCH_CHECK("spv_gemm_nt(strip)");
            return 0;
        }
    }
    if constexpr (sizeof(T) == 2) {
        // direct-to-LDS double-buffered kernel for long reductions (measured: 896 vs 795 TFLOP/s at 4096^3, 755 vs 700 on
        // the 512 x 8192 x 33280 weight gradient); the skinny K <= 1024 layer GEMMs are faster on the register-staged
        // kernel (42 vs 50 us at 33280 x 768 x 512: three workgroups per CU instead of two)
        static const int force_kb = SPV_LAB_INT("SPV_GEMM_KB", 0);  // tuning aid: 64 / 128 / -1 (off)
        // K <= 1024 (the layer GEMMs) stays on the register-staged kernel: the three-stage 64-byte-row ring measures the
        // same step time (2.978 vs 2.972 ms over three alternating runs) and 4 us more per isolated launch
        if (force_kb >= 0 && K % GBK == 0 && k_per_split % GBK == 0 && (force_kb || kend_len(K, k_per_split) > 1024)) {
            const int kb = force_kb ? force_kb : 128;
            (void)kb;
#ifdef SPV_LAB
            if (kb == 64)
                hipLaunchKernelGGL((gemm_nt_glds_kernel<TO, 64, 3>), grid, dim3(256), 0, st, static_cast<const bf16_t*>(A),
                                   static_cast<const bf16_t*>(B), bias, static_cast<TO*>(C), ws, M, N, K, lda, ldb, ldc, k_per_split,
                                   accumulate, tiles_n, tiles_m * tiles_n, splits, rg, gs, roff, bias2d, bc, bc_pw, bc_bf, t_drop_p, t_drop_seed);
            else
#endif
                hipLaunchKernelGGL((gemm_nt_glds_kernel<TO, 128, 2>), grid, dim3(256), 0, st, static_cast<const bf16_t*>(A),
                                   static_cast<const bf16_t*>(B), bias, static_cast<TO*>(C), ws, M, N, K, lda, ldb, ldc, k_per_split,
                                   accumulate, tiles_n, tiles_m * tiles_n, splits, rg, gs, roff, bias2d, bc, bc_pw, bc_bf, t_drop_p, t_drop_seed);
            SPV_LAUNCH_CHECK("spv_gemm_nt(glds)");
            goto reduce;
        }
    }
    hipLaunchKernelGGL((gemm_nt_kernel<T, TO>), grid, dim3(256), 0, st, static_cast<const T*>(A),
                       static_cast<const T*>(B), bias, static_cast<TO*>(C), ws, M, N, K, lda, ldb, ldc, k_per_split,
                       accumulate, tiles_n, tiles_m * tiles_n, splits, rg, gs, roff, bias2d, bc, bc_pw, bc_bf, t_drop_p, t_drop_seed);
    SPV_LAUNCH_CHECK("spv_gemm_nt");
reduce:
    if (splits > 1) {
        int blocks = (int)std::min<int64_t>(((int64_t)M * N / ((N & 3) == 0 ? 4 : 1) + 255) / 256, 2048);
        hipLaunchKernelGGL((splitk_reduce_kernel<TO>), dim3(blocks), dim3(256), 0, st, ws, bias, static_cast<TO*>(C), M,
                           N, ldc, splits, accumulate);
        SPV_LAUNCH_CHECK("spv_gemm_nt(split-k reduce)");
    }
    return 0;
}

}  // namespace

static int gemm_entry(const void* A, const void* B, const float* bias, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                      int in_dtype, int out_dtype, int accumulate, int splits, void* workspace, void* stream, int rg, int gs,
                      int roff, const float* bias2d, const void* bc = nullptr, int bc_pw = 0, int bc_bf = 0);

extern "C" int spv_set_reserved_cus(int n) {
    SPV_CHECK(n >= 0 && n <= 128, "spv_set_reserved_cus: %d outside 0..128", n);
    g_reserved_cus = n;
    return 0;
}

extern "C" int spv_gemm_nt_pool_bwd(const void* A, const void* B, void* C, const void* dout, int pool_window, int M, int N, int K,
                                    int lda, int ldb, int ldc, int in_dtype, int out_dtype, int dout_dtype, void* stream) {
    SPV_CHECK(dout != nullptr && pool_window > 0 && N % pool_window == 0, "spv_gemm_nt_pool_bwd: bad pooling window %d for N=%d",
              pool_window, N);
    SPV_CHECK(dout_dtype == SPV_F32 || dout_dtype == SPV_BF16, "spv_gemm_nt_pool_bwd: bad dout dtype");
    return gemm_entry(A, B, nullptr, C, M, N, K, lda, ldb, ldc, in_dtype, out_dtype, 0, 1, nullptr, stream, 0, 0, 0, nullptr, dout,
                      pool_window, dout_dtype == SPV_BF16);
}

extern "C" int spv_gemm_nt(const void* A, const void* B, const float* bias, void* C, int M, int N, int K, int lda, int ldb,
                           int ldc, int in_dtype, int out_dtype, int accumulate, int splits, void* workspace, void* stream) {
    return gemm_entry(A, B, bias, C, M, N, K, lda, ldb, ldc, in_dtype, out_dtype, accumulate, splits, workspace, stream, 0, 0, 0,
                      nullptr);
}

extern "C" int spv_gemm_nt_grouped_rows(const void* A, const void* B, const float* bias, const float* bias2d, void* C, int M,
                                        int N, int K, int lda, int ldb, int ldc, int in_dtype, int out_dtype,
                                        int rows_per_group, int group_stride, int row_offset, void* stream) {
    SPV_CHECK(rows_per_group > 0 && group_stride >= rows_per_group + row_offset && row_offset >= 0,
              "spv_gemm_nt_grouped_rows: bad grouping %d/%d/%d", rows_per_group, group_stride, row_offset);
    return gemm_entry(A, B, bias, C, M, N, K, lda, ldb, ldc, in_dtype, out_dtype, 0, 1, nullptr, stream, rows_per_group,
                      group_stride, row_offset, bias2d);
}

extern "C" int spv_gemm_nt_grouped_rows_drop(const void* A, const void* B, const float* bias, const float* bias2d, void* C, int M,
                                             int N, int K, int lda, int ldb, int ldc, int in_dtype, int out_dtype,
                                             int rows_per_group, int group_stride, int row_offset, float p_drop, uint64_t seed,
                                             void* stream) {
    SPV_CHECK(p_drop >= 0.0f && p_drop < 1.0f, "spv_gemm_nt_grouped_rows_drop: p_drop=%f", p_drop);
    SPV_CHECK(p_drop == 0.0f || ldc == N, "spv_gemm_nt_grouped_rows_drop: the mask is taken on the flat index of a contiguous output (ldc == N)");
    t_drop_p = p_drop;
    t_drop_seed = seed;
    const int rc = spv_gemm_nt_grouped_rows(A, B, bias, bias2d, C, M, N, K, lda, ldb, ldc, in_dtype, out_dtype, rows_per_group, group_stride,
                                            row_offset, stream);
    t_drop_p = 0.0f;
    t_drop_seed = 0;
    return rc;
}

int spv_seed_ptr_set_gemm(const unsigned long long* p) { return spv_seed_symbol_set(p); }

static int gemm_entry(const void* A, const void* B, const float* bias, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                      int in_dtype, int out_dtype, int accumulate, int splits, void* workspace, void* stream, int rg, int gs,
                      int roff, const float* bias2d, const void* bc, int bc_pw, int bc_bf) {
    SPV_CHECK(M > 0 && N > 0 && K > 0, "spv_gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
    SPV_CHECK(in_dtype == SPV_F32 || in_dtype == SPV_BF16, "spv_gemm_nt: bad in_dtype %d", in_dtype);
    SPV_CHECK(out_dtype == SPV_F32 || out_dtype == SPV_BF16, "spv_gemm_nt: bad out_dtype %d", out_dtype);
    const int ch = in_dtype == SPV_BF16 ? 8 : 4;
    SPV_CHECK(K % ch == 0 && lda % ch == 0 && ldb % ch == 0,
              "spv_gemm_nt: K=%d lda=%d ldb=%d must be multiples of %d elements (16 bytes)", K, lda, ldb, ch);
    SPV_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "spv_gemm_nt: A/B must be 16-byte aligned");
    SPV_CHECK(lda >= K && ldb >= K && ldc >= N, "spv_gemm_nt: leading dimension too small");
    SPV_CHECK(splits >= 1, "spv_gemm_nt: splits=%d", splits);
    SPV_CHECK(splits == 1 || workspace != nullptr, "spv_gemm_nt: split-K needs a workspace");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (in_dtype == SPV_BF16) {
        if (out_dtype == SPV_BF16)
            return launch_gemm<bf16_t, bf16_t>(A, B, bias, C, M, N, K, lda, ldb, ldc, accumulate, splits, workspace, st, rg, gs, roff, bias2d, bc, bc_pw, bc_bf);
        return launch_gemm<bf16_t, float>(A, B, bias, C, M, N, K, lda, ldb, ldc, accumulate, splits, workspace, st, rg, gs, roff, bias2d, bc, bc_pw, bc_bf);
    }
    if (out_dtype == SPV_BF16)
        return launch_gemm<float, bf16_t>(A, B, bias, C, M, N, K, lda, ldb, ldc, accumulate, splits, workspace, st, rg, gs, roff, bias2d, bc, bc_pw, bc_bf);
    return launch_gemm<float, float>(A, B, bias, C, M, N, K, lda, ldb, ldc, accumulate, splits, workspace, st, rg, gs, roff, bias2d, bc, bc_pw, bc_bf);
}

static int fill_fold_jobs(const spv_fold_job* folds, int nfolds, FoldJobs& fj, int& blocks) {
    fj.njobs = nfolds;
    blocks = 0;
    for (int i = 0; i < nfolds; ++i) {
        const spv_fold_job& f = folds[i];
        if (f.partials == nullptr || f.parts <= 0 || f.nsum < 1 || f.nsum > 5 || f.n <= 0 || f.out[0] == nullptr) return -1;
        fj.first_block[i] = blocks;
        fj.job[i].partials = f.partials;
        for (int u = 0; u < 5; ++u) fj.job[i].o[u] = u < f.nsum ? f.out[u] : nullptr;
        fj.job[i].parts = f.parts;
        fj.job[i].np = f.nsum;
        fj.job[i].n = f.n;
        blocks += cdiv((int64_t)f.nsum * f.n, FOLD_COLS);
    }
    fj.first_block[nfolds] = blocks;
    return 0;
}

static int gemm_tn_impl(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int out_dtype,
                        int accumulate, int splits, void* workspace, void* stream, const spv_fold_job* folds, int nfolds) {
    SPV_COUNT_PATH(SPV_PATH_GEMM_TN);
    SPV_CHECK(M > 0 && N > 0 && K > 0, "spv_gemm_tn: empty problem M=%d N=%d K=%d", M, N, K);
    SPV_CHECK(out_dtype == SPV_F32 || out_dtype == SPV_BF16, "spv_gemm_tn: bad out_dtype %d", out_dtype);
    SPV_CHECK(M % 8 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0,
              "spv_gemm_tn: M=%d N=%d lda=%d ldb=%d must be multiples of 8 (bf16, 16-byte chunks)", M, N, lda, ldb);
    SPV_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "spv_gemm_tn: A/B must be 16-byte aligned");
    SPV_CHECK(lda >= M && ldb >= N && ldc >= N, "spv_gemm_tn: leading dimension too small");
    SPV_CHECK(splits >= 1 && (splits == 1 || workspace != nullptr), "spv_gemm_tn: split-K needs a workspace");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
    int k_per_split = K;
    if (splits > 1) {
        k_per_split = cdiv(cdiv(K, splits), TBK) * TBK;
        splits = cdiv(K, k_per_split);
    }
    float* ws = splits > 1 ? static_cast<float*>(workspace) : nullptr;
    // opt-in (SPV_TN_DMA=1): isolated it is 3-6 % faster than the register-staged kernel (52.5 vs 54.5 us incl. the reduce), inside
    // the training step it measured SLOWER (68 vs 47 us per launch), so the register-staged kernel stays the default
    static const int use_dma = SPV_LAB_INT("SPV_TN_DMA", 0);
    (void)use_dma;
    static const int use_wide = SPV_LAB_INT("SPV_TN_WIDE", 1);
    static const int wide_min = SPV_LAB_INT("SPV_TN_WIDE_MIN", 4);   // tuning aid
#ifdef SPV_LAB
    if (use_dma && M % BM == 0 && N % BN == 0 && K % TDK == 0 && k_per_split % TDK == 0 && (ws != nullptr || ldc % 4 == 0)) {
        const int nwg = tiles_m * tiles_n * splits;
        if (out_dtype == SPV_BF16) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_dma_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, TD_SMEM);
            hipLaunchKernelGGL((gemm_tn_dma_kernel<bf16_t>), dim3(nwg), dim3(512), TD_SMEM, st, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, ws,
                               M, N, K, lda, ldb, ldc, k_per_split, accumulate, tiles_n, tiles_m * tiles_n, splits);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_dma_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, TD_SMEM);
            hipLaunchKernelGGL((gemm_tn_dma_kernel<float>), dim3(nwg), dim3(512), TD_SMEM, st, (const bf16_t*)A, (const bf16_t*)B, (float*)C, ws,
                               M, N, K, lda, ldb, ldc, k_per_split, accumulate, tiles_n, tiles_m * tiles_n, splits);
        }
        SPV_LAUNCH_CHECK("spv_gemm_tn(dma)");
        SPV_COUNT_PATH(SPV_PATH_GEMM_TN_DMA);
    } else
#endif
    static const int use_tall = SPV_LAB_INT("SPV_TN_TALL", 1);
    if (use_tall && M == TTM && N % BN == 0 && N >= 8 * BN && tiles_n * splits >= 192 && k_per_split >= 16 * TTK) {
        // all of M in one workgroup: the B panel (the gathered matrix of the MHPermutMix gradient) is read once
        const int nwg = tiles_n * splits;
        if (out_dtype == SPV_BF16) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_tall_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, TT_SMEM);
            hipLaunchKernelGGL((gemm_tn_tall_kernel<bf16_t>), dim3(nwg), dim3(512), TT_SMEM, st, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, ws, M, N, K, lda, ldb,
                               ldc, k_per_split, accumulate, tiles_n, tiles_n, splits);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_tall_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, TT_SMEM);
            hipLaunchKernelGGL((gemm_tn_tall_kernel<float>), dim3(nwg), dim3(512), TT_SMEM, st, (const bf16_t*)A, (const bf16_t*)B, (float*)C, ws, M, N, K, lda, ldb,
                               ldc, k_per_split, accumulate, tiles_n, tiles_n, splits);
        }
        SPV_LAUNCH_CHECK("spv_gemm_tn(tall)");
    } else
    if (use_wide && splits >= wide_min && M % TWM == 0 && N % BN == 0 && (M / TWM) * tiles_n * splits >= 128) {
        // the 256 x 128 tile (8 waves, one workgroup per CU): the split-K layer weight gradients (SPV_TN_WIDE=0 for the 128 x 128
        // kernel).  Measured in graph mode, alternating: 2.372 vs 2.385 ms/step -- 1.6 us per launch; without split-K (the MHPermutMix
        // weight gradient, 512 x 8192 x 33 280) it is SLOWER (651 vs 510 us), hence splits >= 4
        const int wt = (M / TWM) * tiles_n;
        static const int wide_db = SPV_LAB_INT("SPV_TN_WIDE_DB", 1);   // two LDS buffers, one barrier per K-tile
#define SPV_TNW(TOV, DBV, SM)                                                                                                             \
    do {                                                                                                                                  \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_wide_kernel<TOV, DBV>), hipFuncAttributeMaxDynamicSharedMemorySize, SM); \
        hipLaunchKernelGGL((gemm_tn_wide_kernel<TOV, DBV>), dim3(wt * splits), dim3(512), SM, st, (const bf16_t*)A, (const bf16_t*)B, (TOV*)C,   \
                           ws, M, N, K, lda, ldb, ldc, k_per_split, accumulate, tiles_n, wt, splits);                                     \
    } while (0)
        if (out_dtype == SPV_BF16) {
            if (wide_db) SPV_TNW(bf16_t, true, TW_SMEM_DB); else SPV_TNW(bf16_t, false, TW_SMEM);
        } else {
            if (wide_db) SPV_TNW(float, true, TW_SMEM_DB); else SPV_TNW(float, false, TW_SMEM);
        }
#undef SPV_TNW
        SPV_LAUNCH_CHECK("spv_gemm_tn(wide)");
        SPV_COUNT_PATH(SPV_PATH_GEMM_TN_WIDE);
    } else {
    dim3 grid(tiles_m * tiles_n * splits);
    // One K-tile in flight (144 VGPRs, 40 KB of LDS) for the sliver-shaped gradient of the patch embedding (512 x 48 x 33 280): that
    // launch runs on the main stream BESIDE the batched layer gradients, whose workgroups (8 waves x 184 VGPRs, 112 KB) leave exactly
    // 144 VGPRs per SIMD and 45 KB per CU -- with three tiles in flight (216 VGPRs) its workgroups waited for those CUs to drain.
    static const int depth_env = SPV_LAB_INT("SPV_TN_DEPTH", 0);  // tuning aid: 1 / 3 force a depth
    const int depth = depth_env ? depth_env : (N <= 64 ? 1 : 3);
#define SPV_TN(TOV, DV)                                                                                                     \
    hipLaunchKernelGGL((gemm_tn_kernel<TOV, DV>), grid, dim3(256), 0, st, (const bf16_t*)A, (const bf16_t*)B, (TOV*)C, ws, M, N, K, \
                       lda, ldb, ldc, k_per_split, accumulate, tiles_n, tiles_m * tiles_n, splits)
    if (out_dtype == SPV_BF16) {
        if (depth == 1) SPV_TN(bf16_t, 1); else SPV_TN(bf16_t, 3);
    } else {
        if (depth == 1) SPV_TN(float, 1); else SPV_TN(float, 3);
    }
#undef SPV_TN
    SPV_LAUNCH_CHECK("spv_gemm_tn");
    }
    FoldJobs fj{};
    int fold_blocks = 0;
    SPV_CHECK(nfolds >= 0 && nfolds <= FJ_MAX && (nfolds == 0 || folds != nullptr), "spv_gemm_tn_fold: 0..%d fold jobs", FJ_MAX);
    SPV_CHECK(fill_fold_jobs(folds, nfolds, fj, fold_blocks) == 0, "spv_gemm_tn_fold: bad fold job");
    if (splits > 1) {
        if (fold_blocks > 0) {
            constexpr int RT1 = FOLD_COLS * FOLD_ROWS;
            const int blocks = (int)std::min<int64_t>(((int64_t)M * N / ((N & 3) == 0 ? 4 : 1) + 255) / 256, 2048);
            if (out_dtype == SPV_BF16)
                hipLaunchKernelGGL((splitk_reduce_fold_kernel<bf16_t>), dim3(blocks + fold_blocks), dim3(RT1), 0, st, ws, (bf16_t*)C, M, N, ldc, splits,
                                   accumulate, blocks, fj);
            else
                hipLaunchKernelGGL((splitk_reduce_fold_kernel<float>), dim3(blocks + fold_blocks), dim3(RT1), 0, st, ws, (float*)C, M, N, ldc, splits,
                                   accumulate, blocks, fj);
            SPV_LAUNCH_CHECK("spv_gemm_tn_fold(split-k reduce + fold)");
            return 0;
        }
        int blocks = (int)std::min<int64_t>(((int64_t)M * N / ((N & 3) == 0 ? 4 : 1) + 255) / 256, 2048);
        if (out_dtype == SPV_BF16)
            hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, ws, (const float*)nullptr, (bf16_t*)C, M, N,
                               ldc, splits, accumulate);
        else
            hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3(blocks), dim3(256), 0, st, ws, (const float*)nullptr, (float*)C, M, N, ldc,
                               splits, accumulate);
        SPV_LAUNCH_CHECK("spv_gemm_tn(split-k reduce)");
    }
    if (fold_blocks > 0 && splits <= 1) {   // no reduce to ride on: the fold runs by itself
        hipLaunchKernelGGL(fold_multi_kernel, dim3(fold_blocks), dim3(FOLD_COLS * FOLD_ROWS / FOLD_NARROW), 0, st, fj);
        SPV_LAUNCH_CHECK("spv_gemm_tn_fold(fold)");
    }
    return 0;
}

extern "C" int spv_gemm_tn(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int out_dtype,
                           int accumulate, int splits, void* workspace, void* stream) {
    return gemm_tn_impl(A, B, C, M, N, K, lda, ldb, ldc, out_dtype, accumulate, splits, workspace, stream, nullptr, 0);
}

extern "C" int spv_gemm_tn_fold(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int out_dtype,
                                int accumulate, int splits, void* workspace, const spv_fold_job* folds, int nfolds, void* stream) {
    return gemm_tn_impl(A, B, C, M, N, K, lda, ldb, ldc, out_dtype, accumulate, splits, workspace, stream, folds, nfolds);
}

extern "C" int spv_fold_multi(const spv_fold_job* folds, int nfolds, void* stream) {
    SPV_CHECK(folds != nullptr && nfolds > 0, "spv_fold_multi: no jobs");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int j0 = 0; j0 < nfolds; j0 += FJ_MAX) {
        FoldJobs fj{};
        int blocks = 0;
        SPV_CHECK(fill_fold_jobs(folds + j0, std::min(FJ_MAX, nfolds - j0), fj, blocks) == 0, "spv_fold_multi: bad fold job");
        hipLaunchKernelGGL(fold_multi_kernel, dim3(blocks), dim3(FOLD_COLS * FOLD_ROWS / FOLD_NARROW), 0, st, fj);
        SPV_LAUNCH_CHECK("spv_fold_multi");
    }
    return 0;
}

static int gemm_tn_batch_impl(const spv_tn_problem* probs, int nprob, int K, int splits, void* workspace, const spv_fold_job* folds,
                              int nfolds, void* stream, int parts) {
    SPV_CHECK(probs != nullptr && nprob >= 1 && nprob <= TNB_MAX, "spv_gemm_tn_batch: 1..%d problems", TNB_MAX);
    SPV_CHECK(K > 0 && splits >= 1 && workspace != nullptr, "spv_gemm_tn_batch: K=%d splits=%d need a workspace", K, splits);
    SPV_CHECK(nfolds >= 0 && nfolds <= FJ_MAX && (nfolds == 0 || folds != nullptr), "spv_gemm_tn_batch: 0..%d fold jobs", FJ_MAX);
    TnBatch tb{};
    tb.nprob = nprob;
    int tiles = 0;
    long long off = 0;
    int k_per_split = cdiv(cdiv(K, splits), TBK) * TBK;
    splits = cdiv(K, k_per_split);
    int reduce_blocks = 1;
    // the long problems (k == 0 or k == K: the call's K rows, split-K) first, then the short ones (k < K: their own rows, unsplit, stored
    // by the GEMM launch itself); order[] maps the kernel's problem index to the caller's
    int order[TNB_MAX], nlong = 0, no = 0;
    for (int i = 0; i < nprob; ++i) {
        SPV_CHECK(probs[i].k >= 0 && probs[i].k <= K, "spv_gemm_tn_batch: problem %d: k=%d must be 0 (= K) or 1..K=%d", i, probs[i].k, K);
        if (probs[i].k == 0 || probs[i].k == K) order[no++] = i;
    }
    nlong = no;
    for (int i = 0; i < nprob; ++i)
        if (!(probs[i].k == 0 || probs[i].k == K)) order[no++] = i;
    SPV_CHECK(nlong >= 1, "spv_gemm_tn_batch: no problem reduces over the call's K=%d rows", K);
    tb.nlong = nlong;
    int long_tiles = 0;
    for (int i = 0; i < nprob; ++i) {
        const spv_tn_problem& q = probs[order[i]];
        SPV_CHECK(q.a != nullptr && q.b != nullptr && q.c != nullptr && q.m > 0 && q.n > 0, "spv_gemm_tn_batch: empty problem %d", order[i]);
        SPV_CHECK(q.m % 8 == 0 && q.n % 8 == 0 && q.lda % 8 == 0 && q.ldb % 8 == 0 && q.lda >= q.m && q.ldb >= q.n && q.ldc >= q.n,
                  "spv_gemm_tn_batch: problem %d: M=%d N=%d lda=%d ldb=%d ldc=%d (multiples of 8, leading dimensions >= extents)", order[i], q.m,
                  q.n, q.lda, q.ldb, q.ldc);
        SPV_CHECK(((uintptr_t)q.a & 15) == 0 && ((uintptr_t)q.b & 15) == 0, "spv_gemm_tn_batch: problem %d: A/B must be 16-byte aligned", order[i]);
        SPV_CHECK(i < nlong || q.ldc % 4 == 0, "spv_gemm_tn_batch: problem %d (short): ldc=%d must be a multiple of 4", order[i], q.ldc);
        tb.first_tile[i] = tiles;
        tb.ws_off[i] = off;
        tb.p[i] = {static_cast<const bf16_t*>(q.a), static_cast<const bf16_t*>(q.b), static_cast<float*>(q.c), q.m, q.n, q.lda, q.ldb, q.ldc, cdiv(q.n, BN),
                   i < nlong ? K : q.k};
        tiles += cdiv(q.m, BM) * cdiv(q.n, BN);
        if (i < nlong) {
            long_tiles = tiles;
            off += (long long)splits * q.m * q.n;
            constexpr int RT = FOLD_COLS * FOLD_ROWS;
            reduce_blocks = std::max(reduce_blocks, (int)std::min<int64_t>(((int64_t)q.m * q.n / ((q.n & 3) == 0 ? 4 : 1) + RT - 1) / RT, 2048));
        }
    }
    tb.first_tile[nprob] = tiles;
    hipStream_t st = static_cast<hipStream_t>(stream);
#ifdef SPV_LAB
    {
        static int last_abl = 0;
        const int abl = SPV_LAB_INT("SPV_TN_ABLATE", 0);
        if (abl != last_abl) {   // (a copy per launch cost every timed launch ~60 us of host synchronisation)
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tn_ablate), &abl, sizeof(abl));
            last_abl = abl;
        }
    }
#endif
    // the 256 x 128 tile when every problem divides into it and the K-slices are long enough to pay for its longer prologue
    static const int wide_env = SPV_LAB_INT("SPV_TNB_WIDE", 1);
    bool wide = wide_env != 0 && k_per_split >= 8 * TBK;
    for (int i = 0; i < nprob && wide; ++i) wide = probs[i].m % TWM == 0 && probs[i].n % BN == 0 && (probs[i].k == 0 || probs[i].k >= TBK);
    if (parts & 1) {
        SPV_COUNT_PATH(SPV_PATH_GEMM_TN);
        SPV_COUNT_PATH(SPV_PATH_GEMM_TN_BATCH);
        if (wide) {
            TnBatch tw = tb;
            int wt = 0, wlong = 0;
            for (int i = 0; i < nprob; ++i) {
                tw.first_tile[i] = wt;
                wt += (tb.p[i].M / TWM) * tb.p[i].tiles_n;
                if (i < nlong) wlong = wt;
            }
            tw.first_tile[nprob] = wt;
            const int wgrid = wlong * splits + ((wt - wlong + 7) & ~7);
            static const int wide_db = SPV_LAB_INT("SPV_TN_WIDE_DB", 1);   // two LDS buffers, one barrier per K-tile
            if (wide_db) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_batch_wide_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TW_SMEM_DB);
                hipLaunchKernelGGL(gemm_tn_batch_wide_kernel<true>, dim3(wgrid), dim3(512), TW_SMEM_DB, st, tw, static_cast<float*>(workspace), K, k_per_split, splits);
            } else {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_batch_wide_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TW_SMEM);
                hipLaunchKernelGGL(gemm_tn_batch_wide_kernel<false>, dim3(wgrid), dim3(512), TW_SMEM, st, tw, static_cast<float*>(workspace), K, k_per_split, splits);
            }
            SPV_COUNT_PATH(SPV_PATH_GEMM_TN_WIDE);
        } else {
            const int grid128 = long_tiles * splits + ((tiles - long_tiles + 7) & ~7);
#ifdef SPV_LAB
            static const int bdepth = SPV_LAB_INT("SPV_TNB_DEPTH", 3);
            if (SPV_LAB_INT("SPV_TNB_DB", 0) && k_per_split >= 8 * TBK) {
                constexpr int DBSMEM = 2 * 2 * TBK * TROWB;
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_batch_db_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DBSMEM);
                hipLaunchKernelGGL(gemm_tn_batch_db_kernel, dim3(grid128), dim3(256), DBSMEM, st, tb, static_cast<float*>(workspace), K, k_per_split, splits);
            } else if (bdepth == 1)
                hipLaunchKernelGGL(gemm_tn_batch_kernel<1>, dim3(grid128), dim3(256), 0, st, tb, static_cast<float*>(workspace), K, k_per_split, splits);
            else
#endif
                hipLaunchKernelGGL(gemm_tn_batch_kernel<3>, dim3(grid128), dim3(256), 0, st, tb, static_cast<float*>(workspace), K, k_per_split, splits);
        }
        SPV_LAUNCH_CHECK("spv_gemm_tn_batch");
    }
    if (parts & 2) {
        FoldJobs fj{};
        int fold_blocks = 0;
        SPV_CHECK(fill_fold_jobs(folds, nfolds, fj, fold_blocks) == 0, "spv_gemm_tn_batch: bad fold job");
        hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3(fold_blocks + nlong * reduce_blocks), dim3(FOLD_COLS * FOLD_ROWS), 0, st, tb,
                           static_cast<const float*>(workspace), splits, reduce_blocks, fj);
        SPV_LAUNCH_CHECK("spv_gemm_tn_batch(split-k reduce + fold)");
    }
    return 0;
}

extern "C" int spv_gemm_tn_batch(const spv_tn_problem* probs, int nprob, int K, int splits, void* workspace, const spv_fold_job* folds,
                                 int nfolds, void* stream) {
    return gemm_tn_batch_impl(probs, nprob, K, splits, workspace, folds, nfolds, stream, 3);
}

extern "C" int spv_gemm_tn_batch_part(const spv_tn_problem* probs, int nprob, int K, int splits, void* workspace, const spv_fold_job* folds,
                                      int nfolds, int part, void* stream) {
    SPV_CHECK(part == 1 || part == 2, "spv_gemm_tn_batch_part: part must be 1 (the GEMM launch) or 2 (the reduce + fold launch), got %d", part);
    return gemm_tn_batch_impl(probs, nprob, K, splits, workspace, folds, nfolds, stream, part);
}
