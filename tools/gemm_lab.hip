// gemm_lab.hip -- ablation timings of the NT GEMM tile loop at the layer shapes (development tool, not shipped).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_lab.hip -o tools/_build/gemm_lab && tools/_build/gemm_lab
// Re-uses the product kernels' building blocks by including the translation unit; adds copies of the tile loop with
// parts switched off (results are then wrong on purpose) to see which resource the loop is waiting for.
#include "../vit-spectre-experiments_amd/csrc/spv_gemm.hip"
#include "../vit-spectre-experiments_amd/csrc/spv_misc.hip"

#include <math.h>
#include <string.h>
#include <type_traits>
#include <algorithm>
#include <vector>

namespace {

// ABL bits: 1 = no C store, 2 = B staged once, 4 = A staged once, 8 = no MFMA
template <int WMB, int NST, int ABL>
__global__ __launch_bounds__(256) void lab_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                  int M, int N, int K, int tiles_n, int tiles_mn) {
    constexpr int WBM = 64 * WMB, KB = 64, KE = KB / 2;
    constexpr int STAGE = (WBM + BN) * KB;
    constexpr int IPA = WBM / 16 / 4, IPB = BN / 16 / 4;
    constexpr int SM = NST * STAGE > 36864 ? NST * STAGE : 36864;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, tiles_mn);
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * WBM, n0 = tn * BN;
    const bf16_t* asrc[IPA];
    const bf16_t* bsrc[IPB];
#pragma unroll
    for (int t = 0; t < IPA; ++t) {
        const int R = 16 * (wave * IPA + t) + (lane >> 2);
        const int c = (lane & 3) ^ ((R >> 2) & 3);
        asrc[t] = A + (size_t)min(m0 + R, M - 1) * K + c * 8;
    }
#pragma unroll
    for (int t = 0; t < IPB; ++t) {
        const int R = 16 * (wave * IPB + t) + (lane >> 2);
        const int c = (lane & 3) ^ ((R >> 2) & 3);
        bsrc[t] = B + (size_t)min(n0 + R, N - 1) * K + c * 8;
    }
    auto stage = [&](int buf, int k0, bool first) {
        unsigned char* sa = smem + buf * STAGE + wave * IPA * 1024;
        unsigned char* sb = smem + buf * STAGE + WBM * KB + wave * IPB * 1024;
        if (first || !(ABL & 4)) {
#pragma unroll
            for (int t = 0; t < IPA; ++t)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[t] + k0),
                                                 (__attribute__((address_space(3))) void*)(sa + t * 1024), 16, 0, 0);
        }
        if (first || !(ABL & 2)) {
#pragma unroll
            for (int t = 0; t < IPB; ++t)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[t] + k0),
                                                 (__attribute__((address_space(3))) void*)(sb + t * 1024), 16, 0, 0);
        }
    };
    f32x16 acc[WMB][2];
#pragma unroll
    for (int i = 0; i < WMB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int frow = lane & 31, fh = lane >> 5, swz = (frow >> 2) & 3;
    const int fa_off = (wm * 32 * WMB + frow) * KB, fb_off = WBM * KB + (wn * 64 + frow) * KB;
    const int nk = K / KE;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) stage(s, s * KE, true);
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!(ABL & 16) && t + NST - 1 < nk) stage((buf + NST - 1) % NST, (t + NST - 1) * KE, false);
        const unsigned char* sa = smem + buf * STAGE + fa_off;
        const unsigned char* sb = smem + buf * STAGE + fb_off;
        if (ABL & 16) {
            // all fragment reads of the stage first, then the MFMAs with the next stage's DMA pieces spread between them
            auto body = [&](auto more_tag) {
                constexpr bool MORE = decltype(more_tag)::value;
                bf16x8 a[2][WMB], b[2][2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int ch = ((ks * 2 + fh) ^ swz) * 16;
#pragma unroll
                    for (int f = 0; f < 2; ++f) b[ks][f] = *reinterpret_cast<const bf16x8*>(sb + f * 32 * KB + ch);
#pragma unroll
                    for (int f = 0; f < WMB; ++f) a[ks][f] = *reinterpret_cast<const bf16x8*>(sa + f * 32 * KB + ch);
                }
                const int nb = (buf + NST - 1) % NST, k0 = (t + NST - 1) * KE;
                unsigned char* da = smem + nb * STAGE + wave * IPA * 1024;
                unsigned char* db = smem + nb * STAGE + WBM * KB + wave * IPB * 1024;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < WMB; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
                            constexpr int EVERY = (4 * WMB) / (IPA + IPB);
                            const int m = (ks * WMB + i) * 2 + j;  // MFMA index 0 .. 4*WMB-1
                            const int piece = m / EVERY;
                            if (MORE && (m % EVERY) == EVERY - 1 && piece < IPA + IPB) {
                                if (piece < IPA)
                                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[piece < IPA ? piece : 0] + k0),
                                                                     (__attribute__((address_space(3))) void*)(da + piece * 1024), 16, 0, 0);
                                else
                                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[piece >= IPA ? piece - IPA : 0] + k0),
                                                                     (__attribute__((address_space(3))) void*)(db + (piece - IPA) * 1024), 16, 0, 0);
                            }
                        }
            };
            if (t + NST - 1 < nk) body(std::true_type{});
            else body(std::false_type{});
            buf = (buf + 1 == NST) ? 0 : buf + 1;
            continue;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ((ks * 2 + fh) ^ swz) * 16;
            bf16x8 a[WMB], b[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) b[f] = *reinterpret_cast<const bf16x8*>(sb + f * 32 * KB + ch);
#pragma unroll
            for (int f = 0; f < WMB; ++f) a[f] = *reinterpret_cast<const bf16x8*>(sa + f * 32 * KB + ch);
            if (!(ABL & 8)) {
#pragma unroll
                for (int i = 0; i < WMB; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < WMB; ++i) acc[i][0][0] += (float)a[i][0] + (float)b[i & 1][1];
            }
        }
        buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
    if (ABL & 1) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < WMB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 1234.5678f) C[0] = 1;
        return;
    }
#pragma unroll
    for (int h = 0; h < WMB / 2; ++h)
        store_acc_tile<bf16_t>(acc + 2 * h, smem, nullptr, C, nullptr, M, N, N, 0, m0, n0, 0, 0, 0, 0, nullptr, nullptr, 0, 0,
                               wm * 32 * WMB + 64 * h);
}

// ---------------------------------------------------------------------------------------------------------
// full-N tile: one 512-thread workgroup per 128 rows x ALL N columns (N = 4 * NBLK * 32), 8 waves as 2 (M) x 4 (N),
// wave tile 64 x (NBLK*32).  A is read once from HBM, B (the weight) streams from L2 once per workgroup.  K staged 16
// elements (32-byte rows) per stage, NST-deep LDS ring filled by LDS-DMA with counted vmcnt, one barrier per stage.
// Accumulators hold C^T blocks (operands swapped) so that a lane owns 4 consecutive columns of one row: 8-byte stores,
// no LDS epilogue.
template <int NBLK, int NST, int ABL>
__global__ __launch_bounds__(512) void fulln_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                    int M, int K) {
    constexpr int N = 4 * NBLK * 32, NW = NBLK * 32;
    constexpr int ROWS = 128 + N;           // tile rows per stage (A then B), 32 B each
    constexpr int STAGE = ROWS * 32;
    constexpr int NI = ROWS / 32;           // DMA wave-instructions per stage
    constexpr int LPW_HI = (NI + 7) / 8, REM = NI % 8;  // waves < REM issue LPW_HI, the others LPW_HI - 1 (REM == 0: all LPW_HI)
    constexpr int D = NST - 1;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int m0 = blockIdx.x * 128;
    // per-lane source pointers for this wave's DMA instructions
    const bf16_t* src[LPW_HI];
#pragma unroll
    for (int t = 0; t < LPW_HI; ++t) {
        const int i = wave + 8 * t;                 // instruction index within the stage
        const int R = 32 * i + (lane >> 1);         // tile row (A rows first)
        const int c = (lane & 1) ^ ((R >> 3) & 1);  // logical 16-byte chunk stored at this lane's LDS slot
        if (i < NI) src[t] = R < 128 ? A + (size_t)min(m0 + R, M - 1) * K + c * 8 : B + (size_t)(R - 128) * K + c * 8;
        else src[t] = A;
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int t = 0; t < LPW_HI; ++t) {
            const int i = wave + 8 * t;
            if (i < NI)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + k0),
                                                 (__attribute__((address_space(3))) void*)(smem + buf * STAGE + i * 1024), 16, 0, 0);
        }
    };
    f32x16 acc[2][NBLK];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NBLK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int frow = lane & 31, fh = lane >> 5;
    const int pch = (fh ^ ((frow >> 3) & 1)) * 16;  // 32-row block offsets do not change the swizzle bit
    const int fa_off = (wm * 64 + frow) * 32 + pch, fb_off = (128 + wn * NW + frow) * 32 + pch;
    const int nk = K / 16;
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < nk) stage(s, s * 16);
    int buf = 0;
    const bool hi = REM == 0 || wave < REM;
    for (int t = 0; t < nk; ++t) {
        if (t + D - 1 < nk) {
            if (hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * LPW_HI) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * (LPW_HI - 1)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (t + D < nk && !(ABL & 6)) stage((buf + D) % NST, (t + D) * 16);
        const unsigned char* sa = smem + buf * STAGE + fa_off;
        const unsigned char* sb = smem + buf * STAGE + fb_off;
        bf16x8 a[2], b[NBLK];
#pragma unroll
        for (int f = 0; f < 2; ++f) a[f] = *reinterpret_cast<const bf16x8*>(sa + f * 32 * 32);
#pragma unroll
        for (int f = 0; f < NBLK; ++f) b[f] = *reinterpret_cast<const bf16x8*>(sb + f * 32 * 32);
        if (!(ABL & 8)) {
#pragma unroll
            for (int j = 0; j < NBLK; ++j)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < NBLK; ++j) acc[j & 1][j][0] += (float)a[j & 1][0] + (float)b[j][1];
        }
        buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
    if (ABL & 1) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NBLK; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 1234.5678f) C[0] = 1;
        return;
    }
    // C^T blocks: lane -> row m = frow, registers r -> column (r & 3) + 8 (r >> 2) + 4 fh
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = m0 + wm * 64 + i * 32 + frow;
        if (row >= M) continue;
        bf16_t* cp = C + (size_t)row * N + wn * NW + 4 * fh;
#pragma unroll
        for (int j = 0; j < NBLK; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 o;
                o.x = (unsigned)f2bf(acc[i][j][4 * q]) | ((unsigned)f2bf(acc[i][j][4 * q + 1]) << 16);
                o.y = (unsigned)f2bf(acc[i][j][4 * q + 2]) | ((unsigned)f2bf(acc[i][j][4 * q + 3]) << 16);
                *reinterpret_cast<uint2*>(cp + j * 32 + 8 * q) = o;
            }
    }
}

template <int NBLK, int NST, int ABL>
float run_fulln(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int K, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = (M + 127) / 128;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fulln_kernel<NBLK, NST, ABL>), dim3(grid), dim3(512), 0, 0, A, B, C, M, K);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((fulln_kernel<NBLK, NST, ABL>), dim3(grid), dim3(512), 0, 0, A, B, C, M, K);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
    return ms * 1e3f / iters;
}

// reference check of a few rows on the host
static double check_rows(const std::vector<bf16_t>& hA, const std::vector<bf16_t>& hB, const bf16_t* dC, int M, int N, int K) {
    auto f = [](bf16_t v) { unsigned u = ((unsigned)v) << 16; float x; memcpy(&x, &u, 4); return x; };
    const int rows[] = {0, 1, 31, 32, 63, 64, 127, 128, 129, 1000, M - 129, M - 1};
    std::vector<bf16_t> hc(N);
    double worst = 0;
    for (int r : rows) {
        hipMemcpy(hc.data(), dC + (size_t)r * N, N * 2, hipMemcpyDeviceToHost);
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)f(hA[(size_t)r * K + k]) * f(hB[(size_t)n * K + k]);
            const double d = fabs(s - f(hc[n])) / (fabs(s) + 1.0);
            if (d > worst) worst = d;
        }
    }
    return worst;
}

// 128 x (64 WN) tile, 2 x WN waves of 64 x 64 each (WN = 4: 128 x 256, 512 threads): same wave tile as the 128 x 128 kernel,
// 25 % fewer operand bytes through the L2 -> LDS path
template <int WN, int NST, int ABL>
__global__ __launch_bounds__(128 * WN) void lab_kernel_n(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                         int M, int N, int K, int tiles_n, int tiles_mn) {
    constexpr int NWV = 2 * WN, TBN = 64 * WN, KB = 64, KE = KB / 2;
    constexpr int STAGE = (128 + TBN) * KB;
    constexpr int NI = (128 + TBN) / 16;  // DMA instructions per stage (16 rows each)
    constexpr int IPW = NI / NWV;         // per wave (WN = 4: 24 / 8 = 3; WN = 2: 16 / 4 = 4)
    constexpr int SM = NST * STAGE > NWV * 9216 ? NST * STAGE : NWV * 9216;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_remap(blockIdx.x, tiles_mn);
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * 128, n0 = tn * TBN;
    const bf16_t* src[IPW];
#pragma unroll
    for (int t = 0; t < IPW; ++t) {
        const int R = 16 * (wave * IPW + t) + (lane >> 2);  // stage row: A rows 0..127 then B rows
        const int c = (lane & 3) ^ ((R >> 2) & 3);
        src[t] = R < 128 ? A + (size_t)min(m0 + R, M - 1) * K + c * 8 : B + (size_t)min(n0 + R - 128, N - 1) * K + c * 8;
    }
    auto stage = [&](int buf, int k0) {
        unsigned char* d = smem + buf * STAGE + wave * IPW * 1024;
#pragma unroll
        for (int t = 0; t < IPW; ++t)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + k0),
                                             (__attribute__((address_space(3))) void*)(d + t * 1024), 16, 0, 0);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int frow = lane & 31, fh = lane >> 5, swz = (frow >> 2) & 3;
    const int fa_off = (wm * 64 + frow) * KB, fb_off = (128 + wn * 64 + frow) * KB;
    const int nk = K / KE;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) stage(s, s * KE);
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
        if (NST >= 3 && t + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char* sa = smem + buf * STAGE + fa_off;
        const unsigned char* sb = smem + buf * STAGE + fb_off;
        if (ABL & 32) {
            // fragment reads first, the next stage's DMA pieces behind them (their ~170-cycle issue each hides the LDS
            // latency), MFMAs last
            bf16x8 a[2][2], b[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int ch = ((ks * 2 + fh) ^ swz) * 16;
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    a[ks][f] = *reinterpret_cast<const bf16x8*>(sa + f * 32 * KB + ch);
                    b[ks][f] = *reinterpret_cast<const bf16x8*>(sb + f * 32 * KB + ch);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t + NST - 1 < nk) stage((buf + NST - 1) % NST, (t + NST - 1) * KE);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
            buf = (buf + 1 == NST) ? 0 : buf + 1;
            continue;
        }
        if (t + NST - 1 < nk) stage((buf + NST - 1) % NST, (t + NST - 1) * KE);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
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
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
    if (ABL & 1) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 1234.5678f) C[0] = 1;
        return;
    }
    store_acc_tile<bf16_t>(acc, smem, nullptr, C, nullptr, M, N, N, 0, m0, n0, 0, 0, 0, 0, nullptr, nullptr, 0, 0, wm * 64, wn * 64);
}

// stamped 128 x 128 tile loop (wave 0 of every workgroup): [0] start, then per K step (wait done, MFMAs issued), epilogue end
__device__ __forceinline__ unsigned long long lab_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
template <int NST>
__global__ __launch_bounds__(256) void lab_stamp_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                        int M, int N, int K, int tiles_n, int tiles_mn, unsigned long long* __restrict__ stamps) {
    constexpr int WN = 2, NWV = 4, TBN = 128, KB = 64, KE = KB / 2;
    constexpr int STAGE = (128 + TBN) * KB;
    constexpr int IPW = (128 + TBN) / 16 / NWV;
    constexpr int SM = NST * STAGE > NWV * 9216 ? NST * STAGE : NWV * 9216;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[SM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    unsigned long long* my = stamps + (size_t)blockIdx.x * 64;
    const bool rec = tid == 0;
    if (rec) { my[52] = __builtin_amdgcn_s_memrealtime(); my[0] = lab_now(); }
    const int tile = xcd_remap(blockIdx.x, tiles_mn);
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int m0 = tm * 128, n0 = tn * TBN;
    const bf16_t* src[IPW];
#pragma unroll
    for (int t = 0; t < IPW; ++t) {
        const int R = 16 * (wave * IPW + t) + (lane >> 2);
        const int c = (lane & 3) ^ ((R >> 2) & 3);
        src[t] = R < 128 ? A + (size_t)min(m0 + R, M - 1) * K + c * 8 : B + (size_t)min(n0 + R - 128, N - 1) * K + c * 8;
    }
    auto stage = [&](int buf, int k0) {
        unsigned char* d = smem + buf * STAGE + wave * IPW * 1024;
#pragma unroll
        for (int t = 0; t < IPW; ++t)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + k0),
                                             (__attribute__((address_space(3))) void*)(d + t * 1024), 16, 0, 0);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int frow = lane & 31, fh = lane >> 5, swz = (frow >> 2) & 3;
    const int fa_off = (wm * 64 + frow) * KB, fb_off = (128 + wn * 64 + frow) * KB;
    const int nk = K / KE;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) stage(s, s * KE);
    int buf = 0;
    for (int t = 0; t < nk; ++t) {
        if (NST >= 3 && t + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (rec && t < 24) my[1 + 2 * t] = lab_now();
        if (t + NST - 1 < nk) stage((buf + NST - 1) % NST, (t + NST - 1) * KE);
        __builtin_amdgcn_sched_barrier(0);
        if (rec && t == 4) my[40] = lab_now();
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sa = smem + buf * STAGE + fa_off;
        const unsigned char* sb = smem + buf * STAGE + fb_off;
        bf16x8 a[2][2], b[2][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ((ks * 2 + fh) ^ swz) * 16;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                a[ks][f] = *reinterpret_cast<const bf16x8*>(sa + f * 32 * KB + ch);
                b[ks][f] = *reinterpret_cast<const bf16x8*>(sb + f * 32 * KB + ch);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (rec && t == 4) my[41] = lab_now();  // includes lgkmcnt(0): fragments are in registers
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (rec && t < 24) my[2 + 2 * t] = lab_now();
        buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
    if (rec) my[50] = lab_now();
    store_acc_tile<bf16_t>(acc, smem, nullptr, C, nullptr, M, N, N, 0, m0, n0, 0, 0, 0, 0, nullptr, nullptr, 0, 0, wm * 64, wn * 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (rec) { my[51] = lab_now(); my[53] = __builtin_amdgcn_s_memrealtime(); }
}

template <int NST>
void run_stamps(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K) {
    const int tiles_n = N / 128, tiles_m = (M + 127) / 128, nt = tiles_m * tiles_n;
    unsigned long long* d;
    hipMalloc(&d, (size_t)nt * 64 * 8);
    for (int i = 0; i < 3; ++i) {
        hipMemset(d, 0, (size_t)nt * 64 * 8);
        hipLaunchKernelGGL((lab_stamp_kernel<NST>), dim3(nt), dim3(256), 0, 0, A, B, C, M, N, K, tiles_n, nt, d);
    }
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)nt * 64);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nt; ++b) { t0 = std::min(t0, h[(size_t)b * 64]); t1 = std::max(t1, h[(size_t)b * 64 + 51]); }
    const int nk = std::min(K / 32, 24);
    auto med = [&](std::vector<long long>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::vector<long long> start, wait0, waitk, mfma, epi, total;
    for (int b = 0; b < nt; ++b) {
        const unsigned long long* s = &h[(size_t)b * 64];
        start.push_back((long long)(s[0] - t0));
        wait0.push_back((long long)(s[1] - s[0]));
        for (int t = 1; t < nk; ++t) waitk.push_back((long long)(s[1 + 2 * t] - s[2 * t]));
        for (int t = 0; t < nk; ++t) mfma.push_back((long long)(s[2 + 2 * t] - s[1 + 2 * t]));
        epi.push_back((long long)(s[51] - s[50]));
        total.push_back((long long)(s[51] - s[0]));
    }
    {   // in-kernel clock: shader cycles per 100 MHz tick, per workgroup; and the kernel's span on the constant-rate counter
        std::vector<long long> mhz;
        unsigned long long r0 = ~0ull, r1 = 0;
        for (int b = 0; b < nt; ++b) {
            const unsigned long long* s2 = &h[(size_t)b * 64];
            if (s2[53] > s2[52]) mhz.push_back((long long)((double)(s2[51] - s2[0]) / (double)(s2[53] - s2[52]) * 100.0));
            r0 = std::min(r0, s2[52]);
            r1 = std::max(r1, s2[53]);
        }
        std::vector<long long> life;
        for (int b = 0; b < nt; ++b) life.push_back((long long)(h[(size_t)b * 64 + 53] - h[(size_t)b * 64 + 52]));
        std::vector<long long> st;
        for (int b = 0; b < nt; ++b) st.push_back((long long)(h[(size_t)b * 64 + 52] - r0));
        std::sort(st.begin(), st.end());
        printf("clock: in-kernel shader clock median %lld MHz | kernel span %.2f us (100 MHz counter) | WG lifetime median %.2f us | "
               "WG start times: 25%% %.2f us, 50%% %.2f, 75%% %.2f, 99%% %.2f\n",
               med(mhz), (double)(r1 - r0) / 100.0, (double)med(life) / 100.0, st[nt / 4] / 100.0, st[nt / 2] / 100.0, st[nt * 3 / 4] / 100.0,
               st[nt * 99 / 100] / 100.0);
    }
    {
        std::vector<long long> a, b, c;
        for (int bb = 0; bb < nt; ++bb) {
            const unsigned long long* s2 = &h[(size_t)bb * 64];
            a.push_back((long long)(s2[40] - s2[1 + 2 * 4]));
            b.push_back((long long)(s2[41] - s2[40]));
            c.push_back((long long)(s2[2 + 2 * 4] - s2[41]));
        }
        printf("step 4 split (cycles): DMA issue %lld | fragment reads until data %lld | 8 MFMAs issued %lld\n", med(a), med(b), med(c));
    }
    std::vector<long long> st2 = start;
    std::sort(st2.begin(), st2.end());
    printf("stamps (cycles, stages %d, %d tiles): kernel span %llu | WG start median %lld, 90%% %lld, max %lld | first wait %lld | later waits %lld | "
           "issue+MFMA per step %lld | epilogue (incl. store drain) %lld | WG lifetime %lld\n",
           NST, nt, (unsigned long long)(t1 - t0), med(start), st2[st2.size() * 9 / 10], st2.back(), med(wait0), med(waitk), med(mfma), med(epi),
           med(total));
    hipFree(d);
}

// ---------------------------------------------------------------------------------------------------------
// persistent 128 x 128 kernel: workgroup w processes tiles w, w + G, w + 2G, ...; the K stages of all its tiles form ONE
// stream through a 3-slot LDS ring (prefetch distance 2), so the first stages of the next tile are in flight while the
// current tile finishes; the epilogue is staged 16 rows at a time through the ring slot the last stage just vacated
// (wave-private 4 KiB, no workgroup barrier inside) and its global stores drain under the next tile's K loop
// (counted vmcnt: loads, LDS-DMA and stores retire in issue order).
template <int ABL>
__global__ __launch_bounds__(256) void lab_persist_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C,
                                                          int M, int N, int K, int tiles_n, int tiles_mn) {
    constexpr int NST = 3, KB = 64, KE = KB / 2, STAGE = 256 * KB, IPW = 4;  // 16 KiB per stage, 4 DMA pieces per wave
    constexpr int NSTORE = 8;                                                 // epilogue global stores per wave
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = K / KE;
    const int G = gridDim.x;
    const int n_mine = (tiles_mn - (int)blockIdx.x + G - 1) / G;
    const int total = n_mine * nk;  // stages in this workgroup's stream
    // prefetch cursor
    int p_tile = 0, p_k = 0, p_slot = 0;
    const bf16_t* src[IPW];
    auto set_src = [&](int ti) {
        const int tile = xcd_remap((int)blockIdx.x + ti * G, tiles_mn);
        const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
#pragma unroll
        for (int t = 0; t < IPW; ++t) {
            const int R = 16 * (wave * IPW + t) + (lane >> 2);
            const int c = (lane & 3) ^ ((R >> 2) & 3);
            src[t] = R < 128 ? A + (size_t)min(m0 + R, M - 1) * K + c * 8 : B + (size_t)min(n0 + R - 128, N - 1) * K + c * 8;
        }
    };
    auto issue = [&]() {  // next stage of the stream into p_slot
        unsigned char* d = smem + p_slot * STAGE + wave * IPW * 1024;
#pragma unroll
        for (int t = 0; t < IPW; ++t)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[t] + p_k * KE),
                                             (__attribute__((address_space(3))) void*)(d + t * 1024), 16, 0, 0);
        p_slot = (p_slot + 1 == NST) ? 0 : p_slot + 1;
        if (++p_k == nk) { p_k = 0; ++p_tile; if (p_tile < n_mine) set_src(p_tile); }
    };
    if (n_mine <= 0) return;
    set_src(0);
    int issued = 0;
    for (; issued < NST - 1 && issued < total; ++issued) issue();
    const int frow = lane & 31, fh = lane >> 5, swz = (frow >> 2) & 3;
    const int fa_off = (wm * 64 + frow) * KB, fb_off = (128 + wn * 64 + frow) * KB;
    int slot = 0, done = 0;
    bool after_epi = false;
    for (int ti = 0; ti < n_mine; ++ti) {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        for (int ks = 0; ks < nk; ++ks, ++done) {
            // stage `done` must have landed; younger operations: the next stage's pieces (if issued) and, right after an
            // epilogue, its stores
            if (issued - done >= 2) {
                if (after_epi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW + NSTORE) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
            } else {
                if (after_epi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            after_epi = false;
            __builtin_amdgcn_s_barrier();
            if (issued < total && !(ABL & 6)) { issue(); }
            if (issued < total) ++issued;
            const unsigned char* sa = smem + slot * STAGE + fa_off;
            const unsigned char* sb = smem + slot * STAGE + fb_off;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int ch = ((k2 * 2 + fh) ^ swz) * 16;
                bf16x8 a[2], b[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    a[f] = *reinterpret_cast<const bf16x8*>(sa + f * 32 * KB + ch);
                    b[f] = *reinterpret_cast<const bf16x8*>(sb + f * 32 * KB + ch);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            slot = (slot + 1 == NST) ? 0 : slot + 1;
        }
        // epilogue through the slot the last stage just vacated (= previous `slot`), after everyone has read it
        const int eslot = (slot + NST - 1) % NST;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ABL & 1) {
            float sacc = 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
            if (sacc == 1234.5678f) C[0] = 1;
            continue;
        }
        const int tile = xcd_remap((int)blockIdx.x + ti * G, tiles_mn);
        const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
        float* stg = reinterpret_cast<float*>(smem + eslot * STAGE + wave * 4096);  // [16 rows][64 cols] fp32
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int i = pass >> 1, hr = pass & 1;  // 32-row block i, rows 16 hr .. 16 hr + 15 of it: registers r with (r >> 2) in {2 hr, 2 hr + 1}
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = 8 * hr + rr;
                    const int row16 = (r & 3) + 8 * ((r >> 2) & 1) + 4 * fh;  // row within the 16-row slab
                    stg[row16 * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // read back: lane -> (row = lane >> 2, 16 consecutive columns), two 16-byte stores
            const int lr = lane >> 2, c16 = (lane & 3) * 16;
            const int row = m0 + wm * 64 + i * 32 + hr * 16 + lr;
            float v[16];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 t4 = *reinterpret_cast<const float4*>(stg + lr * 64 + c16 + 4 * u);
                v[4 * u] = t4.x; v[4 * u + 1] = t4.y; v[4 * u + 2] = t4.z; v[4 * u + 3] = t4.w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (row < M) {
                bf16_t* cp = C + (size_t)row * N + n0 + wn * 64 + c16;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    uint4 o;
                    o.x = (unsigned)f2bf(v[8 * u]) | ((unsigned)f2bf(v[8 * u + 1]) << 16);
                    o.y = (unsigned)f2bf(v[8 * u + 2]) | ((unsigned)f2bf(v[8 * u + 3]) << 16);
                    o.z = (unsigned)f2bf(v[8 * u + 4]) | ((unsigned)f2bf(v[8 * u + 5]) << 16);
                    o.w = (unsigned)f2bf(v[8 * u + 6]) | ((unsigned)f2bf(v[8 * u + 7]) << 16);
                    *reinterpret_cast<uint4*>(cp + 8 * u) = o;
                }
            }
        }
        after_epi = true;
    }
}

template <int ABL>
float run_persist(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, int grid, int iters) {
    const int tiles_n = N / 128, tiles_m = (M + 127) / 128, nt = tiles_m * tiles_n;
    grid = std::min(grid, nt);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((lab_persist_kernel<ABL>), dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K, tiles_n, nt);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((lab_persist_kernel<ABL>), dim3(grid), dim3(256), 0, 0, A, B, C, M, N, K, tiles_n, nt);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
    return ms * 1e3f / iters;
}

template <int WN, int NST, int ABL>
float run_n(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, int iters) {
    const int tiles_n = N / (64 * WN), tiles_m = (M + 127) / 128;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((lab_kernel_n<WN, NST, ABL>), dim3(tiles_m * tiles_n), dim3(128 * WN), 0, 0, A, B, C, M, N, K, tiles_n, tiles_m * tiles_n);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((lab_kernel_n<WN, NST, ABL>), dim3(tiles_m * tiles_n), dim3(128 * WN), 0, 0, A, B, C, M, N, K, tiles_n, tiles_m * tiles_n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

template <int WMB, int NST, int ABL>
float run(const bf16_t* A, const bf16_t* B, bf16_t* C, int M, int N, int K, int iters) {
    const int tiles_n = N / BN, tiles_m = (M + 64 * WMB - 1) / (64 * WMB);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((lab_kernel<WMB, NST, ABL>), dim3(tiles_m * tiles_n), dim3(256), 0, 0, A, B, C, M, N, K, tiles_n, tiles_m * tiles_n);
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((lab_kernel<WMB, NST, ABL>), dim3(tiles_m * tiles_n), dim3(256), 0, 0, A, B, C, M, N, K, tiles_n, tiles_m * tiles_n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

}  // namespace

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 33280, N = argc > 2 ? atoi(argv[2]) : 768, K = argc > 3 ? atoi(argv[3]) : 512;
    bf16_t *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 2);
    hipMalloc(&B, (size_t)N * K * 2);
    hipMalloc(&C, (size_t)M * N * 2);
    std::vector<bf16_t> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (bf16_t)(0x3c00 + (i * 2654435761u >> 20 & 0x1ff) + ((i & 1) << 15));
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
    const double gf = 2.0 * M * N * K * 1e-6;
    printf("M=%d N=%d K=%d\n", M, N, K);
    std::vector<bf16_t> hB(h.begin(), h.begin() + (size_t)N * K);
#define RUNF(NB, S, ABL, label)                                                                   \
    {                                                                                             \
        float us = run_fulln<NB, S, ABL>(A, B, C, M, K, 20);                                      \
        printf("full-N %3d stages %d %-34s %8.2f us  %7.1f TFLOP/s", NB * 128, S, label, us, gf / us); \
        if (ABL == 0) printf("  max rel err %.2e", check_rows(h, hB, C, M, N, K));                \
        printf("\n");                                                                             \
        fflush(stdout);                                                                           \
    }
    if (N == 768) {
        RUNF(6, 5, 0, "full")
        RUNF(6, 5, 1, "no C store")
        RUNF(6, 5, 6, "staged once")
        RUNF(6, 5, 7, "staged once, no C store")
        RUNF(6, 5, 8, "no MFMA")
        RUNF(6, 4, 0, "full")
        RUNF(6, 3, 0, "full")
    }
    if (N == 512) {
        RUNF(4, 6, 0, "full")
        RUNF(4, 6, 1, "no C store")
        RUNF(4, 6, 6, "staged once")
        RUNF(4, 6, 7, "staged once, no C store")
        RUNF(4, 4, 0, "full")
        RUNF(4, 7, 0, "full")
    }
#define RUN(W, S, ABL, label)                                                                     \
    {                                                                                             \
        float us = run<W, S, ABL>(A, B, C, M, N, K, 20);                                          \
        printf("tile %3dx128 stages %d %-34s %8.2f us  %7.1f TFLOP/s\n", 64 * W, S, label, us, gf / us); \
        fflush(stdout);                                                                           \
    }
#define RUNN(WN, S, ABL, label)                                                                   \
    {                                                                                             \
        float us = run_n<WN, S, ABL>(A, B, C, M, N, K, 20);                                       \
        printf("tile 128x%3d stages %d %-34s %8.2f us  %7.1f TFLOP/s", 64 * WN, S, label, us, gf / us); \
        if (ABL == 0) printf("  max rel err %.2e", check_rows(h, hB, C, M, N, K));                \
        printf("\n");                                                                             \
        fflush(stdout);                                                                           \
    }
#define RUNP(GRID, ABL, label)                                                                    \
    {                                                                                             \
        float us = run_persist<ABL>(A, B, C, M, N, K, GRID, 20);                                  \
        printf("persistent 128x128 grid %4d %-26s %8.2f us  %7.1f TFLOP/s", GRID, label, us, gf / us); \
        if (ABL == 0) printf("  max rel err %.2e", check_rows(h, hB, C, M, N, K));                \
        printf("\n");                                                                             \
        fflush(stdout);                                                                           \
    }
    RUNP(768, 0, "full")
    RUNP(768, 1, "no C store")
    RUNP(520, 0, "full")
    RUNP(780, 0, "full")
    RUNP(1024, 0, "full")
    RUNP(512, 0, "full")
    run_stamps<2>(A, B, C, M, N, K);
    run_stamps<3>(A, B, C, M, N, K);
    RUNN(4, 2, 0, "8 waves full")
    RUNN(4, 2, 1, "8 waves no C store")
    RUNN(4, 3, 0, "8 waves full")
    RUNN(4, 3, 1, "8 waves no C store")
    RUNN(2, 2, 32, "4 waves, reads before DMA")
    RUNN(2, 3, 32, "4 waves, reads before DMA")
    RUNN(4, 2, 32, "8 waves, reads before DMA")
    RUNN(4, 3, 32, "8 waves, reads before DMA")
    RUNN(2, 2, 0, "4 waves full")
    RUNN(2, 3, 0, "4 waves full")
    RUN(2, 2, 0, "full")
    RUN(2, 2, 1, "no C store")
    RUN(2, 2, 2, "B staged once")
    RUN(2, 2, 4, "A staged once")
    RUN(2, 2, 6, "A and B staged once")
    RUN(2, 2, 7, "A, B once, no C store")
    RUN(2, 2, 8, "no MFMA")
    RUN(2, 2, 9, "no MFMA, no C store")
    RUN(2, 3, 0, "full")
    RUN(2, 3, 1, "no C store")
    RUN(4, 2, 0, "full")
    RUN(4, 2, 1, "no C store")
    RUN(4, 2, 6, "A and B staged once")
    RUN(4, 2, 7, "A, B once, no C store")
    RUN(4, 3, 0, "full")
    RUN(4, 3, 1, "no C store")
    return 0;
}
