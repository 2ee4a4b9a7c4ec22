// gemm_ws_lab.hip -- development bench: the strip GEMM's K loop with DEDICATED LOADER WAVES.
//
// Round 2's lab (tools/gemm_w4_lab.hip) found that every structure in which the MFMA waves also issue the LDS-DMA pieces lands at
// 1.1-1.25 PFLOP/s, while the same loops without the DMA reach 1.4-1.5: a piece costs its issuing wave 60-185 cycles.  Here the
// 256 x 256 x 64 tile loop runs on a 12-wave workgroup: waves 0-7 only read fragments and issue MFMAs (wave tile 128 x 64, 128
// accumulator registers, three waves per SIMD => <= 168 registers), waves 8-11 only issue the 64 one-KiB `global_load_lds` pieces of
// the NEXT K-tile and wait for them; one s_barrier per K-tile hands the buffer over (two 64 KiB stages).
//   L=vit-spectre-experiments_amd/lib; hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/gemm_ws_lab.hip -L$L -lspv_hip -Wl,-rpath,$PWD/$L -o gpurun_out/gemm_ws_lab
//   gpurun_out/gemm_ws_lab M N K      (M, N multiples of 256, K of 64; bit-for-bit check against the shipped kernel)
#include "../vit-spectre-experiments_amd/csrc/spv_common.h"
#include "../include/spv.h"

#include <math.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace {

__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = id & 7;
    return ((x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

constexpr int KD = 64, ROWB = 128, STAGE = 512 * ROWB, NST = 2, KS = 4;

// MODE 0: loaders wait for their pieces in front of the barrier (one K-tile in flight under the MFMAs of the current one).
template <int MODE>
__global__ __launch_bounds__(768) void ws_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, const float* __restrict__ bias,
                                                 bf16_t* __restrict__ C, int M, int N, int K, int nstrips, int nwg, int store) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int id = xcd_remap(blockIdx.x, nwg);
    const int g = id / nstrips, strip = id % nstrips;
    const int m0 = g * 256, n0 = strip * 256;
    const int nkt = K / KD;

    if (wave >= 8) {
        // ---- loader wave: 16 pieces per K-tile.  Waves 8, 9: the A rows (pieces 0..31, 8 rows of 128 B each), waves 10, 11: B.
        const int lw = wave - 8;
        const bool isb = lw >= 2;
        const bf16_t* src = isb ? B + (size_t)n0 * K : A + (size_t)m0 * K;
        const int pbase = (lw & 1) * 16;               // first piece of this wave inside its operand
        // row R = 8 piece + lane / 8; chunk c = (lane % 8) ^ ((R >> 1) & 7) = (lane % 8) ^ ((4 piece + (lane >> 4)) & 7): two variants by piece parity
        uint32_t loff[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) loff[e] = (uint32_t)(((size_t)(lane / 8) * K + (((lane % 8) ^ ((4 * e + (lane >> 4)) & 7)) * 8)) * 2);
        auto issue_tile = [&](int buf, int k0) __attribute__((always_inline)) {
            unsigned char* dst = smem + buf * STAGE + (isb ? 256 * ROWB : 0) + pbase * 1024;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const unsigned char* base = reinterpret_cast<const unsigned char*>(src + (size_t)(8 * (pbase + p)) * K + k0);
                asm volatile("" : "+s"(base));
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + loff[p & 1]),
                                                 (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
            }
        };
        issue_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nkt; ++t) {
            if (t + 1 < nkt) issue_tile((t + 1) & 1, (t + 1) * KD);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ---- MFMA wave: rows wm * 128 + 32 i (i < 4), columns wn * 64 + 32 j (j < 2)
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 31, fh = lane >> 5, fs = (frow >> 1) & 7;
    const int fa_off = (wm * 128 + frow) * ROWB;
    const int fb_off = (256 + wn * 64 + frow) * ROWB;
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nkt; ++t) {
        const unsigned char* sp = smem + (t & 1) * STAGE;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int ch = ((ks * 2 + fh) ^ fs) * 16;
            bf16x8 fa[4], fb[2];
#pragma unroll
            for (int f = 0; f < 4; ++f) fa[f] = *reinterpret_cast<const bf16x8*>(sp + fa_off + f * 32 * ROWB + ch);
#pragma unroll
            for (int f = 0; f < 2; ++f) fb[f] = *reinterpret_cast<const bf16x8*>(sp + fb_off + f * 32 * ROWB + ch);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (store & 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn * 64 + j * 32 + (lane & 31);
                const float bb = bias[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    C[(size_t)row * N + col] = (bf16_t)(pack_bf16x2(acc[i][j][r] + bb, 0.0f) & 0xffffu);
                }
            }
    }
}

static int launch_ws(const bf16_t* A, const bf16_t* B, const float* bias, bf16_t* C, int M, int N, int K, int store) {
    const int nstrips = N / 256, nwg = (M / 256) * nstrips;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ws_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE);
    hipLaunchKernelGGL((ws_kernel<0>), dim3(nwg), dim3(768), NST * STAGE, nullptr, A, B, bias, C, M, N, K, nstrips, nwg, store);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

template <typename F>
static float time_us(F&& f, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms * 1e3f / iters;
}

}  // namespace

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32768, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 8192;
    const int rounds = argc > 4 ? atoi(argv[4]) : 3;
    if (M % 256 || N % 256 || K % 64) { printf("M, N multiples of 256, K of 64\n"); return 1; }
    bf16_t *A, *B, *C, *Cref;
    float* bias;
    hipMalloc(&A, (size_t)M * K * 2);
    hipMalloc(&B, (size_t)N * K * 2);
    hipMalloc(&C, (size_t)M * N * 2);
    hipMalloc(&Cref, (size_t)M * N * 2);
    hipMalloc(&bias, N * 4);
    std::vector<bf16_t> h((size_t)M * K), hb((size_t)N * K);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    auto tobf = [](float f) { unsigned u; memcpy(&u, &f, 4); return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16); };
    for (auto& v : h) v = tobf(((int)(rnd() >> 8 & 0xffff) - 32768) / 32768.0f);
    for (auto& v : hb) v = tobf(((int)(rnd() >> 8 & 0xffff) - 32768) / 32768.0f * 0.05f);
    std::vector<float> hbias(N);
    for (auto& v : hbias) v = ((int)(rnd() >> 8 & 0xffff) - 32768) / 32768.0f;
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(bias, hbias.data(), N * 4, hipMemcpyHostToDevice);
    const double gf = 2.0 * M * N * K * 1e-6;
    printf("M=%d N=%d K=%d  (%.1f GFLOP)\n", M, N, K, gf * 1e-3);
    if (spv_gemm_nt(A, B, bias, Cref, M, N, K, K, K, N, SPV_BF16, SPV_BF16, 0, 1, nullptr, nullptr)) {
        printf("reference launch failed: %s\n", spv_last_error());
        return 1;
    }
    hipDeviceSynchronize();
    std::vector<bf16_t> href((size_t)M * N), hc((size_t)M * N);
    hipMemcpy(href.data(), Cref, href.size() * 2, hipMemcpyDeviceToHost);
    hipMemset(C, 0xff, (size_t)M * N * 2);
    if (launch_ws(A, B, bias, C, M, N, K, 1)) { printf("ws launch failed\n"); return 1; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("ws kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < hc.size(); ++i)
        if (hc[i] != href[i]) { if (!bad) first = i; ++bad; }
    printf("  ws (8 MFMA + 4 loader waves): mismatching elements vs shipped kernel: %zu", bad);
    if (bad) printf("  (first at row %zu col %zu)", first / N, first % N);
    printf("\n");
    fflush(stdout);
    for (int r = 0; r < rounds; ++r) {
        const float t0 = time_us([&] { spv_gemm_nt(A, B, bias, Cref, M, N, K, K, K, N, SPV_BF16, SPV_BF16, 0, 1, nullptr, nullptr); }, 20);
        const float a = time_us([&] { launch_ws(A, B, bias, C, M, N, K, 0); }, 20);
        printf("round %d: shipped (with store) %7.2f us (%.0f TF) | ws K loop only %7.2f us (%.0f TF)\n", r, t0, gf / t0, a, gf / a);
        fflush(stdout);
    }
    return 0;
}
