// Diagnostic: per-phase s_memtime stamps of the FNet v2 kernel (development tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSPV_FNET_STAMPS tools/fnet_stamps.hip vit-spectre-experiments_amd/csrc/spv_misc.hip -o gpurun_out/fnet_stamps
#include "../vit-spectre-experiments_amd/csrc/spv_fft.hip"

#include <algorithm>
#include <vector>

static void report(const char* title, const std::vector<unsigned long long>& h, int B) {
    const char* names[] = {"A load", "B fft(mfma)", "B->C wait", "C mfma", "C valu row", "wait", "stage write", "wait", "D out"};
    printf("== %s\n", title);
    for (int wave : {0, 3, 7}) {
        printf("wave %d: median shader cycles per phase over %d workgroups\n", wave, B);
        for (int p = 0; p < 9; ++p) {
            std::vector<long long> v;
            for (int b = 0; b < B; ++b) v.push_back((long long)(h[((size_t)b * 8 + wave) * 16 + p + 1] - h[((size_t)b * 8 + wave) * 16 + p]));
            std::sort(v.begin(), v.end());
            printf("   %-12s %8lld   (p10 %lld, p90 %lld)\n", names[p], v[v.size() / 2], v[v.size() / 10], v[v.size() * 9 / 10]);
        }
        std::vector<long long> v;
        for (int b = 0; b < B; ++b) v.push_back((long long)(h[((size_t)b * 8 + wave) * 16 + 9] - h[((size_t)b * 8 + wave) * 16 + 0]));
        std::sort(v.begin(), v.end());
        printf("   %-12s %8lld\n", "total", v[v.size() / 2]);
    }
    // timeline in 100 MHz real-time ticks (slot 10 = start, 11 = end), relative to the first start
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < B; ++b) t0 = std::min(t0, h[((size_t)b * 8) * 16 + 10]);
    std::vector<long long> s1, e1, s2, e2;
    for (int b = 0; b < B; ++b) {
        auto& s = b < B / 2 ? s1 : s2;
        auto& e = b < B / 2 ? e1 : e2;
        s.push_back((long long)(h[((size_t)b * 8) * 16 + 10] - t0));
        e.push_back((long long)(h[((size_t)b * 8) * 16 + 11] - t0));
    }
    for (auto* v : {&s1, &e1, &s2, &e2}) std::sort(v->begin(), v->end());
    auto med = [](std::vector<long long>& v) { return v.empty() ? 0ll : v[v.size() / 2]; };
    printf("timeline (x 10 ns): first half start med %lld end med %lld max %lld | second half start med %lld end med %lld max %lld\n", med(s1),
           med(e1), e1.empty() ? 0ll : e1.back(), med(s2), med(e2), e2.empty() ? 0ll : e2.back());
}

int main() {
    const int B = 512, N = 65, D = 512;
    bf16_t *x, *y, *m;
    float *tw, *gam, *bet, *mean, *rstd;
    unsigned long long* st;
    hipMalloc(&x, (size_t)B * N * D * 2);
    hipMalloc(&y, (size_t)B * N * D * 2);
    hipMalloc(&m, (size_t)B * N * D * 2);
    hipMemset(x, 0x3c, (size_t)B * N * D * 2);
    hipMalloc(&gam, D * 4); hipMalloc(&bet, D * 4); hipMalloc(&mean, B * N * 4); hipMalloc(&rstd, B * N * 4);
    hipMemset(gam, 0, D * 4); hipMemset(bet, 0, D * 4);
    const int64_t nt = spv_fnet_twiddle_floats(N);
    hipMalloc(&tw, nt * 4);
    hipMalloc(&st, (size_t)B * 8 * 16 * 8);
    spv_fnet_make_twiddle(tw, N, nullptr);
    hipMemcpyToSymbol(HIP_SYMBOL(g_fnet_stamps), &st, sizeof(st));
    std::vector<unsigned long long> h((size_t)B * 8 * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        float ms = 0;
        for (int it = 0; it < 4; ++it) {
            hipMemset(st, 0, (size_t)B * 8 * 16 * 8);
            hipEventRecord(e0, nullptr);
            if (mode == 0) spv_fnet_mix(x, y, nullptr, tw, B, N, D, SPV_BF16, nullptr, nullptr);
            else spv_fnet_ln_fwd(x, m, y, gam, bet, mean, rstd, tw, B, N, D, SPV_BF16, nullptr);
            hipEventRecord(e1, nullptr);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
        char title[128];
        snprintf(title, sizeof title, "%s  (stamped launch: %.1f us)", mode == 0 ? "spv_fnet_mix" : "spv_fnet_ln_fwd", ms * 1e3);
        report(title, h, B);
    }
    return 0;
}
