// Diagnostic: per-phase s_memtime stamps of the FNet v2 kernel (development tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSPV_FNET_STAMPS tools/fnet_stamps.hip vit-spectre-experiments_amd/csrc/spv_misc.hip -o gpurun_out/fnet_stamps
#include "../vit-spectre-experiments_amd/csrc/spv_fft.hip"

#include <algorithm>
#include <vector>

int main() {
    const int B = 512, N = 65, D = 512;
    bf16_t *x, *y;
    float* tw;
    unsigned long long* st;
    hipMalloc(&x, (size_t)B * N * D * 2);
    hipMalloc(&y, (size_t)B * N * D * 2);
    hipMemset(x, 0x3c, (size_t)B * N * D * 2);
    const int64_t nt = spv_fnet_twiddle_floats(N);
    hipMalloc(&tw, nt * 4);
    hipMalloc(&st, (size_t)B * 8 * 16 * 8);
    hipMemset(st, 0, (size_t)B * 8 * 16 * 8);
    spv_fnet_make_twiddle(tw, N, nullptr);
    hipMemcpyToSymbol(HIP_SYMBOL(g_fnet_stamps), &st, sizeof(st));
    for (int it = 0; it < 3; ++it) spv_fnet_mix(x, y, nullptr, tw, B, N, D, SPV_BF16, nullptr, nullptr);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)B * 8 * 16);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"A load", "B fft", "B->C wait", "C mfma", "C valu row", "wait", "stage write", "wait", "D out"};
    for (int wave : {0, 3, 7}) {
        printf("wave %d: median cycles per phase over %d workgroups (s_memtime ticks @100MHz? see total)\n", wave, B);
        for (int p = 0; p < 9; ++p) {
            std::vector<long long> v;
            for (int b = 0; b < B; ++b) v.push_back((long long)(h[((size_t)b * 8 + wave) * 16 + p + 1] - h[((size_t)b * 8 + wave) * 16 + p]));
            std::sort(v.begin(), v.end());
            printf("   %-12s %8lld\n", names[p], v[v.size() / 2]);
        }
        std::vector<long long> v;
        for (int b = 0; b < B; ++b) v.push_back((long long)(h[((size_t)b * 8 + wave) * 16 + 9] - h[((size_t)b * 8 + wave) * 16 + 0]));
        std::sort(v.begin(), v.end());
        printf("   %-12s %8lld\n", "total", v[v.size() / 2]);
    }
    return 0;
}
