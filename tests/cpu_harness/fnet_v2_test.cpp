// Host check of the FNet v2 formulation used by fnet_mfma_kernel (spv_fft.hip):
//   Z = planar rows (Zr_f, Zi_f) of FFT_D(x[2f] + i x[2f+1]);   Y1 = W1 . Z,  Y2 = W2 . Z  (plain matrix products)
//   y[m, k] = Y1[m, k] + Y2[m, (D-k)%D]   for m <= N/2,     y[N-m, k] = Y1[m, (D-k)%D] + Y2[m, k]
// with bf16 rounding emulated wherever the kernel stores bf16 (input, LDS between FFT passes, Z, W tables, Y stage, output).
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../vit-spectre-experiments_amd/csrc/spv_fft_core.h"

static float bf(float x) {  // round to nearest even bf16
    uint32_t u; memcpy(&u, &x, 4);
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    float y; memcpy(&y, &u, 4); return y;
}

static int run_case(int N, int D, bool round) {
    auto R = [&](float v) { return round ? bf(v) : v; };
    std::vector<float> x((size_t)N * D);
    unsigned s = 777u + N * 31 + D;
    for (auto& v : x) { s = s * 1664525u + 1013904223u; v = R(((s >> 8) & 0xffff) / 32768.0f - 1.0f); }
    // reference (double)
    std::vector<double> ref((size_t)N * D);
    {
        std::vector<double> A((size_t)N * D), B((size_t)N * D);
        for (int n = 0; n < N; ++n) for (int k = 0; k < D; ++k) {
            double a = 0, b = 0;
            for (int d = 0; d < D; ++d) { double ang = 2 * M_PI * ((long long)k * d % D) / D; a += x[(size_t)n * D + d] * cos(ang); b += x[(size_t)n * D + d] * sin(ang); }
            A[(size_t)n * D + k] = a; B[(size_t)n * D + k] = b;
        }
        for (int m = 0; m < N; ++m) for (int k = 0; k < D; ++k) {
            double y = 0;
            for (int n = 0; n < N; ++n) { double ang = 2 * M_PI * ((long long)m * n % N) / N; y += cos(ang) * A[(size_t)n * D + k] - sin(ang) * B[(size_t)n * D + k]; }
            ref[(size_t)m * D + k] = y;
        }
    }
    const int NF = (N + 1) / 2, J = 2 * NF, nh1 = N / 2 + 1;
    std::vector<float> Z((size_t)J * D, 0.0f), twr(D), twi(D);
    for (int i = 0; i < N * D; ++i) Z[i] = x[i];
    for (int t = 0; t < D; ++t) { twr[t] = (float)cos(2 * M_PI * t / D); twi[t] = (float)-sin(2 * M_PI * t / D); }
    FftPlan plan = make_fft_plan(D);
    for (int f = 0; f < NF; ++f) {
        float* re = &Z[(size_t)(2 * f) * D]; float* im = re + D;
        for (int p = 0; p < plan.npass; ++p) {
            const int Rd = plan.radix[p], Ns = plan.ns[p], nb = D / Rd;
            std::vector<cpx> regs((size_t)nb * 8);
            for (int j = 0; j < nb; ++j) {
                if (Rd == 8) stockham_load<8>(re, im, twr.data(), twi.data(), D, Ns, j, &regs[(size_t)j * 8]);
                else if (Rd == 4) stockham_load<4>(re, im, twr.data(), twi.data(), D, Ns, j, &regs[(size_t)j * 8]);
                else stockham_load<2>(re, im, twr.data(), twi.data(), D, Ns, j, &regs[(size_t)j * 8]);
            }
            for (int j = 0; j < nb; ++j) {
                for (int r = 0; r < Rd; ++r) { regs[(size_t)j * 8 + r].re = R(regs[(size_t)j * 8 + r].re); regs[(size_t)j * 8 + r].im = R(regs[(size_t)j * 8 + r].im); }
                if (Rd == 8) stockham_store<8>(re, im, Ns, j, &regs[(size_t)j * 8]);
                else if (Rd == 4) stockham_store<4>(re, im, Ns, j, &regs[(size_t)j * 8]);
                else stockham_store<2>(re, im, Ns, j, &regs[(size_t)j * 8]);
            }
        }
    }
    // weight tables
    std::vector<float> W1((size_t)nh1 * J), W2((size_t)nh1 * J);
    for (int m = 0; m < nh1; ++m) for (int f = 0; f < NF; ++f) {
        const int n1 = 2 * f, n2 = 2 * f + 1;
        auto C = [&](int n) { return n < N ? cos(2 * M_PI * ((long long)m * n % N) / N) : 0.0; };
        auto S = [&](int n) { return n < N ? sin(2 * M_PI * ((long long)m * n % N) / N) : 0.0; };
        fnet_v2_weights((float)C(n1), (float)S(n1), (float)C(n2), (float)S(n2), W1[(size_t)m * J + n1], W1[(size_t)m * J + n2],
                        W2[(size_t)m * J + n1], W2[(size_t)m * J + n2]);
    }
    std::vector<float> Y1((size_t)nh1 * D), Y2((size_t)nh1 * D);
    for (int m = 0; m < nh1; ++m) for (int k = 0; k < D; ++k) {
        float a = 0, b = 0;
        const bool mf = m < 32;  // MFMA rows use bf16 weights, the VALU rows fp32 weights
        for (int j = 0; j < J; ++j) {
            a += (mf ? R(W1[(size_t)m * J + j]) : W1[(size_t)m * J + j]) * Z[(size_t)j * D + k];
            b += (mf ? R(W2[(size_t)m * J + j]) : W2[(size_t)m * J + j]) * Z[(size_t)j * D + k];
        }
        Y1[(size_t)m * D + k] = R(a); Y2[(size_t)m * D + k] = R(b);
    }
    double maxerr = 0, maxref = 0, sumsq = 0, refsq = 0;
    for (int r = 0; r < N; ++r) for (int k = 0; k < D; ++k) {
        const int mk = (D - k) % D;
        float y;
        if (r <= N / 2) y = Y1[(size_t)r * D + k] + Y2[(size_t)r * D + mk];
        else { const int m = N - r; y = Y1[(size_t)m * D + mk] + Y2[(size_t)m * D + k]; }
        y = R(y);
        const double e = y - ref[(size_t)r * D + k];
        maxerr = fmax(maxerr, fabs(e)); maxref = fmax(maxref, fabs(ref[(size_t)r * D + k]));
        sumsq += e * e; refsq += ref[(size_t)r * D + k] * ref[(size_t)r * D + k];
    }
    printf("N=%d D=%d bf16=%d max|err|/max|ref|=%.3e  rms rel=%.3e\n", N, D, (int)round, maxerr / maxref, sqrt(sumsq / refsq));
    return (round ? maxerr / maxref <= 6e-3 : maxerr / maxref <= 3e-6) ? 0 : 1;
}

int main() {
    int bad = 0;
    const int cases[][2] = {{65, 512}, {65, 64}, {50, 64}, {17, 512}, {79, 64}, {2, 64}, {64, 512}};
    for (auto& c : cases) { bad += run_case(c[0], c[1], false); bad += run_case(c[0], c[1], true); }
    printf(bad ? "FAIL\n" : "OK\n");
    return bad;
}
