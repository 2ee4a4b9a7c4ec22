// Host emulation of the FNet mixer kernel's phases (spv_fft.hip) using the SAME index math and butterflies
// (csrc/spv_fft_core.h), checked against a naive double-precision Re(F_N x F_D).  Run by tests/test_host_harness.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../vit-spectre-experiments_amd/csrc/spv_fft_core.h"

static int run_case(int N, int D) {
    std::vector<double> x((size_t)N * D);
    unsigned s = 12345u + N * 131 + D;
    for (auto& v : x) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0 - 1.0; }
    // naive reference
    std::vector<double> ref((size_t)N * D, 0.0);
    {
        std::vector<double> A((size_t)N * D), B((size_t)N * D);
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < D; ++k) {
                double a = 0, b = 0;
                for (int d = 0; d < D; ++d) {
                    double ang = 2.0 * M_PI * ((long long)k * d % D) / D;
                    a += x[(size_t)n * D + d] * cos(ang);
                    b += x[(size_t)n * D + d] * sin(ang);
                }
                A[(size_t)n * D + k] = a; B[(size_t)n * D + k] = b;
            }
        for (int m = 0; m < N; ++m)
            for (int k = 0; k < D; ++k) {
                double y = 0;
                for (int n = 0; n < N; ++n) {
                    double ang = 2.0 * M_PI * ((long long)m * n % N) / N;
                    y += cos(ang) * A[(size_t)n * D + k] - sin(ang) * B[(size_t)n * D + k];
                }
                ref[(size_t)m * D + k] = y;
            }
    }
    // ---- emulated kernel
    std::vector<float> lds((size_t)(N + 1) * D, 0.0f), twr(D), twi(D);
    for (int i = 0; i < N * D; ++i) lds[i] = (float)x[i];
    for (int t = 0; t < D; ++t) { twr[t] = (float)cos(2.0 * M_PI * t / D); twi[t] = (float)-sin(2.0 * M_PI * t / D); }
    FftPlan plan = make_fft_plan(D);
    const int NF = (N + 1) / 2;
    for (int f = 0; f < NF; ++f) {
        float* re = &lds[(size_t)(2 * f) * D];
        float* im = &lds[(size_t)(2 * f + 1) * D];
        for (int p = 0; p < plan.npass; ++p) {
            const int R = plan.radix[p], Ns = plan.ns[p], nb = D / R;
            std::vector<cpx> regs((size_t)nb * 8);
            for (int j = 0; j < nb; ++j) {  // read phase (all threads), then barrier
                if (R == 8) stockham_load<8>(re, im, twr.data(), twi.data(), D, Ns, j, &regs[(size_t)j * 8]);
                else if (R == 4) stockham_load<4>(re, im, twr.data(), twi.data(), D, Ns, j, &regs[(size_t)j * 8]);
                else stockham_load<2>(re, im, twr.data(), twi.data(), D, Ns, j, &regs[(size_t)j * 8]);
            }
            for (int j = 0; j < nb; ++j) {  // write phase
                if (R == 8) stockham_store<8>(re, im, Ns, j, &regs[(size_t)j * 8]);
                else if (R == 4) stockham_store<4>(re, im, Ns, j, &regs[(size_t)j * 8]);
                else stockham_store<2>(re, im, Ns, j, &regs[(size_t)j * 8]);
            }
        }
    }
    const int nh1 = N / 2 + 1;
    std::vector<float> tw((size_t)(N + 1) * 2 * FNET_TWS, 0.0f);
    for (int n = 0; n < N; ++n)
        for (int m = 0; m < nh1; ++m) {
            double ang = 2.0 * M_PI * ((long long)m * n % N) / N;
            tw[((size_t)n * 2 + 0) * FNET_TWS + m] = (float)cos(ang);
            tw[((size_t)n * 2 + 1) * FNET_TWS + m] = (float)sin(ang);
        }
    std::vector<float> y((size_t)N * D, 1e30f);
    for (int k = 0; k <= D / 2; ++k) {
        const int mk = (D - k) % D;
        std::vector<float> P(nh1, 0.0f), Q(nh1, 0.0f);
        for (int f = 0; f < NF; ++f) {
            const float* zr = &lds[(size_t)(2 * f) * D];
            const float* zi = &lds[(size_t)(2 * f + 1) * D];
            float a1, b1, a2, b2;
            unpack_pair(zr[k], zr[mk], zi[k], zi[mk], a1, b1, a2, b2);
            const float* t1 = &tw[(size_t)(2 * f) * 2 * FNET_TWS];
            const float* t2 = &tw[(size_t)(2 * f + 1) * 2 * FNET_TWS];
            for (int m = 0; m < nh1; ++m) {
                P[m] += t1[m] * a1 + t2[m] * a2;
                Q[m] += t1[FNET_TWS + m] * b1 + t2[FNET_TWS + m] * b2;
            }
        }
        for (int m = 0; m < nh1; ++m) {
            const int mm = (N - m) % N;
            y[(size_t)m * D + k] = P[m] - Q[m];
            y[(size_t)mm * D + mk] = P[m] - Q[m];
            if (mm != m) y[(size_t)mm * D + k] = P[m] + Q[m];
            if (mk != k) y[(size_t)m * D + mk] = P[m] + Q[m];
            if (mm != m && mk != k) y[(size_t)mm * D + mk] = P[m] - Q[m];
        }
    }
    double maxerr = 0, maxref = 0;
    for (size_t i = 0; i < y.size(); ++i) {
        maxerr = fmax(maxerr, fabs(y[i] - ref[i]));
        maxref = fmax(maxref, fabs(ref[i]));
    }
    printf("N=%d D=%d max|err|=%.3e max|ref|=%.3e\n", N, D, maxerr, maxref);
    return maxerr <= 2e-5 * maxref + 1e-5 ? 0 : 1;
}

int main() {
    int bad = 0;
    const int cases[][2] = {{65, 512}, {5, 16}, {50, 64}, {6, 8}, {17, 64}, {65, 128}, {7, 32}, {9, 256}, {4, 1024}, {1, 8}, {2, 16}};
    for (auto& c : cases) bad += run_case(c[0], c[1]);
    printf(bad ? "FAIL\n" : "OK\n");
    return bad;
}
