// spv_fft_core.h -- index math and butterflies of the FNet mixer kernel, written so that the very same
// functions compile for the device (spv_fft.hip) and for the host (tests/cpu_harness/fft_core_test.cpp,
// which emulates the workgroup thread by thread and checks against a naive DFT).
//
// y = Re(F_N x F_D) for one (N x D) real sample x:
//   phase B: rows are paired (2f, 2f+1) into z = x[2f] + i x[2f+1]; one in-place complex Stockham FFT of
//            length D per pair, radix-8 passes (+ one radix-2/4 pass when log2 D is not a multiple of 3),
//            D/8 threads per FFT, all of an FFT's data held in registers between the read and the write
//            of a pass (so in-place is safe with one barrier in between).
//   phase C: per frequency k <= D/2 the two real spectra are unpacked from Z[k], Z[D-k] and the N-point
//            DFT along tokens is done as P[m] = sum_n cos(2 pi m n/N) A[n,k], Q[m] = sum_n sin(..) B[n,k]
//            for m <= N/2 only; the four symmetric outputs are
//            y[m,k] = y[N-m,D-k] = P - Q,  y[N-m,k] = y[m,D-k] = P + Q.
#pragma once

#if defined(__HIPCC__)
#define SPV_HD __host__ __device__ __forceinline__
#else
#define SPV_HD inline
#endif

struct cpx {
    float re, im;
};
SPV_HD cpx cmk(float r, float i) { cpx c; c.re = r; c.im = i; return c; }
SPV_HD cpx cadd(cpx a, cpx b) { return cmk(a.re + b.re, a.im + b.im); }
SPV_HD cpx csub(cpx a, cpx b) { return cmk(a.re - b.re, a.im - b.im); }
SPV_HD cpx cmul(cpx a, cpx b) { return cmk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
SPV_HD cpx cmul_mi(cpx a) { return cmk(a.im, -a.re); }  // a * (-i)

// forward DFTs (kernel e^{-2 pi i n k / R}), in place, natural order in and out
SPV_HD void dft2(cpx* v) {
    cpx a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
SPV_HD void dft4(cpx* v) {
    cpx s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
    cpx s13 = cadd(v[1], v[3]), d13 = cmul_mi(csub(v[1], v[3]));  // (v1 - v3) * (-i)
    v[0] = cadd(s02, s13);
    v[1] = cadd(d02, d13);
    v[2] = csub(s02, s13);
    v[3] = csub(d02, d13);
}
SPV_HD void dft8(cpx* v) {
    const float h = 0.70710678118654752440f;
    cpx a[4], b[4];
    // X[2q] = DFT4(v[n] + v[n+4]),  X[2q+1] = DFT4((v[n] - v[n+4]) w8^n)
    for (int n = 0; n < 4; ++n) { a[n] = cadd(v[n], v[n + 4]); b[n] = csub(v[n], v[n + 4]); }
    b[1] = cmk((b[1].re + b[1].im) * h, (b[1].im - b[1].re) * h);   // * (1 - i)/sqrt2
    b[2] = cmul_mi(b[2]);                                           // * (-i)
    b[3] = cmk((b[3].im - b[3].re) * h, -(b[3].re + b[3].im) * h);  // * (-1 - i)/sqrt2
    dft4(a);
    dft4(b);
    for (int q = 0; q < 4; ++q) { v[2 * q] = a[q]; v[2 * q + 1] = b[q]; }
}

// The same radix-8 butterfly on a generic element type T (float, or a 2-wide vector that lowers to packed
// v_pk_*_f32 math: two independent butterflies per instruction).
template <typename T> struct cpx_t { T re, im; };
template <typename T> SPV_HD cpx_t<T> tadd(cpx_t<T> a, cpx_t<T> b) { cpx_t<T> c; c.re = a.re + b.re; c.im = a.im + b.im; return c; }
template <typename T> SPV_HD cpx_t<T> tsub(cpx_t<T> a, cpx_t<T> b) { cpx_t<T> c; c.re = a.re - b.re; c.im = a.im - b.im; return c; }
template <typename T> SPV_HD cpx_t<T> tmul(cpx_t<T> a, cpx_t<T> b) {
    cpx_t<T> c; c.re = a.re * b.re - a.im * b.im; c.im = a.re * b.im + a.im * b.re; return c;
}
template <typename T> SPV_HD cpx_t<T> tmul_mi(cpx_t<T> a) { cpx_t<T> c; c.re = a.im; c.im = -a.re; return c; }
template <typename T> SPV_HD void dft4_t(cpx_t<T>* v) {
    cpx_t<T> s02 = tadd(v[0], v[2]), d02 = tsub(v[0], v[2]);
    cpx_t<T> s13 = tadd(v[1], v[3]), d13 = tmul_mi(tsub(v[1], v[3]));
    v[0] = tadd(s02, s13);
    v[1] = tadd(d02, d13);
    v[2] = tsub(s02, s13);
    v[3] = tsub(d02, d13);
}
template <typename T> SPV_HD void dft8_t(cpx_t<T>* v) {
    const float h = 0.70710678118654752440f;
    cpx_t<T> a[4], b[4];
    for (int n = 0; n < 4; ++n) { a[n] = tadd(v[n], v[n + 4]); b[n] = tsub(v[n], v[n + 4]); }
    { cpx_t<T> t = b[1]; b[1].re = (t.re + t.im) * h; b[1].im = (t.im - t.re) * h; }   // * (1 - i)/sqrt2
    b[2] = tmul_mi(b[2]);                                                            // * (-i)
    { cpx_t<T> t = b[3]; b[3].re = (t.im - t.re) * h; b[3].im = -(t.re + t.im) * h; }  // * (-1 - i)/sqrt2
    dft4_t(a);
    dft4_t(b);
    for (int q = 0; q < 4; ++q) { v[2 * q] = a[q]; v[2 * q + 1] = b[q]; }
}

// One Stockham butterfly of radix R (2, 4 or 8), sub-transform length Ns, on the planar (re, im) rows of
// length D.  j in [0, D/R).  tw_re/tw_im: table of exp(-2 pi i t / D), t in [0, D).
template <int R>
SPV_HD void stockham_load(const float* re, const float* im, const float* tw_re, const float* tw_im, int D, int Ns, int j,
                          cpx* v) {
    const int stride = D / R;
    const int tstep = (j % Ns) * (D / (Ns * R));
    for (int r = 0; r < R; ++r) {
        cpx x = cmk(re[j + r * stride], im[j + r * stride]);
        const int t = r * tstep;
        v[r] = (t == 0) ? x : cmul(x, cmk(tw_re[t], tw_im[t]));
    }
    if (R == 2) dft2(v);
    else if (R == 4) dft4(v);
    else dft8(v);
}
template <int R>
SPV_HD void stockham_store(float* re, float* im, int Ns, int j, const cpx* v) {
    const int j0 = (j / Ns) * Ns * R + (j % Ns);
    for (int r = 0; r < R; ++r) {
        re[j0 + r * Ns] = v[r].re;
        im[j0 + r * Ns] = v[r].im;
    }
}

// pass plan for a power-of-two D >= 8: radix-8 passes, then one radix-4 or radix-2 pass if needed
struct FftPlan {
    int npass;
    int radix[8];
    int ns[8];
};
inline FftPlan make_fft_plan(int D) {
    FftPlan p;
    p.npass = 0;
    int ns = 1, rem = D;
    while (rem >= 8) { p.radix[p.npass] = 8; p.ns[p.npass] = ns; ns *= 8; rem /= 8; ++p.npass; }
    if (rem > 1) { p.radix[p.npass] = rem; p.ns[p.npass] = ns; ++p.npass; }
    return p;
}

// unpack the two real rows' spectra at frequency k from Z = FFT(x[n1] + i x[n2])
//   X = A - i B (A = sum x cos, B = sum x sin)
SPV_HD void unpack_pair(float zr_k, float zr_mk, float zi_k, float zi_mk, float& a1, float& b1, float& a2, float& b2) {
    a1 = 0.5f * (zr_k + zr_mk);
    b1 = -0.5f * (zi_k - zi_mk);
    a2 = 0.5f * (zi_k + zi_mk);
    b2 = 0.5f * (zr_k - zr_mk);
}

// v2 (MFMA) formulation: with Z the planar FFT rows (row 2f = Re, row 2f+1 = Im of FFT(x[2f] + i x[2f+1])),
//   y[m,k] = sum_j W1[m,j] Z[j][k] + W2[m,j] Z[j][D-k]; for the pair (n1, n2) = (2f, 2f+1) with c = cos, s = sin of
//   2 pi m n / N (zero for n >= N):
SPV_HD void fnet_v2_weights(float c1, float s1, float c2, float s2, float& w1_re, float& w1_im, float& w2_re, float& w2_im) {
    w1_re = 0.5f * (c1 - s2);  // multiplies Zr[k]
    w1_im = 0.5f * (c2 + s1);  // multiplies Zi[k]
    w2_re = 0.5f * (c1 + s2);  // multiplies Zr[D-k]
    w2_im = 0.5f * (c2 - s1);  // multiplies Zi[D-k]
}

// stage-2 twiddle table layout: tw[n][2][TWS], n in [0, N] (row N all zero: the pad row of an odd N),
//   tw[n][0][m] = cos(2 pi (m n mod N) / N), tw[n][1][m] = sin(...), m in [0, N/2], zero beyond.
constexpr int FNET_MH = 20;           // per-thread m range (two halves cover m <= 39  =>  N <= 79)
constexpr int FNET_TWS = 2 * FNET_MH;  // table row stride
