// Host-side filter design for the reference-parity mode.
//
// The reference designs its filters by calling scipy (an un-vendored dependency):
//   processor.py:254  signal.decimate(x, q)  -> cheby1(8, 0.05, 0.8/q, output='sos'), sosfilt_zi
//   processor.py:78   signal.butter(4, cutoff, btype='low'); filtfilt -> lfilter_zi
// This file computes the same coefficients with the same published procedure (analog
// prototype -> lp2lp -> bilinear -> zpk2sos / zpk2tf) so that the device filters are the
// reference's filters to within an ulp or two; tests/test_design.py checks them against tables
// dumped from scipy (tests/golden/design.npz).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace tdm {

struct cplx {
    double re, im;
};
static inline cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
static inline cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
static inline cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
// Smith's algorithm, the form numpy's complex true_divide uses
static inline cplx cdiv(cplx a, cplx b)
{
    if (std::fabs(b.re) >= std::fabs(b.im)) {
        double rat = b.im / b.re, scl = 1.0 / (b.re + b.im * rat);
        return {(a.re + a.im * rat) * scl, (a.im - a.re * rat) * scl};
    }
    double rat = b.re / b.im, scl = 1.0 / (b.im + b.re * rat);
    return {(a.re * rat + a.im) * scl, (a.im * rat - a.re) * scl};
}

// prewarp + lp2lp_zpk + bilinear_zpk (fs = 2) of an all-pole analog prototype
static inline void lowpass_digital(std::vector<cplx> &p, double &k, double Wn)
{
    const double fs = 2.0;
    const double warped = 2 * fs * std::tan(M_PI * Wn / fs);
    const int deg = (int)p.size();
    for (auto &pp : p) pp = {warped * pp.re, warped * pp.im};
    k = k * std::pow(warped, (double)deg);
    const double fs2 = 2.0 * fs;
    cplx prod = {1.0, 0.0};
    for (auto &pp : p) {
        cplx den = {fs2 - pp.re, -pp.im};
        prod = cmul(prod, den);
        pp = cdiv({fs2 + pp.re, pp.im}, den);
    }
    cplx inv = cdiv({1.0, 0.0}, prod);
    k = k * inv.re;
}

struct Sos4 {
    double sos[4][6];
    double zi[4][2];
};

// lfilter_zi for order-K tf (a[0] == 1): solve (I - companion(a)^T) zi = b[1:] - a[1:] b[0]
// by Gaussian elimination with partial pivoting (what LAPACK gesv does).
static inline void lfilter_zi(const double *b, const double *a, int K, double *zi_out)
{
    long double M[8][9], zi[8];
    for (int i = 0; i < K; ++i) {
        for (int j = 0; j < K; ++j) {
            // companion(a)[0][j] = -a[j+1]; companion[i][i-1] = 1;  use its transpose
            long double compT = (j == 0 ? -(long double)a[i + 1] : 0.0L) + ((i + 1 == j) ? 1.0L : 0.0L);
            M[i][j] = (i == j ? 1.0L : 0.0L) - compT;
        }
        M[i][K] = (long double)b[i + 1] - (long double)a[i + 1] * (long double)b[0];
    }
    for (int c = 0; c < K; ++c) {
        int piv = c;
        for (int r = c + 1; r < K; ++r)
            if (std::fabs((double)M[r][c]) > std::fabs((double)M[piv][c])) piv = r;
        if (piv != c)
            for (int j = 0; j <= K; ++j) std::swap(M[c][j], M[piv][j]);
        for (int r = c + 1; r < K; ++r) {
            long double f = M[r][c] / M[c][c];
            for (int j = c; j <= K; ++j) M[r][j] -= f * M[c][j];
        }
    }
    for (int r = K - 1; r >= 0; --r) {
        long double s = M[r][K];
        for (int j = r + 1; j < K; ++j) s -= M[r][j] * zi[j];
        zi[r] = s / M[r][r];
    }
    for (int r = 0; r < K; ++r) zi_out[r] = (double)zi[r];
}

// cheby1(8, rp, Wn, output='sos') + sosfilt_zi
static inline Sos4 design_cheby1_8(double rp, double Wn)
{
    const int N = 8;
    double eps = std::sqrt(std::pow(10.0, 0.1 * rp) - 1.0);
    double mu = 1.0 / N * std::asinh(1 / eps);
    std::vector<cplx> p;
    for (int m = -N + 1; m < N; m += 2) {
        double theta = M_PI * m / (2 * N);
        // p = -sinh(mu + j theta)
        p.push_back({-(std::sinh(mu) * std::cos(theta)), -(std::cosh(mu) * std::sin(theta))});
    }
    cplx kp = {-p[0].re, -p[0].im};
    for (int i = 1; i < N; ++i) kp = cmul(kp, {-p[i].re, -p[i].im});
    double k = kp.re / std::sqrt(1 + eps * eps);
    lowpass_digital(p, k, Wn);
    // _cplxreal: sort by (real, |imag|), average each pole with its conjugate partner
    std::vector<cplx> pos, neg;
    std::vector<cplx> sorted = p;
    std::stable_sort(sorted.begin(), sorted.end(), [](cplx a, cplx b) {
        if (a.re != b.re) return a.re < b.re;
        return std::fabs(a.im) < std::fabs(b.im);
    });
    for (auto &pp : sorted) (pp.im > 0 ? pos : neg).push_back(pp);
    std::vector<cplx> pc;
    for (size_t i = 0; i < pos.size(); ++i)
        pc.push_back({(pos[i].re + neg[i].re) / 2, (pos[i].im - neg[i].im) / 2});
    Sos4 out;
    for (int si = 3; si >= 0; --si) {
        int idx = 0;
        double best = 1e300;
        for (size_t i = 0; i < pc.size(); ++i) {
            double d = std::fabs(1 - std::hypot(pc[i].re, pc[i].im));
            if (d < best) { best = d; idx = (int)i; }
        }
        cplx p1 = pc[idx];
        pc.erase(pc.begin() + idx);
        out.sos[si][0] = 1.0; out.sos[si][1] = 2.0; out.sos[si][2] = 1.0;
        out.sos[si][3] = 1.0;
        out.sos[si][4] = -(p1.re + p1.re);
        out.sos[si][5] = p1.re * p1.re + p1.im * p1.im;
    }
    for (int j = 0; j < 3; ++j) out.sos[0][j] *= k;
    double scale = 1.0;
    for (int s = 0; s < 4; ++s) {
        double zi[2];
        lfilter_zi(&out.sos[s][0], &out.sos[s][3], 2, zi);
        out.zi[s][0] = scale * zi[0];
        out.zi[s][1] = scale * zi[1];
        double bs = (out.sos[s][0] + out.sos[s][1]) + out.sos[s][2];
        double as = (out.sos[s][3] + out.sos[s][4]) + out.sos[s][5];
        scale *= bs / as;
    }
    return out;
}

struct Tf4 {
    double b[5], a[5], zi[4];  // transfer-function form, as scipy.signal.butter returns it
    // The same filter as two cascaded biquads (what the device runs): the order-4 companion form
    // has ~1e4 transient growth for narrow cutoffs, which a blocked evaluation would square;
    // biquad coordinates keep the carried states well conditioned.
    double sos[2][6];
    double soszi[2][2];
};

// np.poly of a root list (sequential convolution with [1, -r])
static inline std::vector<cplx> poly(const std::vector<cplx> &roots)
{
    std::vector<cplx> c = {{1.0, 0.0}};
    for (auto r : roots) {
        std::vector<cplx> n(c.size() + 1, {0.0, 0.0});
        for (size_t i = 0; i < c.size(); ++i) {
            n[i] = cadd(n[i], c[i]);
            n[i + 1] = csub(n[i + 1], cmul(c[i], r));
        }
        c = n;
    }
    return c;
}

// butter(4, Wn, 'low') -> b, a ; lfilter_zi
static inline Tf4 design_butter4(double Wn)
{
    const int N = 4;
    std::vector<cplx> p;
    for (int m = -N + 1; m < N; m += 2) {
        double th = M_PI * m / (2 * N);
        p.push_back({-std::cos(th), -std::sin(th)});
    }
    double k = 1.0;
    lowpass_digital(p, k, Wn);
    std::vector<cplx> z(N, cplx{-1.0, 0.0});
    auto bz = poly(z);
    auto ap = poly(p);
    Tf4 out;
    for (int i = 0; i <= N; ++i) {
        out.b[i] = k * bz[i].re;
        out.a[i] = ap[i].re;
    }
    lfilter_zi(out.b, out.a, N, out.zi);
    // biquad form: conjugate pairs, the pair closest to the unit circle last (zpk2sos order)
    std::vector<cplx> pos;
    for (auto &pp : p)
        if (pp.im > 0) pos.push_back(pp);
    std::sort(pos.begin(), pos.end(), [](cplx x, cplx y) {
        return std::fabs(1 - std::hypot(x.re, x.im)) > std::fabs(1 - std::hypot(y.re, y.im));
    });
    double scale = 1.0;
    for (int si = 0; si < 2; ++si) {
        const cplx p1 = pos[si];
        const double g = si == 0 ? k : 1.0;
        out.sos[si][0] = g; out.sos[si][1] = 2 * g; out.sos[si][2] = g;
        out.sos[si][3] = 1.0;
        out.sos[si][4] = -(p1.re + p1.re);
        out.sos[si][5] = p1.re * p1.re + p1.im * p1.im;
        double zi2[2];
        lfilter_zi(&out.sos[si][0], &out.sos[si][3], 2, zi2);
        out.soszi[si][0] = scale * zi2[0];
        out.soszi[si][1] = scale * zi2[1];
        const double bs = (out.sos[si][0] + out.sos[si][1]) + out.sos[si][2];
        const double as = (out.sos[si][3] + out.sos[si][4]) + out.sos[si][5];
        scale *= bs / as;
    }
    return out;
}

// filter_signal's cutoff (processor.py:69-75)
static inline double butter_cutoff(double bandwidth, double fs)
{
    double nyquist = fs / 2;
    double cutoff = (bandwidth / 2) / nyquist;
    return std::min(0.99, std::max(0.01, cutoff));
}

// process()'s decimation decision (processor.py:245-255)
static inline int decimation_factor(double sample_rate)
{
    const double target = 240000;
    if (sample_rate > target * 2) {
        int q = (int)(sample_rate / target);
        if (q > 1) return q;
    }
    return 1;
}

}  // namespace tdm
