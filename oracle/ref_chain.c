/*
 * ORACLE -- test infrastructure, NOT product code.
 *
 * Plain-C, single-thread restatement of the reference's IQ->symbol chain
 * (syrex1013/TetraEar v2.2, tetraear/signal/processor.py) including the
 * scipy.signal routines it calls (scipy 1.15.3; scipy is an un-vendored
 * dependency of the reference, requirements.txt:2).  Filter COEFFICIENTS are
 * inputs here; they come from oracle/design.py (pinned bit-for-bit against
 * scipy tables in tests/golden/design.npz).
 *
 * Parity status: PINNED -- checked against golden vectors produced by importing
 * the reference itself (tests/golden/make_golden.py -> tests/golden/ npz files); see
 * tests/test_oracle_golden.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Build: make -C oracle   (gcc -O2 -ffp-contract=off: scipy's
 * generic x86-64 build has no FMA contraction, so none here either).
 *
 * Complex data is carried as separate re/im arrays of doubles: every filter
 * coefficient is real (cast to complex with zero imaginary part by the
 * reference), so numpy's complex product (ar*br - ai*bi, ar*bi + ai*br) reduces
 * to the per-component real product used below, bit for bit, for finite data.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ---- scipy.signal.sosfilt (scipy/signal/_sosfilt.pyx, complex path), in place, one component.
 * per sample, per section:  y = b0*x + z0;  z0 = b1*x - a1*y + z1;  z1 = b2*x - a2*y          */
static void sosfilt_real(const double *sos, int nsec, double *x, int64_t n, double *z /*[nsec][2]*/)
{
    for (int64_t i = 0; i < n; ++i) {
        double xc = x[i];
        for (int s = 0; s < nsec; ++s) {
            const double *c = sos + 6 * s;
            double xn = c[0] * xc + z[2 * s];
            z[2 * s] = (c[1] * xc - c[4] * xn) + z[2 * s + 1];
            z[2 * s + 1] = c[2] * xc - c[5] * xn;
            xc = xn;
        }
        x[i] = xc;
    }
}

/* ---- scipy.signal.lfilter (scipy/signal/_lfilter.c.in, CDOUBLE_filt with a[0]==1), direct
 * form II transposed:  y = z0 + b0*x;  z[k-1] = (z[k] + b[k]*x) - a[k]*y;  z[K-1] = b[K]*x - a[K]*y */
static void lfilter_real(const double *b, const double *a, int order, double *x, int64_t n, double *z)
{
    for (int64_t i = 0; i < n; ++i) {
        double xi = x[i];
        double y = z[0] + b[0] * xi;
        for (int k = 1; k < order; ++k)
            z[k - 1] = (z[k] + b[k] * xi) - a[k] * y;
        z[order - 1] = b[order] * xi - a[order] * y;
        x[i] = y;
    }
}

/* scipy.signal._arraytools.odd_ext: [2*x[0]-x[edge:0:-1], x, 2*x[-1]-x[-2:-(edge+2):-1]] */
static void odd_ext(const double *x, int64_t n, int edge, double *ext)
{
    for (int i = 0; i < edge; ++i)
        ext[i] = 2 * x[0] - x[edge - i];
    memcpy(ext + edge, x, (size_t)n * sizeof(double));
    for (int i = 0; i < edge; ++i)
        ext[edge + n + i] = 2 * x[n - 1] - x[n - 2 - i];
}

static void reverse(double *x, int64_t n)
{
    for (int64_t i = 0, j = n - 1; i < j; ++i, --j) {
        double t = x[i];
        x[i] = x[j];
        x[j] = t;
    }
}

/* ---- scipy.signal.sosfiltfilt (_signaltools.py:4718-4828), one component.
 * edge = 3*ntaps, ntaps = 2*nsec+1 - min(#(b2==0), #(a2==0)).  Returns -1 if n <= edge
 * (scipy raises ValueError there).  y has n entries.                                   */
static int sosfiltfilt_real(const double *sos, const double *zi, int nsec, const double *x, int64_t n,
                            double *y)
{
    int nb0 = 0, na0 = 0;
    for (int s = 0; s < nsec; ++s) {
        nb0 += (sos[6 * s + 2] == 0.0);
        na0 += (sos[6 * s + 5] == 0.0);
    }
    int ntaps = 2 * nsec + 1 - (nb0 < na0 ? nb0 : na0);
    int edge = 3 * ntaps;
    if (n <= edge)
        return -1;
    int64_t ne = n + 2 * edge;
    double *ext = (double *)malloc((size_t)ne * sizeof(double));
    double *z = (double *)malloc((size_t)nsec * 2 * sizeof(double));
    odd_ext(x, n, edge, ext);
    for (int k = 0; k < 2 * nsec; ++k)
        z[k] = zi[k] * ext[0];
    sosfilt_real(sos, nsec, ext, ne, z);
    double y0 = ext[ne - 1];
    reverse(ext, ne);
    for (int k = 0; k < 2 * nsec; ++k)
        z[k] = zi[k] * y0;
    sosfilt_real(sos, nsec, ext, ne, z);
    reverse(ext, ne);
    memcpy(y, ext + edge, (size_t)n * sizeof(double));
    free(ext);
    free(z);
    return 0;
}

/* ---- scipy.signal.filtfilt, method='pad' (_signaltools.py:4358-4557), one component.
 * edge = 3*max(len(a),len(b)); -1 if n <= edge.                                        */
static int filtfilt_real(const double *b, const double *a, const double *zi, int order, const double *x,
                         int64_t n, double *y)
{
    int edge = 3 * (order + 1);
    if (n <= edge)
        return -1;
    int64_t ne = n + 2 * edge;
    double *ext = (double *)malloc((size_t)ne * sizeof(double));
    double z[16];
    odd_ext(x, n, edge, ext);
    for (int k = 0; k < order; ++k)
        z[k] = zi[k] * ext[0];
    lfilter_real(b, a, order, ext, ne, z);
    double y0 = ext[ne - 1];
    reverse(ext, ne);
    for (int k = 0; k < order; ++k)
        z[k] = zi[k] * y0;
    lfilter_real(b, a, order, ext, ne, z);
    reverse(ext, ne);
    memcpy(y, ext + edge, (size_t)n * sizeof(double));
    free(ext);
    return 0;
}

/* ---- non-finite samples.  The per-component form above is numpy's complex product only for FINITE data: scipy filters
 * complex samples with complex coefficients (b + 0j), whose zero cross terms (0*inf, 0*nan) turn a non-finite component
 * into NaN in BOTH components of that output; the forward pass carries it to the end of the extended array, the backward
 * pass -- started from zi * (the forward pass's last value) -- back to its start.  So one non-finite sample anywhere in a
 * zero-phase filter's input is NaN + NaN j at every output (goldens of the imported reference: tests/golden/nonfinite.npz;
 * a real float64 input is a real array there and comes out all-NaN the same way).  Returns 1 and fills when it applies. */
static int smear_nonfinite(const double *xr, const double *xi, int64_t n, double *yr, double *yi, int64_t m)
{
    int bad = 0;
    for (int64_t i = 0; i < n && !bad; ++i)
        bad = !isfinite(xr[i]) || !isfinite(xi[i]);
    if (bad)
        for (int64_t i = 0; i < m; ++i)
            yr[i] = yi[i] = NAN;
    return bad;
}

/* ---- scipy.signal.decimate(x, q) IIR zero-phase branch (_signaltools.py:4975-4989):
 * sosfiltfilt then y[::q].  out has ceil(n/q) entries. Returns that count or -1.       */
ORC_API int64_t orc_decimate(const double *sos, const double *soszi, int nsec, int q, const double *xr,
                             const double *xi, int64_t n, double *yr, double *yi)
{
    double *t = (double *)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
    int64_t m = (n + q - 1) / q;
    if (sosfiltfilt_real(sos, soszi, nsec, xr, n, t) != 0) {
        free(t);
        return -1;
    }
    if (smear_nonfinite(xr, xi, n, yr, yi, m)) {
        free(t);
        return m;
    }
    for (int64_t i = 0; i < m; ++i)
        yr[i] = t[i * q];
    sosfiltfilt_real(sos, soszi, nsec, xi, n, t);
    for (int64_t i = 0; i < m; ++i)
        yi[i] = t[i * q];
    free(t);
    return m;
}

/* ---- SignalProcessor.frequency_shift (processor.py:85-100):
 *   t = np.arange(n)/fs;  shift = np.exp(-1j*2*np.pi*f*t);  samples*shift
 * Python evaluates c = ((-1j*2)*pi)*f -> (0, -(2*pi)*f); c*t -> (0, c.imag*t[n]); exp -> (cos, sin). */
ORC_API void orc_frequency_shift(double *xr, double *xi, int64_t n, double freq_offset, double fs)
{
    double ci = -(2.0 * M_PI) * freq_offset;
    for (int64_t i = 0; i < n; ++i) {
        double t = (double)i / fs;
        double th = ci * t;
        double s, c;
        sincos(th, &s, &c);
        double a = xr[i], b = xi[i];
        xr[i] = a * c - b * s;
        xi[i] = a * s + b * c;
    }
}

/* ---- SignalProcessor.filter_signal's filtfilt (processor.py:79). In place; -1 => unchanged
 * (the reference catches the ValueError and returns the input, processor.py:81-83).     */
ORC_API int orc_filtfilt(const double *b, const double *a, const double *zi, int order, double *xr,
                         double *xi, int64_t n)
{
    if (n <= 3 * (order + 1))
        return -1;
    if (smear_nonfinite(xr, xi, n, xr, xi, n))
        return 0;
    double *t = (double *)malloc((size_t)n * sizeof(double));
    filtfilt_real(b, a, zi, order, xr, n, t);
    memcpy(xr, t, (size_t)n * sizeof(double));
    filtfilt_real(b, a, zi, order, xi, n, t);
    memcpy(xi, t, (size_t)n * sizeof(double));
    free(t);
    return 0;
}

/* numpy pairwise summation (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum_DOUBLE) so that
 * np.mean() is reproduced bit for bit, including near-tie timing-phase decisions.        */
static double pairwise_sum(const double *a, int64_t n, int64_t stride)
{
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; ++i)
            res += a[i * stride];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        for (int k = 0; k < 8; ++k)
            r[k] = a[k * stride];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k)
                r[k] += a[(i + k) * stride];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i)
            res += a[i * stride];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2, stride) + pairwise_sum(a + n2 * stride, n - n2, stride);
    }
}

/* ---- SignalProcessor.extract_symbols (processor.py:168-219).  Returns n_sym, writes the chosen
 * phase; power of phase p = mean(abs(x[p::sps][:n_p])**2), first strictly greater wins.  */
ORC_API int64_t orc_extract_symbols(const double *xr, const double *xi, int64_t n, double fs,
                                    double symbol_rate, double *sr, double *si, int32_t *best_phase_out,
                                    double *phase_power_out /* [sps] or NULL */)
{
    if (n == 0)
        return 0;
    int64_t sps = (int64_t)(fs / symbol_rate);
    if (sps <= 1) {
        memcpy(sr, xr, (size_t)n * sizeof(double));
        memcpy(si, xi, (size_t)n * sizeof(double));
        if (best_phase_out)
            *best_phase_out = 0;
        return n;
    }
    int64_t step = sps / 8 > 1 ? sps / 8 : 1;
    int64_t best = 0;
    double maxp = -1;
    double *p2 = (double *)malloc((size_t)(n / sps + 2) * sizeof(double));
    for (int64_t ph = 0; ph < sps; ph += step) {
        int64_t ns = (n - ph) / sps;
        if (phase_power_out)
            phase_power_out[ph] = -1;
        if (n - ph <= 0 || ns <= 0)
            continue;
        for (int64_t k = 0; k < ns; ++k) {
            double m = hypot(xr[ph + k * sps], xi[ph + k * sps]);
            p2[k] = m * m;
        }
        double power = pairwise_sum(p2, ns, 1) / (double)ns;
        if (phase_power_out)
            phase_power_out[ph] = power;
        if (power > maxp) {
            maxp = power;
            best = ph;
        }
    }
    free(p2);
    int64_t ns = (n - best) / sps;
    for (int64_t k = 0; k < ns; ++k) {
        sr[k] = xr[best + k * sps];
        si[k] = xi[best + k * sps];
    }
    if (best_phase_out)
        *best_phase_out = (int32_t)best;
    return ns;
}

/* ---- SignalProcessor.demodulate_dqpsk (processor.py:102-166).  Returns n-1 (0 if n<2).
 * samples/max|s| is numpy complex/(m+0j) = Smith's algorithm = component * (1.0/m).
 * margin_out (optional): min |phase - threshold| over all decisions.                    */
ORC_API int64_t orc_demodulate_dqpsk(const double *sr, const double *si, int64_t n, uint8_t *out,
                                     double *margin_out)
{
    if (margin_out)
        *margin_out = INFINITY;
    if (n < 2)
        return 0;
    double maxm = 0;
    for (int64_t i = 0; i < n; ++i) {
        double m = hypot(sr[i], si[i]);
        if (m > maxm || i == 0)
            maxm = m; /* np.max */
    }
    double scl = 1.0;
    int norm = maxm > 0;
    if (norm)
        scl = 1.0 / maxm;
    const double t0 = -5 * M_PI / 8, t1 = -3 * M_PI / 8, t2 = 3 * M_PI / 8, t3 = 5 * M_PI / 8;
    double pr = norm ? (sr[0] + si[0] * 0.0) * scl : sr[0];
    double pi_ = norm ? (si[0] - sr[0] * 0.0) * scl : si[0];
    for (int64_t i = 1; i < n; ++i) {
        double cr = norm ? (sr[i] + si[i] * 0.0) * scl : sr[i];
        double ci = norm ? (si[i] - sr[i] * 0.0) * scl : si[i];
        /* diff = sample * conj(prev) */
        double br = pr, bi = -pi_;
        double dr = cr * br - ci * bi;
        double di = cr * bi + ci * br;
        double ph = atan2(di, dr);
        uint8_t sym;
        if (ph < t0)
            sym = 3;
        else if (ph < t1)
            sym = 2;
        else if (ph < t2)
            sym = 0;
        else if (ph < t3)
            sym = 1;
        else
            sym = 3;
        out[i - 1] = sym;
        if (margin_out) {
            double m = fabs(ph - t0);
            if (fabs(ph - t1) < m) m = fabs(ph - t1);
            if (fabs(ph - t2) < m) m = fabs(ph - t2);
            if (fabs(ph - t3) < m) m = fabs(ph - t3);
            if (m < *margin_out) *margin_out = m;
        }
        pr = cr;
        pi_ = ci;
    }
    return n - 1;
}

/* ---- SignalProcessor.process (processor.py:221-273), given pre-designed filters.
 *  q          decimation factor (1 = no decimation branch)
 *  sos/soszi  cheby1 sections for the decimator (ignored when q == 1)
 *  b1/a1/zi1  butter(4) designed for the post-decimation rate   (used when decimation succeeded)
 *  b0/a0/zi0  butter(4) designed for the ORIGINAL rate          (used when decimate raised, n<=27,
 *             because the reference then leaves current_rate unchanged, processor.py:253-257)
 * Writes soft symbols (self.symbols) and hard symbols; returns n_hard, *n_soft_out = n_soft.   */
ORC_API int64_t orc_process(const double *xr_in, const double *xi_in, int64_t n, double sample_rate,
                            double freq_offset, int q, const double *sos, const double *soszi, int nsec,
                            const double *b1, const double *a1, const double *zi1, const double *b0,
                            const double *a0, const double *zi0, double *soft_r, double *soft_i,
                            int64_t *n_soft_out, uint8_t *hard, int32_t *best_phase_out,
                            double *margin_out)
{
    *n_soft_out = 0;
    if (best_phase_out)
        *best_phase_out = 0;
    if (margin_out)
        *margin_out = INFINITY;
    if (n == 0)
        return 0;
    double *wr = (double *)malloc((size_t)n * sizeof(double));
    double *wi = (double *)malloc((size_t)n * sizeof(double));
    int64_t m = n;
    double rate = sample_rate;
    const double *b = b0, *a = a0, *zi = zi0;
    int decimated = 0;
    if (q > 1) {
        int64_t r = orc_decimate(sos, soszi, nsec, q, xr_in, xi_in, n, wr, wi);
        if (r >= 0) {
            m = r;
            rate = sample_rate / q;
            b = b1; a = a1; zi = zi1;
            decimated = 1;
        }
    }
    if (!decimated) {
        memcpy(wr, xr_in, (size_t)n * sizeof(double));
        memcpy(wi, xi_in, (size_t)n * sizeof(double));
    }
    if (freq_offset != 0)
        orc_frequency_shift(wr, wi, m, freq_offset, rate);
    orc_filtfilt(b, a, zi, 4, wr, wi, m); /* failure leaves the data unfiltered */
    int64_t ns = orc_extract_symbols(wr, wi, m, rate, 18000.0, soft_r, soft_i, best_phase_out, NULL);
    *n_soft_out = ns;
    int64_t nh = orc_demodulate_dqpsk(soft_r, soft_i, ns, hard, margin_out);
    free(wr);
    free(wi);
    return nh;
}
