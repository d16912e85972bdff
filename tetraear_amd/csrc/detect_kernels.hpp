// Scanner heuristics on device: TetraSignalDetector
//   calculate_power          tetraear/signal/scanner.py:42-55
//   detect_tetra_modulation  tetraear/signal/scanner.py:57-100
//   detect_sync_pattern      tetraear/signal/scanner.py:102-147
// One workgroup per row of complex128 samples.
//   Comm: tid(), nthreads(), sync(), reduce_sum/max(double)
#pragma once
#include "zp_kernels.hpp"

namespace tdm {

constexpr int kDetectOut = 8;  // power_db, is_tetra, confidence, found_sync, max_corr, 0, 0, 0

struct DetectArgs {
    const double *x;      // [rows][n] c128
    int64_t n;
    double fs;
    double bottom_threshold;  // calculate_power's value for empty input (-85)
    double *ang;          // [rows][n] scratch
    uint8_t *bits;        // [rows][n] scratch
    double *out;          // [rows][kDetectOut]
};

// numpy's float `%` (npy_divmod remainder): sign follows the divisor
TDM_HD double py_mod(double a, double b)
{
    double r = fmod(a, b);
    if (r != 0.0) {
        if ((b < 0) != (r < 0)) r += b;
    } else {
        r = copysign(0.0, b);
    }
    return r;
}

TDM_HD double wrap_pi(double d) { return py_mod(d + M_PI, 2 * M_PI) - M_PI; }

template <class Comm>
TDM_HD void detect_body(const DetectArgs &A, Comm &cm, int row)
{
    const int tid = cm.tid(), nt = cm.nthreads();
    const int64_t n = A.n;
    const double *x = A.x + (int64_t)row * n * 2;
    double *ang = A.ang + (int64_t)row * n;
    uint8_t *bits = A.bits + (int64_t)row * n;
    double *out = A.out + (int64_t)row * kDetectOut;
    // ---- calculate_power: 10*log10(mean(abs(x)**2) + 1e-10)
    double acc = 0, mx = 0;
    for (int64_t i = tid; i < n; i += nt) {
        const double m = hypot(x[2 * i], x[2 * i + 1]);
        acc += m * m;
        mx = fmax(mx, m);
    }
    acc = cm.reduce_sum(acc);
    mx = cm.reduce_max(mx);
    const double power_db = n == 0 ? A.bottom_threshold : 10.0 * log10(acc / (double)n + 1e-10);
    // ---- detect_tetra_modulation
    double is_tetra = 0, confidence = 0;
    if (n >= 1000) {
        const double scl = 1.0 / (mx + 1e-10);  // complex / real == component * fl(1/s) in numpy
        for (int64_t i = tid; i < n; i += nt) ang[i] = atan2(mul_rn(x[2 * i + 1], scl), mul_rn(x[2 * i], scl));
        cm.sync();
        const double e0 = -M_PI, e1 = -3 * M_PI / 4, e2 = -M_PI / 2, e3 = -M_PI / 4, e5 = M_PI / 4, e6 = M_PI / 2,
                     e7 = 3 * M_PI / 4;
        const double tol = M_PI / 8;
        double matches = 0;
        for (int64_t i = tid; i + 1 < n; i += nt) {
            const double w = wrap_pi(ang[i + 1] - ang[i]);
            double d = fabs(e0 - w);
            d = fmin(d, fabs(e1 - w)); d = fmin(d, fabs(e2 - w)); d = fmin(d, fabs(e3 - w));
            d = fmin(d, fabs(0.0 - w)); d = fmin(d, fabs(e5 - w)); d = fmin(d, fabs(e6 - w));
            d = fmin(d, fabs(e7 - w));
            if (d < tol) matches += 1.0;
        }
        matches = cm.reduce_sum(matches);
        confidence = matches / (double)(n - 1);
        is_tetra = confidence > 0.4 ? 1.0 : 0.0;
        cm.sync();
    }
    // ---- detect_sync_pattern
    double found_sync = 0, max_corr = 0;
    {
        int ds = (int)(A.fs / 18000.0 / 10.0);
        if (ds < 1) ds = 1;
        const int64_t ns = (n + ds - 1) / ds;  // len(samples[::ds])
        if (ns >= 100) {
            for (int64_t i = tid; i < ns; i += nt) ang[i] = atan2(x[2 * i * ds + 1], x[2 * i * ds]);
            cm.sync();
            const int64_t nb = ns - 1;
            for (int64_t i = tid; i < nb; i += nt) {
                const double w = wrap_pi(ang[i + 1] - ang[i]);
                const double q = rint(w / (M_PI / 4)) * (M_PI / 4);  // Python round(): half to even
                bits[i] = fabs(q) < M_PI / 8 ? 1 : 0;
            }
            cm.sync();
            if (nb >= 31) {
                const uint32_t pat = 0b0101100111000100101100111000100u;  // scanner.py:131-132, first element = MSB
                double best = 0;
                for (int64_t i = tid; i < nb - 31; i += nt) {
                    uint32_t w = 0;
                    for (int k = 0; k < 31; ++k) w = (w << 1) | bits[i + k];
#if defined(__HIP_DEVICE_COMPILE__)
                    const int diff = __popc(w ^ pat);
#else
                    const int diff = __builtin_popcount(w ^ pat);
#endif
                    best = fmax(best, (double)(31 - diff) / 31.0);
                }
                max_corr = cm.reduce_max(best);
                found_sync = max_corr > 0.75 ? 1.0 : 0.0;
            }
        }
    }
    if (tid == 0) {
        out[0] = power_db; out[1] = is_tetra; out[2] = confidence; out[3] = found_sync; out[4] = max_corr;
        out[5] = 0; out[6] = 0; out[7] = 0;
    }
}

}  // namespace tdm
