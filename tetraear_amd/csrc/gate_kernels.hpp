// Spectrum / AFC / signal gate in front of process().
// Restates the block of CaptureThread.run that decides WHETHER process() is called and with WHICH
// freq_offset (tetraear/ui/modern.py:1921-2021):
//   fft   = fftshift(fft(samples[:2048] * hanning(2048)))                        :1923-1929
//   power = 20*log10(abs(fft)/2048 + 1e-20)                                      :1934
//   band  = +-(int(25000/(fs/2048))//2) bins around the centre                    :1948-1954
//   signal_power = mean(power[band]); peak_power = max; peak bin -> freq offset  :1957-1967
//   noise_floor  = mean(power outside band +- 10 bins)                           :1971-1986
//   strong = snr > 15 and peak_power > -70 and peak_power - signal_power > 3    :1989-2001
//   afc    = peak_freq_offset if strong and peak_power > -70 else 0             :2021
// One workgroup per carrier: 2048-point radix-2 FFT in LDS (fp64), then workgroup reductions.
//   Comm: tid(), nthreads(), sync(), smem() -> >= 4*2048 doubles of workgroup scratch,
//         reduce_sum/max/min(double)
#pragma once
#include "zp_kernels.hpp"

namespace tdm {

constexpr int kGateFft = 2048;
constexpr int kGateLog2 = 11;
constexpr int kGateOut = 8;  // doubles per carrier: see gate_body

struct GateArgs {
    const void *iq;
    int64_t row_stride;   // samples
    int64_t n;            // samples per row (gate needs n >= 2048)
    int32_t fmt;
    int32_t pad_;
    double fs;
    double *out;          // [rows][kGateOut]: peak_freq_offset, signal_power, peak_power, noise_floor, snr, strong, afc, 0
    double *afc;          // [rows] or null: freq_offset to hand to process()
};

TDM_HD void gate_load(const void *rowp, int fmt, int64_t k, double &re, double &im)
{
    switch (fmt) {
    case FMT_CU8: convert_one<FMT_CU8>(rowp, k, re, im); break;
    case FMT_CS8: convert_one<FMT_CS8>(rowp, k, re, im); break;
    case FMT_CF32: convert_one<FMT_CF32>(rowp, k, re, im); break;
    default: convert_one<FMT_CF64>(rowp, k, re, im); break;
    }
}

TDM_HD unsigned bitrev11(unsigned x)
{
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < kGateLog2; ++i) r |= ((x >> i) & 1u) << (kGateLog2 - 1 - i);
    return r;
}

template <class Comm>
TDM_HD void gate_body(const GateArgs &A, Comm &cm, int row)
{
    constexpr int N = kGateFft;
    const int tid = cm.tid(), nt = cm.nthreads();
    double *xr = cm.smem();        // [N]
    double *xi = xr + N;           // [N]
    double *wr = xi + N;           // [N/2] twiddles exp(-2 pi i k / N)
    double *wi = wr + N / 2;       // [N/2]
    double *outp = A.out + (int64_t)row * kGateOut;
    if (A.n < N) {  // the reference skips the whole block when len(samples) < n_fft
        if (tid == 0) {
            for (int k = 0; k < kGateOut; ++k) outp[k] = 0.0;
            if (A.afc) A.afc[row] = 0.0;
        }
        return;
    }
    const void *rowp = (const char *)A.iq + (int64_t)row * A.row_stride *
                                                (A.fmt == FMT_CU8 || A.fmt == FMT_CS8 ? 2 : (A.fmt == FMT_CF32 ? 8 : 16));
    for (int i = tid; i < N; i += nt) {
        double re, im;
        gate_load(rowp, A.fmt, i, re, im);
        const double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)(N - 1));  // np.hanning
        const unsigned j = bitrev11((unsigned)i);
        xr[j] = re * w;
        xi[j] = im * w;
    }
    for (int k = tid; k < N / 2; k += nt) {
        double s, c;
        sincospi_d(2.0 * (double)k / (double)N, &s, &c);
        wr[k] = c;
        wi[k] = -s;
    }
    cm.sync();
    // decimation-in-time radix-2
    for (int st = 1; st <= kGateLog2; ++st) {
        const int half = 1 << (st - 1);
        const int tstep = N >> st;
        for (int b = tid; b < N / 2; b += nt) {
            const int grp = b / half, pos = b - grp * half;
            const int i0 = grp * (half << 1) + pos, i1 = i0 + half;
            const double c = wr[pos * tstep], s = wi[pos * tstep];
            const double tr = xr[i1] * c - xi[i1] * s, ti = xr[i1] * s + xi[i1] * c;
            const double ar = xr[i0], ai = xi[i0];
            xr[i0] = ar + tr; xi[i0] = ai + ti;
            xr[i1] = ar - tr; xi[i1] = ai - ti;
        }
        cm.sync();
    }
    // power in dB, fftshift order: shifted index p <-> bin (p + N/2) mod N
    double *power = wi + N / 2;    // [N]
    for (int p = tid; p < N; p += nt) {
        const int k = (p + N / 2) & (N - 1);
        power[p] = 20.0 * log10(hypot(xr[k], xi[k]) / (double)N + 1e-20);
    }
    cm.sync();
    const int center = N / 2;
    const double freq_res = A.fs / (double)N;
    const int bw_bins = (int)(25000.0 / freq_res);
    int start = center - bw_bins / 2;
    if (start < 0) start = 0;
    int end = center + bw_bins / 2;
    if (end > N) end = N;
    double peak_off = 0, sig = 0, peak = 0, noise = -100.0, snr = 0, strong = 0, afc = 0;
    if (end > start) {
        double acc = 0, mx = -1e300;
        for (int p = start + tid; p < end; p += nt) { acc += power[p]; mx = fmax(mx, power[p]); }
        sig = cm.reduce_sum(acc) / (double)(end - start);
        peak = cm.reduce_max(mx);
        double cand = 1e9;
        for (int p = start + tid; p < end; p += nt)
            if (power[p] == peak) cand = fmin(cand, (double)p);   // np.argmax: first maximum
        const int peak_idx = (int)cm.reduce_min(cand);
        // fftshift(fftfreq(N, 1/fs))[peak_idx]: numpy forms k * (1.0 / (N * d)) with d = 1/fs rounded first
        const double d = 1.0 / A.fs;
        peak_off = (double)(peak_idx - center) * (1.0 / ((double)N * d));
        int n1 = start - 10;
        if (n1 < 0) n1 = 0;
        int s2 = end + 10;
        if (s2 > N) s2 = N;
        double nacc = 0;
        for (int p = tid; p < n1; p += nt) nacc += power[p];
        for (int p = s2 + tid; p < N; p += nt) nacc += power[p];
        const int ncount = n1 + (N - s2);
        nacc = cm.reduce_sum(nacc);
        noise = ncount > 0 ? nacc / (double)ncount : -100.0;
        snr = sig - noise;
        strong = (snr > 15.0 && peak > -70.0 && (peak - sig) > 3.0) ? 1.0 : 0.0;
        afc = (strong != 0.0 && peak > -70.0) ? peak_off : 0.0;
    }
    if (tid == 0) {
        outp[0] = peak_off; outp[1] = sig; outp[2] = peak; outp[3] = noise;
        outp[4] = snr; outp[5] = strong; outp[6] = afc; outp[7] = 0.0;
        if (A.afc) A.afc[row] = afc;
    }
}

}  // namespace tdm
