// Host-side plan of the reference-parity pipeline: the decisions SignalProcessor.process makes
// per call (processor.py:239-273) turned into a static description for one
// (sample_rate, n_samples) pair, plus the filter tables for the two zero-phase stages.
#pragma once
#include <cstdint>

#include "design.hpp"
#include "zp_tables.hpp"
#include "pz_tables.hpp"
#include "lp2_tables.hpp"

namespace tdm {

#define TDM_LDEC 32
constexpr int kLDec = TDM_LDEC;  // samples per lane, decimator stage (4 biquads)
#define TDM_LLPF 8
constexpr int kLLpf = TDM_LLPF;  // samples per lane, channel-filter stage (order 4 = 2 biquads; LDS-staged I/O)
constexpr int kEdgeSos = 27;  // sosfiltfilt pad for 4 sections: 3*(2*4+1)
constexpr int kEdgeTf = 15;   // filtfilt pad for order 4: 3*5
constexpr double kSymbolRate = 18000.0;

// (decimation factor, outputs per lane) of the parallel-form decimator kernels, samples as doubles / raw bytes: the one
// list the launch switches (ref_pipeline.hpp), the instantiations (tdm_k_pz.hip, tdm_k_raw.hip) and the tables below share
#define TDM_PZ_CASES(X) X(2, 16) X(3, 10) X(4, 8) X(5, 6) X(6, 5) X(7, 4) X(8, 4) X(9, 3) X(10, 3) X(11, 2) X(12, 2) X(13, 2) X(14, 2) X(15, 2) X(16, 2) X(41, 1)
#define TDM_PZR_CASES_A(X) X(3, 16) X(6, 16) X(8, 15) X(12, 10) X(41, 2)
#define TDM_PZR_CASES_B(X) X(4, 16) X(7, 16) X(10, 12) X(13, 8)
#define TDM_PZR_CASES(X) TDM_PZR_CASES_A(X) TDM_PZR_CASES_B(X)

// Parallel-form decimator kernels exist for these decimation factors; S = outputs per lane (lane length
// S*q <= 32 samples, 41 for the 10 MS/s case).  Any other factor runs on the cascade engine.
inline int pz_outputs_per_lane(int q)
{
    switch (q) {
#define TDM_PZ_S(Q, S) case Q: return S;
        TDM_PZ_CASES(TDM_PZ_S)
#undef TDM_PZ_S
    default: return 0;
    }
}

// Raw-integer decimator kernels (pz_raw_body: cu8 samples kept as bytes, about 120 per lane) exist for the decimation
// factors of the RTL-SDR rates and of 10 MS/s.  S = outputs per lane (lane length S*q, even).
inline int pz_raw_outputs_per_lane(int q)
{
    switch (q) {
#define TDM_PZR_S(Q, S) case Q: return S;
        TDM_PZR_CASES(TDM_PZR_S)
#undef TDM_PZR_S
    default: return 0;
    }
}

struct RefPlanHost {
    double sample_rate = 0;
    int64_t n = 0;
    int q = 1;              // factor the reference would use (processor.py:248-250)
    bool decimated = false; // false also when decimate raises (n <= 27, processor.py:253-257)
    double rate_dec = 0;    // current_rate after the decimation step
    int64_t n_dec = 0;
    bool lpf = false;       // false when filtfilt raises (n_dec <= 15, processor.py:81-83)
    int sps = 0;            // int(rate_dec / 18000)
    int phase_step = 1;
    int64_t max_soft = 0;
    Sos4 sos{};
    Tf4 tf{};
    int pz_S = 0;           // > 0: the decimator runs in parallel form with pz_S outputs per lane
    ZpHostTables dec;       // valid if decimated
    ZpHostTables lpf_t;     // valid if lpf
    Lp2Host lp2;            // lp2.ok: the low-rate stage runs as one parallel-form kernel (lp2_kernels.hpp)
    // cu8 plans: the same two stages with the decimator on the raw bytes (longer lanes, other block geometry); used
    // whenever no input-rate pre-shift is requested
    int raw_S = 0;
    // the raw-integer kernel's blocks are four times as long: with fewer than raw_min_blocks blocks in the whole batch
    // (a few carriers) they cannot fill the chip and the double-based kernel finishes sooner.  0 = always raw.
    int64_t raw_min_blocks = 0;
    ZpHostTables dec_raw;
    Lp2Host lp2_raw;
};

// sos rows b = g*[1,2,1], a -> device form (unit numerators, one input gain, matching zi)
inline ZpFilterDesc desc_from_rows(const double (*sos)[6], int nsec)
{
    ZpFilterDesc f{};
    f.nsec = nsec;
    f.K = 2;
    long double g = 1.0L;
    double scale = 1.0;
    for (int i = 0; i < nsec; ++i) {
        g *= (long double)sos[i][0];
        f.b[i][0] = 1.0; f.b[i][1] = 2.0; f.b[i][2] = 1.0;
        for (int k = 0; k < 3; ++k) f.a[i][k] = sos[i][3 + k];
        double zi2[2];
        lfilter_zi(f.b[i], f.a[i], 2, zi2);
        f.zi[i][0] = scale * zi2[0];
        f.zi[i][1] = scale * zi2[1];
        scale *= 4.0 / ((f.a[i][0] + f.a[i][1]) + f.a[i][2]);
    }
    f.in_gain = (double)(g * g);
    return f;
}

inline bool rows_are_lp121(const double (*sos)[6], int nsec)
{
    for (int i = 0; i < nsec; ++i)
        if (sos[i][1] != 2 * sos[i][0] || sos[i][2] != sos[i][0] || sos[i][3] != 1.0) return false;
    return true;
}

inline ZpFilterDesc desc_from_sos(const Sos4 &s) { return desc_from_rows(s.sos, 4); }

// the order-4 channel filter runs as its two-biquad factorisation (see Tf4 in design.hpp)
inline ZpFilterDesc desc_from_tf(const Tf4 &t) { return desc_from_rows(t.sos, 2); }

// base: a plan of the same sample rate, bandwidth and wire format (another length): its designs and the
// length-independent tables of the parallel-form stages are reused, only the length-dependent part is built
inline RefPlanHost build_ref_plan(double sample_rate, int64_t n, double bandwidth = 25000.0, bool allow_pz = true, int in_fmt = -1,
                                  const RefPlanHost *base = nullptr)
{
    RefPlanHost h;
    h.sample_rate = sample_rate;
    h.n = n;
    h.q = decimation_factor(sample_rate);
    h.decimated = (h.q > 1 && n > kEdgeSos);
    h.rate_dec = h.decimated ? sample_rate / h.q : sample_rate;
    h.n_dec = h.decimated ? (n + h.q - 1) / h.q : n;
    h.lpf = h.n_dec > kEdgeTf;
    h.sps = (int)(h.rate_dec / kSymbolRate);
    h.phase_step = h.sps / 8 > 1 ? h.sps / 8 : 1;
    h.max_soft = h.sps > 1 ? h.n_dec / h.sps + 1 : h.n_dec;
    if (h.max_soft < 1) h.max_soft = 1;
    if (base && base->sample_rate != sample_rate) base = nullptr;
    if (h.q > 1) h.sos = base ? base->sos : design_cheby1_8(0.05, 0.8 / h.q);
    h.tf = (base && base->rate_dec == h.rate_dec) ? base->tf : design_butter4(butter_cutoff(bandwidth, h.rate_dec));
    if (h.decimated) {
        h.pz_S = (allow_pz && rows_are_lp121(h.sos.sos, 4)) ? pz_outputs_per_lane(h.q) : 0;
        if (h.pz_S) {
            auto sh = (base && base->pz_S == h.pz_S && base->dec.shared) ? std::static_pointer_cast<const PzShared>(base->dec.shared)
                                                                         : build_pz_shared(h.sos.sos, 4, kEdgeSos, h.q * h.pz_S, h.pz_S, h.q);
            h.dec = build_pz_tables(sh, n, h.n_dec);
        } else {
            h.dec = build_zp_tables(desc_from_sos(h.sos), n, kEdgeSos, kLDec, h.n_dec, h.q);
        }
    }
    if (allow_pz && h.lpf && h.sps > 1 && h.sps <= 32 && (!h.decimated || h.pz_S) && rows_are_lp121(h.tf.sos, 2))
        h.lp2 = build_lp2(h.tf.sos, h.n_dec, kEdgeTf, h.sps, h.decimated ? &h.dec : nullptr, h.sos.sos);
    if (h.lpf && !h.lp2.ok)   // (the cascade engine's tables: only where the one-kernel low-rate stage does not apply)
        h.lpf_t = build_zp_tables(desc_from_tf(h.tf), h.n_dec, kEdgeTf, kLLpf, h.n_dec, 1);
    if (h.lp2.ok && h.decimated && in_fmt == 0 /* FMT_CU8 */ && pz_raw_outputs_per_lane(h.q)) {
        const int S = pz_raw_outputs_per_lane(h.q);
        // x = fl(1/127.5) * u - 1 (pyrtlsdr's conversion without its two roundings, see pz_raw_body); the kernel's samples
        // are u + 4096 (PzRawBias)
        auto sh = (base && base->raw_S == S && base->dec_raw.shared) ? std::static_pointer_cast<const PzShared>(base->dec_raw.shared)
                                                                     : build_pz_shared(h.sos.sos, 4, kEdgeSos, h.q * S, S, h.q, 1.0 / 127.5, 1.0, (double)kPzRawBiasCu8);
        h.dec_raw = build_pz_tables(sh, n, h.n_dec);
        h.lp2_raw = build_lp2(h.tf.sos, h.n_dec, kEdgeTf, h.sps, &h.dec_raw, h.sos.sos);
        if (h.lp2_raw.ok) h.raw_S = S;
    }
    return h;
}

}  // namespace tdm
