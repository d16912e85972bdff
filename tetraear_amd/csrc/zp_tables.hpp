// Host-side construction of the linear-response tables the block engine needs (zp_common.hpp).
// Every table is obtained by simulating the same DF2T recurrences on unit states in long double
// and rounding once to double.
#pragma once
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "zp_common.hpp"

namespace tdm {

// Device form of a lowpass cascade.  The reference's sections are b = g_s*[1,2,1]; the zero-phase
// operator is linear, so the total gain of both passes, (prod g_s)^2, is applied once to the input
// samples (`in_gain`) and every section runs with b = [1,2,1] (lp121_step).  `zi` is the
// steady-state start of THAT cascade per unit input (sosfilt_zi of the unit-numerator sections),
// which is the same physical start condition scipy's zi*x0 sets up.
struct ZpFilterDesc {
    int nsec, K;
    double b[kMaxSec][kMaxOrd + 1];
    double a[kMaxSec][kMaxOrd + 1];
    double zi[kMaxSec][kMaxOrd];
    double in_gain;
};

// Tables of a stage that do not depend on the row length (parallel-form stages: scan matrices, block transitions, the
// full blocks' carry-response tables): built once per plan and shared by every length the plan serves.
struct ZpSharedTables {
    std::vector<double> blob;
    virtual ~ZpSharedTables() = default;
};

struct ZpHostTables {
    ZpParams p;                // table/work pointers are null until patched by the owner
    std::vector<double> blob;  // all tables, concatenated (with `shared`: only those that depend on the row length)
    std::shared_ptr<const ZpSharedTables> shared;   // (parallel form) Mpow, T1/T2reg, Mf, Ureg live here
    size_t off_Mpow, off_zirh, off_cflast, off_T1reg, off_T2reg, off_T1last, off_T2last, off_Mf, off_Mblast, off_Ureg, off_Ulast, off_pz = 0;
    // point p's table pointers into a copy of blob that lives at `base` (and a copy of shared->blob at `sbase`)
    void bind(ZpParams &q, const double *base, const double *sbase = nullptr) const
    {
        const double *rb = shared ? sbase : base;
        q.Mpow = rb + off_Mpow; q.zirh = base + off_zirh; q.cf_last = base + off_cflast;
        q.T1_reg = rb + off_T1reg; q.T2_reg = rb + off_T2reg;
        q.T1_last = base + off_T1last; q.T2_last = base + off_T2last; q.Mf = rb + off_Mf;
        q.Mb_last = base + off_Mblast; q.U_reg = rb + off_Ureg; q.U_last = base + off_Ulast;
        q.pz = q.pform ? base + off_pz : nullptr;
    }
    const double *reg_tables() const { return shared ? shared->blob.data() : blob.data(); }   // host base of the *_reg offsets
};

namespace detail {
template <int K>
inline long double cascade_step(const ZpFilterDesc &f, long double x, long double *Z)
{
    long double y = x;
    for (int s = 0; s < f.nsec; ++s) {
        long double b[kMaxOrd + 1], a[kMaxOrd + 1];
        for (int k = 0; k <= K; ++k) { b[k] = f.b[s][k]; a[k] = f.a[s][k]; }
        y = df2t_step<K, long double>(b, a, y, Z + s * K);
    }
    return y;
}

template <int K>
inline void build(const ZpFilterDesc &f, ZpHostTables &t)
{
    ZpParams &p = t.p;
    const int L = p.L, Bn = kWave * L, D = f.nsec * K;
    const int len_last = p.len_last;
    std::vector<double> &blob = t.blob;
    auto reserve = [&](size_t n) { size_t o = blob.size(); blob.resize(o + n, 0.0); return o; };
    t.off_Mpow = reserve((size_t)f.nsec * kScanSteps * K * K);
    t.off_zirh = reserve((size_t)f.nsec * L * K);
    const int qs = p.out_stride;
    p.R_reg = (Bn + qs - 1) / qs;
    p.R_last = (len_last + qs - 1) / qs;
    t.off_cflast = reserve((size_t)D);
    t.off_T1reg = reserve((size_t)qs * p.R_reg * D);
    t.off_T2reg = reserve((size_t)qs * p.R_reg * D);
    t.off_T1last = reserve((size_t)qs * p.R_last * D);
    t.off_T2last = reserve((size_t)qs * p.R_last * D);
    auto prow = [&](int m, int R) { return (size_t)(m % qs) * R + (size_t)(m / qs); };
    t.off_Mf = reserve((size_t)D * D);
    t.off_Mblast = reserve((size_t)D * D);
    t.off_Ureg = reserve((size_t)D * D);
    t.off_Ulast = reserve((size_t)D * D);

    // ---- per-section scan matrices: Mpow[s][j] = A_s^(L*2^j)
    for (int s = 0; s < f.nsec; ++s) {
        long double b[kMaxOrd + 1], a[kMaxOrd + 1];
        for (int k = 0; k <= K; ++k) { b[k] = f.b[s][k]; a[k] = f.a[s][k]; }
        for (int k = 0; k < K; ++k) {
            long double z[kMaxOrd] = {0, 0, 0, 0};
            z[k] = 1.0L;
            int step = 0;
            for (int j = 0; j < kScanSteps; ++j) {
                const int target = L << j;
                for (; step < target; ++step) (void)df2t_step<K, long double>(b, a, 0.0L, z);
                for (int r = 0; r < K; ++r)
                    blob[t.off_Mpow + (((size_t)s * kScanSteps + j) * K + r) * K + k] = (double)z[r];
            }
        }
    }
    // ---- per-section zero-input response: zirh[s][i][k] = output at step i of section s started in state e_k
    for (int s = 0; s < f.nsec; ++s) {
        long double b[kMaxOrd + 1], a[kMaxOrd + 1];
        for (int k = 0; k <= K; ++k) { b[k] = f.b[s][k]; a[k] = f.a[s][k]; }
        for (int k = 0; k < K; ++k) {
            long double z[kMaxOrd] = {0, 0, 0, 0};
            z[k] = 1.0L;
            for (int i = 0; i < L; ++i)
                blob[t.off_zirh + ((size_t)s * L + i) * K + k] = (double)df2t_step<K, long double>(b, a, 0.0L, z);
        }
    }
    // ---- whole-cascade tables
    std::vector<long double> zir((size_t)Bn);
    for (int k = 0; k < D; ++k) {
        long double Z[kMaxD] = {0};
        Z[k] = 1.0L;
        for (int i = 0; i < Bn; ++i) {
            long double y = cascade_step<K>(f, 0.0L, Z);
            zir[i] = y;
            if (i + 1 == len_last) blob[t.off_cflast + k] = (double)y;
            if (i + 1 == len_last)
                for (int r = 0; r < D; ++r) blob[t.off_Mblast + r * D + k] = (double)Z[r];
        }
        for (int r = 0; r < D; ++r) blob[t.off_Mf + r * D + k] = (double)Z[r];
        // backward run over the forward zero-input response, regular and last-block lengths
        for (int v = 0; v < 2; ++v) {
            const int len = v ? len_last : Bn;
            long double W[kMaxD] = {0};
            const size_t off1 = v ? t.off_T1last : t.off_T1reg;
            const size_t off2 = v ? t.off_T2last : t.off_T2reg;
            const int R = v ? p.R_last : p.R_reg;
            for (int i = len - 1; i >= 0; --i) {
                long double y = cascade_step<K>(f, zir[i], W);
                blob[off1 + prow(i, R) * D + k] = (double)y;
                // backward zero-input response reaching offset i from the block's right end
                blob[off2 + prow(i, R) * D + k] = (double)zir[len - 1 - i];
            }
            const size_t uo = v ? t.off_Ulast : t.off_Ureg;
            for (int r = 0; r < D; ++r) blob[uo + r * D + k] = (double)W[r];
        }
    }
}
}  // namespace detail

// n: signal length per row; edge: odd-extension length; L: samples per lane;
// n_out/out_stride: outputs are padded-ext positions k0L + j*out_stride, j < n_out.
inline ZpHostTables build_zp_tables(const ZpFilterDesc &f, int64_t n, int edge, int L, int64_t n_out,
                                    int out_stride)
{
    ZpHostTables t;
    std::memset(&t.p, 0, sizeof(t.p));
    ZpParams &p = t.p;
    p.nsec = f.nsec;
    p.K = f.K;
    std::memcpy(p.b, f.b, sizeof(p.b));
    std::memcpy(p.a, f.a, sizeof(p.a));
    std::memcpy(p.zi, f.zi, sizeof(p.zi));
    p.in_gain = f.in_gain;
    p.n = n;
    p.edge = edge;
    p.L = L;
    p.P0 = (L - edge % L) % L;
    p.k0L = p.P0 + edge;
    p.Ne = p.P0 + n + 2 * (int64_t)edge;
    const int64_t Bn = (int64_t)kWave * L;
    p.nb = (int32_t)((p.Ne + Bn - 1) / Bn);
    p.len_last = (int32_t)(p.Ne - (int64_t)(p.nb - 1) * Bn);
    p.n_out = n_out;
    p.out_stride = out_stride;
    if (f.K == 2)
        detail::build<2>(f, t);
    else
        detail::build<4>(f, t);
    // carry series length: smallest t with max|Mf^t| < 1e-30, capped at nb (complete series)
    {
        const int D = f.nsec * f.K;
        std::vector<long double> Pw((size_t)D * D), Nx((size_t)D * D);
        const double *Mf = t.blob.data() + t.off_Mf;
        for (int i = 0; i < D * D; ++i) Pw[i] = Mf[i];
        int terms = 1;
        for (; terms < p.nb; ++terms) {
            long double mx = 0;
            for (int i = 0; i < D * D; ++i) mx = std::fmax(mx, std::fabs(Pw[i]));
            if (mx < 1e-30L) break;
            for (int r = 0; r < D; ++r)
                for (int c = 0; c < D; ++c) {
                    long double acc = 0;
                    for (int k = 0; k < D; ++k) acc += Pw[r * D + k] * (long double)Mf[k * D + c];
                    Nx[r * D + c] = acc;
                }
            Pw = Nx;
        }
        p.carry_terms = terms;
    }
    return t;
}

}  // namespace tdm
