// Host-side tables of the PARALLEL-FORM zero-phase decimator (pz_kernels.hpp).
//
// scipy's sosfiltfilt (signal.decimate, processor.py:254) runs the cascade H forward and then
// backward; the composite operator is G(z) = H(z) H(1/z).  With H(z) = k (1+1/z)^N / prod(1 - p_i/z)
// (every section of these lowpass designs is g [1,2,1] / a) G has the partial-fraction expansion
//
//     G(z) = sum_i c_i / (1 - p_i/z)  +  sum_i c_i / (1 - p_i z)  +  (D0 - sum_i c_i),
//     c_i = r_i H(1/p_i),  r_i = residue of H at p_i,  D0 = k^2 / prod p_i,
//
// i.e. a causal and an anticausal bank of one-pole filters that BOTH run on the input itself.  A
// conjugate pole pair of the causal bank is the all-pole biquad w[n] = x[n] - a1 w[n-1] - a2 w[n-2]
// (a1, a2 are the section's own denominator, as designed) followed by b0 w[n] + b1 w[n-1] with
// b0 = 2 Re c, b1 = -2 Re(c conj p); the anticausal bank is its mirror image.  Two multiply-adds per
// real sample, pair and direction, and the two-tap output is formed only at the decimated positions.
// All |c_i| are O(1e-2) with states O(1/|1-p|^2): the terms of the sum are O(1), nothing cancels.
//
// scipy's edge recipe in these coordinates:
//   * forward start zi*ext[0]  == constant history ext[0]: w[-1] = w[-2] = ext[0] / (1 + a1 + a2);
//   * backward start zi*f[last] (f = forward output): the anticausal bank is started at the first
//     position after the extended signal in the state V = A_E * (causal state at the last position)
//     + wx * ext[last], derived in build() below.
// Every table is computed in long double from the double-precision section coefficients and rounded once.
#pragma once
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

#include "zp_common.hpp"
#include "zp_tables.hpp"

namespace tdm {

constexpr int kPzExtraRows = 16;

// layout of the parallel-form constant block ZpParams::pz (doubles); NP = pole pairs, S = outputs per lane
struct PzLayout {
    static constexpr int kMaxPairs = 4;
    static constexpr int off_a1 = 0, off_a2 = 4, off_b0 = 8, off_b1 = 12, off_g = 16, off_dx = 20;
    static constexpr int off_yc = 21;                 // response to the constant input offset left out by the raw-integer kernel (0 otherwise)
    static constexpr int off_zf = 24;                 // [NP][S][2] in-lane response to the causal start state
    TDM_HD static int off_zb(int S) { return off_zf + S * kMaxPairs * 2; }       // [NP][S][2] anticausal
    TDM_HD static int off_AG(int S) { return off_zb(S) + S * kMaxPairs * 2; }    // [D][D] V <- causal carry into the last block
    TDM_HD static int off_AE(int S) { return off_AG(S) + kMaxD * kMaxD; }        // [D][D] V <- exported lane state of the last block
    TDM_HD static int off_wx(int S) { return off_AE(S) + kMaxD * kMaxD; }        // [D]    V <- ext[last]
    TDM_HD static int off_rowm(int S) { return off_wx(S) + kMaxD; }              // [16][NP][4] C^(L (r+1)), r = position in a 16-lane row
    TDM_HD static int size(int S) { return off_rowm(S) + 16 * kMaxPairs * 4; }
};

namespace detail {
typedef long double ldbl;
typedef std::complex<long double> lcx;

struct M2 {
    ldbl a, b, c, d;  // [[a, b], [c, d]]
};
inline M2 m2mul(const M2 &x, const M2 &y)
{
    return {x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}
inline M2 m2pow(M2 base, long k)
{
    M2 r{1, 0, 0, 1};
    while (k > 0) {
        if (k & 1) r = m2mul(r, base);
        base = m2mul(base, base);
        k >>= 1;
    }
    return r;
}
inline M2 m2inv(const M2 &x)
{
    const ldbl det = x.a * x.d - x.b * x.c;
    return {x.d / det, -x.b / det, -x.c / det, x.a / det};
}
}  // namespace detail

// Partial-fraction data of H(z)H(1/z) for a cascade of sections g*[1,2,1]/[1,a1,a2] (rows_are_lp121), in long double.
struct PzDesign {
    int NP = 0;                                  // pole pairs (= sections)
    std::vector<detail::ldbl> a1, a2, b0, b1;    // all-pole recursion and two-tap output per pair
    std::vector<detail::M2> C;                   // one-step transition of (w[n-1], w[n-2])
    detail::ldbl dx = 0;                         // direct term
    std::vector<detail::ldbl> AE;                // [D][D+1]: anticausal start (w'[Ne], w'[Ne+1]) per pair from the causal state
                                                 // (w[Ne-1], w[Ne-2]) per pair (columns 0..D-1) and ext[Ne-1] (column D)
};

inline PzDesign design_pz(const double (*sos)[6], int nsec)
{
    using namespace detail;
    PzDesign d;
    const int NP = nsec, D = 2 * nsec, N = 2 * nsec;
    d.NP = NP;
    ldbl kgain = 1;
    std::vector<lcx> pl(N);   // pl[2s] = pole with Im > 0 of section s, pl[2s+1] its conjugate
    d.a1.resize(NP); d.a2.resize(NP); d.b0.resize(NP); d.b1.resize(NP); d.C.resize(NP);
    for (int s = 0; s < nsec; ++s) {
        kgain *= (ldbl)sos[s][0];
        d.a1[s] = sos[s][4];
        d.a2[s] = sos[s][5];
        const ldbl disc = 4 * d.a2[s] - d.a1[s] * d.a1[s];   // > 0: complex pair
        const lcx r(-d.a1[s] / 2, std::sqrt(disc) / 2);
        pl[2 * s] = r;
        pl[2 * s + 1] = std::conj(r);
        d.C[s] = M2{-d.a1[s], -d.a2[s], 1, 0};
    }
    const lcx one(1, 0);
    std::vector<lcx> res(N), cc(N), hinv(N);
    lcx prodp = one;
    for (int i = 0; i < N; ++i) prodp *= pl[i];
    for (int i = 0; i < N; ++i) {
        lcx num = kgain * std::pow(one + one / pl[i], N);
        lcx den = one;
        for (int j = 0; j < N; ++j)
            if (j != i) den *= (one - pl[j] / pl[i]);
        res[i] = num / den;
        lcx h = kgain * std::pow(one + pl[i], N);
        for (int j = 0; j < N; ++j) h /= (one - pl[j] * pl[i]);
        hinv[i] = h;   // H(1/p_i)
        cc[i] = res[i] * h;
    }
    const lcx d0 = kgain / prodp;             // H at z -> 0
    const lcx D0 = kgain * kgain / prodp;     // G at z -> infinity
    lcx csum(0, 0);
    for (int i = 0; i < N; ++i) csum += cc[i];
    d.dx = (D0 - csum).real();
    for (int s = 0; s < NP; ++s) {
        d.b0[s] = 2 * cc[2 * s].real();
        d.b1[s] = -2 * (cc[2 * s] * std::conj(pl[2 * s])).real();
    }
    // ---- edge map.  alpha_i = inclusive causal modal state at the last position: for the pole with Im > 0
    // of pair s, alpha = w[last] - conj(p) w[last-1].  F = d0 x_last + sum_j r_j alpha_j is the forward
    // output there; scipy continues it as a constant, which is the anticausal modal start
    //   V_i = (1/H(1/p_i)) [ F/(1-p_i) - sum_j r_j p_j alpha_j / (1 - p_i p_j) ]      (at the first position past the end)
    // and in biquad coordinates (w'[Ne], w'[Ne+1]) = (Im(p V)/Im p, Im V / Im p).
    d.AE.assign((size_t)D * (D + 1), 0);
    for (int col = 0; col <= D; ++col) {
        std::vector<lcx> alpha(N, lcx(0, 0));
        ldbl xl = 0;
        if (col < D) {
            const int s = col / 2;
            const lcx al = (col % 2 == 0) ? one : -std::conj(pl[2 * s]);
            alpha[2 * s] = al;
            alpha[2 * s + 1] = std::conj(al);
        } else {
            xl = 1;
        }
        lcx F = d0 * xl;
        for (int j = 0; j < N; ++j) F += res[j] * alpha[j];
        for (int s = 0; s < NP; ++s) {
            const lcx pi_ = pl[2 * s];
            lcx acc = F / (one - pi_);
            for (int j = 0; j < N; ++j) acc -= res[j] * pl[j] * alpha[j] / (one - pi_ * pl[j]);
            const lcx V = acc / hinv[2 * s];
            d.AE[(size_t)(2 * s) * (D + 1) + col] = (pi_ * V).imag() / pi_.imag();
            d.AE[(size_t)(2 * s + 1) * (D + 1) + col] = V.imag() / pi_.imag();
        }
    }
    return d;
}

// ---- tables.  L = samples per lane (a multiple of out_stride whenever S > 0 outputs per lane are tabulated), S =
// outputs per lane of the in-lane tables.
// in_scale / in_offset: the kernel runs on u with x = in_scale*u - in_offset (raw-integer kernel; 1 and 0 otherwise):
// every output-side coefficient carries in_scale, and off_yc holds in_offset * H(1)^2.  in_bias: the kernel's samples are
// u + in_bias (pz_kernels.hpp PzRawBias: the one-instruction byte -> fp64 conversion), i.e. x = in_scale*(u + in_bias) -
// (in_offset + in_scale*in_bias): the constant that off_yc removes grows by in_scale*in_bias, formed in long double.
//
// Everything that does not depend on the row length n is built once (PzShared: the design, the unit-state response
// sequences in long double, the scan / transition matrices and the full blocks' carry tables); the tables of a given
// length (last block's carry tables, edge maps, geometry) come from it in a fraction of the time, so that one plan
// serves ragged read sizes (scanner.py:347, rtl_auto_capture.py:182) without rebuilding or re-uploading the large tables.
struct PzShared : ZpSharedTables {
    int nsec = 0, edge = 0, L = 0, S = 0, qs = 1;
    double in_offset = 0;
    double sos_a[kMaxSec][3];
    PzDesign dz;                       // b0, b1, dx already scaled by in_scale
    detail::ldbl h1 = 1;               // H(1)
    std::vector<detail::ldbl> ubuf;    // [D][ulen] unit-state responses u[-ext-1 .. Bn+ext+1]
    size_t ulen = 0;
    int64_t ext = 0;
    size_t off_Mpow = 0, off_T1reg = 0, off_T2reg = 0, off_Mf = 0, off_Ureg = 0;
    std::vector<double> pz;            // the constant block with every length-independent entry filled in
    const detail::ldbl *useq(int k) const { return ubuf.data() + (size_t)k * ulen + ext + 1; }
};

namespace detail {
// carry-response tables of a block of `len` positions (+ `extra` positions past its end), phase-major rows of D doubles:
//   causal carry (w[-1], w[-2]) = e_kappa     -> output at offset m: b0 u_{m+1} + b1 u_m
//   anticausal carry (w'[len], w'[len+1])     -> output at offset m: b0 u_{len-m} + b1 u_{len-m-1}
inline void pz_fill_carry_tables(const PzShared &h, int len, int64_t extra, int R, double *o1, double *o2)
{
    const int D = 2 * h.nsec, qs = h.qs;
    const int64_t mend = len + extra;
    int ph = 0, r = 0;   // m = r*qs + ph
    for (int64_t m = 0; m < mend; ++m) {
        double *r1 = o1 + ((size_t)ph * R + r) * D, *r2 = o2 + ((size_t)ph * R + r) * D;
        for (int k = 0; k < D; ++k) {
            const ldbl *u = h.useq(k);
            const ldbl b0 = h.dz.b0[k >> 1], b1 = h.dz.b1[k >> 1];
            r1[k] = (double)(b0 * u[m + 1] + b1 * u[m]);
            r2[k] = (double)(b0 * u[len - m] + b1 * u[len - m - 1]);
        }
        if (++ph == qs) { ph = 0; ++r; }
    }
}
}  // namespace detail

// the bias of the raw-integer kernel's cu8 samples (pz_kernels.hpp PzRawBias: 4096 + u by one byte permute)
constexpr int kPzRawBiasCu8 = 4096;

inline std::shared_ptr<const PzShared> build_pz_shared(const double (*sos)[6], int nsec, int edge, int L, int S, int out_stride,
                                                       double in_scale = 1.0, double in_offset = 0.0, double in_bias = 0.0)
{
    using namespace detail;
    auto sh = std::make_shared<PzShared>();
    PzShared &h = *sh;
    const int NP = nsec, D = 2 * nsec, qs = out_stride;
    h.nsec = nsec; h.edge = edge; h.L = L; h.S = S; h.qs = qs; h.in_offset = in_offset;
    for (int s = 0; s < nsec; ++s)
        for (int k = 0; k < 3; ++k) h.sos_a[s][k] = sos[s][3 + k];
    h.dz = design_pz(sos, nsec);
    PzDesign &dz = h.dz;
    for (int s = 0; s < nsec; ++s) h.h1 *= (ldbl)sos[s][0] * 4 / (1 + dz.a1[s] + dz.a2[s]);   // H(1) = prod g_s * 4 / (1 + a1 + a2)
    for (int s = 0; s < nsec; ++s) { dz.b0[s] *= (ldbl)in_scale; dz.b1[s] *= (ldbl)in_scale; }
    dz.dx *= (ldbl)in_scale;
    const std::vector<ldbl> &a1 = dz.a1, &a2 = dz.a2, &b0 = dz.b0, &b1 = dz.b1;
    const std::vector<M2> &C = dz.C;
    const int64_t Bn = (int64_t)kWave * L;
    const int R_reg = (int)((Bn + qs - 1) / qs);

    size_t total = 0;
    auto reserve = [&](size_t cnt) { size_t o = total; total += cnt; return o; };
    h.off_Mpow = reserve((size_t)nsec * kScanSteps * 4);
    h.off_T1reg = reserve((size_t)qs * R_reg * D);
    h.off_T2reg = reserve((size_t)qs * R_reg * D);
    h.off_Mf = reserve((size_t)D * D);
    h.off_Ureg = reserve((size_t)D * D);   // stays zero: the two banks do not couple
    h.blob.assign(total, 0.0);
    std::vector<double> &blob = h.blob;
    h.pz.assign((size_t)PzLayout::size(S), 0.0);
    double *pz = h.pz.data();

    // ---- scan matrices C^(L 2^j); lane-distance matrices of the cross-row scan steps; block transition
    for (int s = 0; s < NP; ++s) {
        for (int r = 0; r < 16; ++r) {
            const M2 m = m2pow(C[s], (long)L * (r + 1));
            double *o = &pz[PzLayout::off_rowm(S) + ((size_t)r * PzLayout::kMaxPairs + s) * 4];
            o[0] = (double)m.a; o[1] = (double)m.b; o[2] = (double)m.c; o[3] = (double)m.d;
        }
        for (int j = 0; j < kScanSteps; ++j) {
            const M2 m = m2pow(C[s], (long)L << j);
            double *o = &blob[h.off_Mpow + ((size_t)s * kScanSteps + j) * 4];
            o[0] = (double)m.a; o[1] = (double)m.b; o[2] = (double)m.c; o[3] = (double)m.d;
        }
        const M2 mf = m2pow(C[s], (long)Bn);
        blob[h.off_Mf + (2 * s) * D + 2 * s] = (double)mf.a;
        blob[h.off_Mf + (2 * s) * D + 2 * s + 1] = (double)mf.b;
        blob[h.off_Mf + (2 * s + 1) * D + 2 * s] = (double)mf.c;
        blob[h.off_Mf + (2 * s + 1) * D + 2 * s + 1] = (double)mf.d;
    }
    // ---- unit-state responses u_m = C^m e_kappa (first component), all D of them
    h.ext = (int64_t)kPzExtraRows * qs;
    h.ulen = (size_t)(Bn + 2 * h.ext + 4);
    h.ubuf.resize(h.ulen * D);
    for (int s = 0; s < NP; ++s)
        for (int kap = 0; kap < 2; ++kap) {
            ldbl *u = h.ubuf.data() + (size_t)(2 * s + kap) * h.ulen + h.ext + 1;
            ldbl v0 = kap == 0 ? 1 : 0, v1 = kap == 0 ? 0 : 1;   // (w[n], w[n-1]) pair; u_0 = first comp of e_kappa
            for (int64_t m = 0; m <= Bn + h.ext + 1; ++m) {
                u[m] = v0;
                const ldbl nv = -a1[s] * v0 - a2[s] * v1;
                v1 = v0;
                v0 = nv;
            }
            // the same sequence continued to negative indices: u[k-1] = -(u[k+1] + a1 u[k]) / a2
            for (int64_t m = 0; m >= -h.ext; --m) u[m - 1] = -(u[m + 1] + a1[s] * u[m]) / a2[s];
            // in-lane tables: outputs of a lane at local positions t*qs
            for (int tt = 0; tt < S; ++tt) {
                const int m = tt * qs;
                pz[PzLayout::off_zf + (s * S + tt) * 2 + kap] = (double)(b0[s] * u[m + 1] + b1[s] * u[m]);
                pz[PzLayout::off_zb(S) + (s * S + tt) * 2 + kap] = (double)(b0[s] * u[L - m] + b1[s] * u[L - m - 1]);
            }
        }
    for (int s = 0; s < NP; ++s) {
        pz[PzLayout::off_a1 + s] = (double)a1[s];
        pz[PzLayout::off_a2 + s] = (double)a2[s];
        pz[PzLayout::off_b0 + s] = (double)b0[s];
        pz[PzLayout::off_b1 + s] = (double)b1[s];
        pz[PzLayout::off_g + s] = (double)(1 / (1 + a1[s] + a2[s]));
    }
    pz[PzLayout::off_dx] = (double)dz.dx;
    pz[PzLayout::off_yc] = (double)(((ldbl)in_offset + (ldbl)in_scale * (ldbl)in_bias) * h.h1 * h.h1);
    for (int r = 0; r < D; ++r) pz[PzLayout::off_wx(S) + r] = (double)dz.AE[(size_t)r * (D + 1) + D];
    pz_fill_carry_tables(h, (int)Bn, 0, R_reg, &blob[h.off_T1reg], &blob[h.off_T2reg]);   // the full blocks' carry tables
    return sh;
}

// tables of one row length from the shared part
inline ZpHostTables build_pz_tables(const std::shared_ptr<const PzShared> &sh, int64_t n, int64_t n_out)
{
    using namespace detail;
    const PzShared &h = *sh;
    ZpHostTables t;
    std::memset(&t.p, 0, sizeof(t.p));
    ZpParams &p = t.p;
    const int nsec = h.nsec, NP = nsec, D = 2 * nsec, L = h.L, S = h.S, edge = h.edge;
    p.nsec = nsec;
    p.K = 2;
    p.pform = 1;
    for (int s = 0; s < nsec; ++s) {
        p.b[s][0] = 1; p.b[s][1] = 2; p.b[s][2] = 1;
        for (int k = 0; k < 3; ++k) p.a[s][k] = h.sos_a[s][k];
    }
    p.in_gain = 1.0;
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
    p.out_stride = h.qs;
    const int qs = h.qs, len_last = p.len_last;
    const std::vector<M2> &C = h.dz.C;
    t.shared = sh;
    t.off_Mpow = h.off_Mpow; t.off_T1reg = h.off_T1reg; t.off_T2reg = h.off_T2reg; t.off_Mf = h.off_Mf; t.off_Ureg = h.off_Ureg;

    // the last block's tables run kPzExtraRows outputs past its end (both responses continued by their own recurrence):
    // a consumer that walks whole groups of outputs (lp2_kernels.hpp) may then read rows of outputs the row does not have
    p.R_reg = (int32_t)((Bn + qs - 1) / qs);
    p.R_last = (len_last + qs - 1) / qs + kPzExtraRows;
    size_t total = 0;
    auto reserve = [&](size_t cnt) { size_t o = total; total += cnt; return o; };
    t.off_zirh = reserve(1);
    t.off_cflast = reserve((size_t)D);
    reserve(1);   // (keeps the tables below on 16-byte boundaries)
    t.off_T1last = reserve((size_t)qs * p.R_last * D);
    t.off_T2last = reserve((size_t)qs * p.R_last * D);
    t.off_Mblast = reserve((size_t)D * D);
    t.off_Ulast = reserve((size_t)D * D);
    t.off_pz = reserve((size_t)PzLayout::size(S));   // 16-byte aligned: its lane tables are read as pairs
    t.blob.assign(total, 0.0);
    std::vector<double> &blob = t.blob;
    for (int s = 0; s < NP; ++s) {
        const M2 ml = m2pow(C[s], (long)len_last);
        blob[t.off_Mblast + (2 * s) * D + 2 * s] = (double)ml.a;
        blob[t.off_Mblast + (2 * s) * D + 2 * s + 1] = (double)ml.b;
        blob[t.off_Mblast + (2 * s + 1) * D + 2 * s] = (double)ml.c;
        blob[t.off_Mblast + (2 * s + 1) * D + 2 * s + 1] = (double)ml.d;
    }
    pz_fill_carry_tables(h, len_last, h.ext, p.R_last, &blob[t.off_T1last], &blob[t.off_T2last]);
    double *pz = &blob[t.off_pz];
    std::memcpy(pz, h.pz.data(), h.pz.size() * sizeof(double));
    {
        const std::vector<ldbl> &AE = h.dz.AE;
        // the kernel exports the inclusive scan state of the lane that holds the last position, i.e. the state
        // after that lane's zero-padded tail: undo the kinv padded steps (a few dozen at most, mildly expanding)
        const int kinv = L - 1 - (len_last - 1) % L;
        for (int s = 0; s < NP; ++s) {
            const M2 ml = m2pow(C[s], (long)len_last);
            const M2 ci = m2pow(m2inv(C[s]), (long)kinv);
            for (int r = 0; r < D; ++r) {
                const ldbl e0 = AE[(size_t)r * (D + 1) + 2 * s], e1 = AE[(size_t)r * (D + 1) + 2 * s + 1];
                pz[PzLayout::off_AG(S) + r * D + 2 * s] = (double)(e0 * ml.a + e1 * ml.c);
                pz[PzLayout::off_AG(S) + r * D + 2 * s + 1] = (double)(e0 * ml.b + e1 * ml.d);
                pz[PzLayout::off_AE(S) + r * D + 2 * s] = (double)(e0 * ci.a + e1 * ci.c);
                pz[PzLayout::off_AE(S) + r * D + 2 * s + 1] = (double)(e0 * ci.b + e1 * ci.d);
            }
        }
    }
    // carry series length: smallest t with max|Mf^t| < 1e-30, capped at nb
    {
        int terms = 1;
        std::vector<M2> pw(NP);
        for (int s = 0; s < NP; ++s) pw[s] = m2pow(C[s], (long)Bn);
        std::vector<M2> cur = pw;
        for (; terms < p.nb; ++terms) {
            ldbl mx = 0;
            for (int s = 0; s < NP; ++s)
                mx = std::fmax(mx, std::fmax(std::fmax(std::fabs(cur[s].a), std::fabs(cur[s].b)),
                                             std::fmax(std::fabs(cur[s].c), std::fabs(cur[s].d))));
            if (mx < 1e-30L) break;
            for (int s = 0; s < NP; ++s) cur[s] = m2mul(cur[s], pw[s]);
        }
        p.carry_terms = terms;
    }
    return t;
}

// both parts at once (single-length users: the stand-alone decimate entry, tests)
inline ZpHostTables build_pz_tables(const double (*sos)[6], int nsec, int64_t n, int edge, int L, int S,
                                    int64_t n_out, int out_stride, double in_scale = 1.0, double in_offset = 0.0)
{
    auto sh = build_pz_shared(sos, nsec, edge, L, S, out_stride, in_scale, in_offset);
    return build_pz_tables(sh, n, n_out);
}

}  // namespace tdm
