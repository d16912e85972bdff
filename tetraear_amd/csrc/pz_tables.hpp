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

// L = samples per lane (a multiple of out_stride whenever S > 0 outputs per lane are tabulated), S = outputs per
// lane of the in-lane tables.
// in_scale / in_offset: the kernel runs on u with x = in_scale*u - in_offset (raw-integer kernel; 1 and 0 otherwise):
// every output-side coefficient carries in_scale, and off_yc holds in_offset * H(1)^2.
inline ZpHostTables build_pz_tables(const double (*sos)[6], int nsec, int64_t n, int edge, int L, int S,
                                    int64_t n_out, int out_stride, double in_scale = 1.0, double in_offset = 0.0)
{
    using namespace detail;
    ZpHostTables t;
    std::memset(&t.p, 0, sizeof(t.p));
    ZpParams &p = t.p;
    const int NP = nsec, D = 2 * nsec;
    p.nsec = nsec;
    p.K = 2;
    p.pform = 1;
    for (int s = 0; s < nsec; ++s) {
        p.b[s][0] = 1; p.b[s][1] = 2; p.b[s][2] = 1;
        for (int k = 0; k < 3; ++k) p.a[s][k] = sos[s][3 + k];
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
    p.out_stride = out_stride;
    const int qs = out_stride, len_last = p.len_last;
    PzDesign dz = design_pz(sos, nsec);
    ldbl h1 = 1;   // H(1) = prod g_s * 4 / (1 + a1 + a2)
    for (int s = 0; s < nsec; ++s) h1 *= (ldbl)sos[s][0] * 4 / (1 + dz.a1[s] + dz.a2[s]);
    for (int s = 0; s < nsec; ++s) { dz.b0[s] *= (ldbl)in_scale; dz.b1[s] *= (ldbl)in_scale; }
    dz.dx *= (ldbl)in_scale;
    const std::vector<ldbl> &a1 = dz.a1, &a2 = dz.a2, &b0 = dz.b0, &b1 = dz.b1;
    const std::vector<M2> &C = dz.C;

    std::vector<double> &blob = t.blob;
    auto reserve = [&](size_t cnt) { size_t o = blob.size(); blob.resize(o + cnt, 0.0); return o; };
    t.off_Mpow = reserve((size_t)nsec * kScanSteps * 4);
    t.off_zirh = reserve(1);
    // the last block's tables run kPzExtraRows outputs past its end (both responses continued by their own recurrence):
    // a consumer that walks whole groups of outputs (lp2_kernels.hpp) may then read rows of outputs the row does not have
    p.R_reg = (int32_t)((Bn + qs - 1) / qs);
    p.R_last = (len_last + qs - 1) / qs + kPzExtraRows;
    t.off_cflast = reserve((size_t)D);
    t.off_T1reg = reserve((size_t)qs * p.R_reg * D);
    t.off_T2reg = reserve((size_t)qs * p.R_reg * D);
    t.off_T1last = reserve((size_t)qs * p.R_last * D);
    t.off_T2last = reserve((size_t)qs * p.R_last * D);
    t.off_Mf = reserve((size_t)D * D);
    t.off_Mblast = reserve((size_t)D * D);
    t.off_Ureg = reserve((size_t)D * D);   // stays zero: the two banks do not couple
    t.off_Ulast = reserve((size_t)D * D);
    if (blob.size() & 1) reserve(1);   // 16-byte alignment of the block (its lane tables are read as pairs)
    t.off_pz = reserve((size_t)PzLayout::size(S));
    auto prow = [&](int m, int R) { return (size_t)(m % qs) * R + (size_t)(m / qs); };

    // ---- scan matrices C^(L 2^j); lane-distance matrices of the cross-row scan steps; block transitions
    for (int s = 0; s < NP; ++s) {
        for (int r = 0; r < 16; ++r) {
            const M2 m = m2pow(C[s], (long)L * (r + 1));
            double *o = &blob[t.off_pz + PzLayout::off_rowm(S) + ((size_t)r * PzLayout::kMaxPairs + s) * 4];
            o[0] = (double)m.a; o[1] = (double)m.b; o[2] = (double)m.c; o[3] = (double)m.d;
        }
        for (int j = 0; j < kScanSteps; ++j) {
            const M2 m = m2pow(C[s], (long)L << j);
            double *o = &blob[t.off_Mpow + ((size_t)s * kScanSteps + j) * 4];
            o[0] = (double)m.a; o[1] = (double)m.b; o[2] = (double)m.c; o[3] = (double)m.d;
        }
        const M2 mf = m2pow(C[s], (long)Bn), ml = m2pow(C[s], (long)len_last);
        auto put = [&](size_t off, const M2 &m) {
            blob[off + (2 * s) * D + 2 * s] = (double)m.a;
            blob[off + (2 * s) * D + 2 * s + 1] = (double)m.b;
            blob[off + (2 * s + 1) * D + 2 * s] = (double)m.c;
            blob[off + (2 * s + 1) * D + 2 * s + 1] = (double)m.d;
        };
        put(t.off_Mf, mf);
        put(t.off_Mblast, ml);
    }
    // ---- carry-response tables.  u_m = C^m e_kappa (first component):
    //   causal carry (w[-1], w[-2]) = e_kappa     -> output at offset m: b0 u_{m+1} + b1 u_m
    //   anticausal carry (w'[len], w'[len+1])     -> output at offset m: b0 u_{len-m} + b1 u_{len-m-1}
    const int64_t ext = (int64_t)kPzExtraRows * qs;
    std::vector<ldbl> ubuf((size_t)(Bn + 2 * ext + 4));
    ldbl *u = ubuf.data() + ext + 1;   // u[-ext-1 .. Bn+ext+1]
    for (int s = 0; s < NP; ++s)
        for (int kap = 0; kap < 2; ++kap) {
            ldbl v0 = kap == 0 ? 1 : 0, v1 = kap == 0 ? 0 : 1;   // (w[n], w[n-1]) pair; u_0 = first comp of e_kappa
            for (int64_t m = 0; m <= Bn + ext + 1; ++m) {
                u[m] = v0;
                const ldbl nv = -a1[s] * v0 - a2[s] * v1;
                v1 = v0;
                v0 = nv;
            }
            // the same sequence continued to negative indices: u[k-1] = -(u[k+1] + a1 u[k]) / a2
            for (int64_t m = 0; m >= -ext; --m) u[m - 1] = -(u[m + 1] + a1[s] * u[m]) / a2[s];
            const int k = 2 * s + kap;
            for (int v = 0; v < 2; ++v) {
                const int len = v ? len_last : (int)Bn;
                const int R = v ? p.R_last : p.R_reg;
                const size_t o1 = v ? t.off_T1last : t.off_T1reg, o2 = v ? t.off_T2last : t.off_T2reg;
                const int64_t mend = v ? len + ext : len;
                for (int64_t m = 0; m < mend; ++m) {
                    blob[o1 + prow((int)m, R) * D + k] = (double)(b0[s] * u[m + 1] + b1[s] * u[m]);
                    blob[o2 + prow((int)m, R) * D + k] = (double)(b0[s] * u[len - m] + b1[s] * u[len - m - 1]);
                }
            }
            // in-lane tables: outputs of a lane at local positions t*qs
            double *pz = &blob[t.off_pz];
            for (int tt = 0; tt < S; ++tt) {
                const int m = tt * qs;
                pz[PzLayout::off_zf + (s * S + tt) * 2 + kap] =
                    (double)(b0[s] * u[m + 1] + b1[s] * u[m]);
                pz[PzLayout::off_zb(S) + (s * S + tt) * 2 + kap] =
                    (double)(b0[s] * u[L - m] + b1[s] * u[L - m - 1]);
            }
        }
    {
        double *pz = &blob[t.off_pz];
        for (int s = 0; s < NP; ++s) {
            pz[PzLayout::off_a1 + s] = (double)a1[s];
            pz[PzLayout::off_a2 + s] = (double)a2[s];
            pz[PzLayout::off_b0 + s] = (double)b0[s];
            pz[PzLayout::off_b1 + s] = (double)b1[s];
            pz[PzLayout::off_g + s] = (double)(1 / (1 + a1[s] + a2[s]));
        }
        pz[PzLayout::off_dx] = (double)dz.dx;
        pz[PzLayout::off_yc] = (double)((ldbl)in_offset * h1 * h1);
    }
    {
        const std::vector<ldbl> &AE = dz.AE;
        // the kernel exports the inclusive scan state of the lane that holds the last position, i.e. the state
        // after that lane's zero-padded tail: undo the kinv padded steps (a few dozen at most, mildly expanding)
        const int kinv = L - 1 - (len_last - 1) % L;
        double *pz = &blob[t.off_pz];
        for (int r = 0; r < D; ++r) {
            for (int s = 0; s < NP; ++s) {
                const M2 ml = m2pow(C[s], (long)len_last);
                const M2 ci = m2pow(m2inv(C[s]), (long)kinv);
                const ldbl e0 = AE[(size_t)r * (D + 1) + 2 * s], e1 = AE[(size_t)r * (D + 1) + 2 * s + 1];
                pz[PzLayout::off_AG(S) + r * D + 2 * s] = (double)(e0 * ml.a + e1 * ml.c);
                pz[PzLayout::off_AG(S) + r * D + 2 * s + 1] = (double)(e0 * ml.b + e1 * ml.d);
                pz[PzLayout::off_AE(S) + r * D + 2 * s] = (double)(e0 * ci.a + e1 * ci.c);
                pz[PzLayout::off_AE(S) + r * D + 2 * s + 1] = (double)(e0 * ci.b + e1 * ci.d);
            }
            pz[PzLayout::off_wx(S) + r] = (double)AE[(size_t)r * (D + 1) + D];
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

}  // namespace tdm
