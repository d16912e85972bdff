// Low-rate stage of the reference chain in one kernel body:
//     decimator fix-up  ->  frequency_shift(samples, freq_offset)  ->  filter_signal(samples, 25000)  ->  |.|^2 per
//     timing phase (extract_symbols' mean powers)                (processor.py:260-264, 196-206)
// A workgroup of kLp2Waves wavefronts owns a chunk of one carrier's low-rate samples plus a halo on either side
// (lp2_tables.hpp); a lane owns kLp2La consecutive samples in registers.  The channel filter runs in parallel form
// (pz_tables.hpp: causal + anticausal all-pole bank per pole pair, both fed by the input):
//   pass 1  recurrences from zero state -> lane end states
//   scan    lanes of a wavefront (DPP rows), wavefronts of the workgroup (LDS) -> every lane's true start states
//   pass 2  recurrences from the true start states, two-tap outputs accumulated -> the filter output, final.
// Because the halo lets the filter's memory decay below 1e-21, no carry crosses a workgroup and no later kernel
// has to touch the output again: it is written once, phase-major (sample p + sps*k at [p][k]), so that the symbol
// gather of the finish stage reads one contiguous row.
//
// When the input comes from the parallel-form decimator its carry responses (y = y0 + T1.Gf + T2.Hb) are added here:
// for the 16 consecutive outputs of a lane they are, per pole pair and direction, a second-order recurrence at the
// decimated rate (x[k+1] = p1 x[k] - p2 x[k-1]), seeded by two table rows instead of one table row per sample.
//
//   Comm: tid(), sync() (workgroup barrier), stage() -> LDS of Lp2Lds::kStage doubles, small() -> Lp2Lds::kSmall doubles,
//   and the wavefront shuffles of pz_kernels.hpp (row_shr2, row_shl2, row_total_prev2, row_total_next2, wave_shr1, wave_shl1).
#pragma once
#include "lp2_tables.hpp"
#include "pz_kernels.hpp"

// TDM_LP2_TIMING builds: s_memtime per phase (thread 0 of every 16th workgroup) summed into g_lp2_dbg
#if defined(TDM_LP2_TIMING) && defined(__HIP_DEVICE_COMPILE__)
extern __device__ unsigned long long g_lp2_dbg[16];
#define LP2_T(i)                                                          \
    do {                                                                  \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
        if (cm.tid() == 0 && (chunk & 3) == 1) atomicAdd(&g_lp2_dbg[i], t_ - lp2_tprev_); \
        lp2_tprev_ = t_;                                                  \
    } while (0)
#define LP2_T0() unsigned long long lp2_tprev_ = __builtin_amdgcn_s_memtime()
#else
#define LP2_T(i)
#define LP2_T0()
#endif

namespace tdm {

// Staging area: one 16-byte slot per position of the workgroup's span, swizzled so that both access patterns are free of
// bank conflicts without padding: a lane's own run (slot = 16 lane + i: the low four bits become i ^ lane, sixteen
// neighbouring lanes hit sixteen different 16-byte bank groups) and the coalesced one (slot = base + lane with base a
// multiple of 16: XOR with a constant).  (One pad slot per 32 -- the cascade engine's layout, made for 8-sample lanes --
// left this kernel's 16-sample lanes with two-way conflicts: 37 % of its LDS cycles.)
TDM_HD int lp2_slot(int s) { return s ^ ((s >> 4) & 15); }
static_assert(kLp2La == 16, "lp2_slot swizzles inside runs of 16 slots = one lane's samples");

struct Lp2Lds {
    static constexpr int kSlots = kLp2Span;
    static constexpr int kStage = 2 * kSlots;
    // small area (doubles): [0,256) wave totals and the causal state at the end of the row, [256, 256 + 40*32) power partials of the groups
    static constexpr int oTot = 0, oPow = 256, kPowGroups = 39;
    static constexpr int kSmall = oPow + (kPowGroups + 1) * kMaxSps;
};

// ---- sample sources --------------------------------------------------------------------------------------------
struct Lp2SrcPlain {   // c128 rows already at the low rate (no decimation: k_convert applied freq_offset)
    const double *x;
    int64_t row_stride;
    static constexpr bool kFix = false;
    static constexpr double fs_out = 0.0;
    TDM_HD double foff(int) const { return 0.0; }
    TDM_HD const f64x2 *raw_row(int row) const { return (const f64x2 *)(x + (int64_t)row * row_stride * 2); }
    struct Pref {};
    TDM_HD void prefetch(const Lp2Params &, int, int64_t, Pref &) const {}
    TDM_HD void finish_lane(const Lp2Params &, const Pref &, double *, double *) const {}
};

struct Lp2SrcDec {     // block-local output of the parallel-form decimator + carries; freq_offset applied here
    ZpParams dec;
    const double *freq_offset;   // per row or null
    double fs_out;
    static constexpr bool kFix = true;
    TDM_HD double foff(int row) const { return freq_offset ? freq_offset[row] : 0.0; }
    TDM_HD const f64x2 *raw_row(int row) const { return (const f64x2 *)(dec.y0 + (int64_t)row * dec.n_out * 2); }
    // Operands of a lane's carry responses, requested at the very start of the kernel so that their latency
    // overlaps the staging of the samples: table rows r0, r0+1 (causal) and r0+La-2, r0+La-1 (anticausal) of the
    // lane's decimator block and that block's carries.
    struct Pref {
        f64x2 t1[2 * PzLayout::kMaxPairs], t2[2 * PzLayout::kMaxPairs], g[2 * PzLayout::kMaxPairs], h[2 * PzLayout::kMaxPairs];
    };
    TDM_HD void prefetch(const Lp2Params &P, int row, int64_t js, Pref &o) const
    {
        constexpr int La = kLp2La, ND = PzLayout::kMaxPairs, D = 2 * ND;
        const int Bn = kWave * dec.L;
        const int64_t pos = dec.k0L + js * dec.out_stride;
        const int b = (int)(pos / Bn);
        const int t = (int)((pos - (int64_t)b * Bn) / (dec.out_stride * La));   // group of La outputs inside the block
        const bool last = (b == dec.nb - 1);
        // the seed rows of all lanes of the chip come from one small table (a few KB, cache-resident): lanes of a
        // wavefront touch only `groups` distinct entries instead of 64 distinct rows of the full tables
        const f64x2 *sd = (const f64x2 *)(P.seeds + ((size_t)(last ? P.seed_groups : 0) + t) * kLp2SeedDoubles);
        const f64x2 *G = (const f64x2 *)(dec.Gf + ((int64_t)row * dec.nb + b) * D * 2);
        const f64x2 *Hh = (const f64x2 *)(dec.Hb + ((int64_t)row * dec.nb + b) * D * 2);
#pragma unroll
        for (int k = 0; k < 2 * ND; ++k) { o.t1[k] = sd[k]; o.t2[k] = sd[2 * ND + k]; o.g[k] = G[k]; o.h[k] = Hh[k]; }
    }
    // a lane's kLp2La consecutive samples, all inside one decimator block: carry responses by recurrence
    TDM_HD void finish_lane(const Lp2Params &P, const Pref &o, double *yr, double *yi) const
    {
        constexpr int La = kLp2La, ND = PzLayout::kMaxPairs;
#pragma unroll
        for (int s = 0; s < ND; ++s) {
            // pair s is components 2s, 2s+1 of a table row; g/h hold (re, im) of each carry component
            const f64x2 ta = o.t1[s], tb = o.t1[ND + s], ua = o.t2[s], ub = o.t2[ND + s];
            const f64x2 g0 = o.g[2 * s], g1 = o.g[2 * s + 1], h0 = o.h[2 * s], h1 = o.h[2 * s + 1];
            const double p1 = P.dec_p1[s], np2 = -P.dec_p2[s];
            double c0r = fma(ta.x, g0.x, ta.y * g1.x), c0i = fma(ta.x, g0.y, ta.y * g1.y);   // causal response at sample 0
            double c1r = fma(tb.x, g0.x, tb.y * g1.x), c1i = fma(tb.x, g0.y, tb.y * g1.y);   // ... 1
            yr[0] += c0r; yi[0] += c0i; yr[1] += c1r; yi[1] += c1i;
#pragma unroll
            for (int i = 2; i < La; ++i) {
                const double nr = fma(p1, c1r, np2 * c0r), ni = fma(p1, c1i, np2 * c0i);
                c0r = c1r; c0i = c1i; c1r = nr; c1i = ni;
                yr[i] += nr; yi[i] += ni;
            }
            double a0r = fma(ub.x, h0.x, ub.y * h1.x), a0i = fma(ub.x, h0.y, ub.y * h1.y);   // anticausal response at sample La-1
            double a1r = fma(ua.x, h0.x, ua.y * h1.x), a1i = fma(ua.x, h0.y, ua.y * h1.y);   // ... La-2
            yr[La - 1] += a0r; yi[La - 1] += a0i; yr[La - 2] += a1r; yi[La - 2] += a1i;
#pragma unroll
            for (int i = La - 3; i >= 0; --i) {
                const double nr = fma(p1, a1r, np2 * a0r), ni = fma(p1, a1i, np2 * a0i);
                a0r = a1r; a0i = a1i; a1r = nr; a1i = ni;
                yr[i] += nr; yi[i] += ni;
            }
        }
    }
};

template <class Src, class Comm>
TDM_HD void lp2_body(const Lp2Params &P, const Src &src, Comm &cm, int chunk, int row)
{
    constexpr int La = kLp2La, NP = kLp2Pairs;
    const int tid = cm.tid(), lane = tid & 63, wave = tid >> 6;
    const int64_t n = P.n;
    const int edge = P.edge;
    const int64_t jc = (int64_t)chunk * P.U - P.H - P.off;   // position of lane 0's first sample
    const int64_t js = jc + (int64_t)tid * La;
    f64x2 *stage = (f64x2 *)cm.stage();
    double *small = cm.small();

    // ---------------- input: coalesced through LDS, then each lane takes its La consecutive samples ----------------
    LP2_T0();
    const bool any_sig = (js + La > 0 && js < n);          // the lane holds at least one sample of the row
    typename Src::Pref pref;
    if (any_sig) src.prefetch(P, row, js, pref);
    {
        const f64x2 *rowp = src.raw_row(row);
        const int64_t jw = jc + (int64_t)wave * (kWave * La);
        // unconditional loads from a clamped 32-bit index (a load under a branch gets a wait of its own: sixteen serial
        // round trips), all of a lane's loads in flight at once, values outside the row zeroed afterwards
        const int jw32 = (int)jw + lane, n32 = (int)n;     // (|positions| < 2^31: tdm_plan_create bounds the chunk length)
        f64x2 v[La];
#pragma unroll
        for (int i = 0; i < La; ++i) {
            const int j = jw32 + i * kWave;
            const int jj = j < 0 ? 0 : (j >= n32 ? n32 - 1 : j);
#ifdef TDM_LP2_FAKE_LOADS   // experiment: every load hits the same cache-resident kilobytes (results are wrong, timing only)
            v[i] = src.raw_row(0)[jj & 1023];
#else
            v[i] = rowp[jj];
#endif
        }
#pragma unroll
        for (int i = 0; i < La; ++i) {
            const int j = jw32 + i * kWave;
            const bool ok = (j >= 0 && j < n32);
            stage[lp2_slot(wave * (kWave * La) + i * kWave + lane)] = f64x2{ok ? v[i].x : 0.0, ok ? v[i].y : 0.0};
        }
    }
    cm.sync();
    LP2_T(0);
    double yr[La], yi[La];
#pragma unroll
    for (int i = 0; i < La; ++i) {
        const f64x2 v = stage[lp2_slot(tid * La + i)];
        yr[i] = v.x;
        yi[i] = v.y;
    }
    // lanes that hold signal samples: carry responses + NCO (for a lane that is only partly inside the row the values at
    // positions outside it are meaningless here and are replaced by the odd extension below)
    const bool inside = (js >= 0 && js + La <= n);
    if (any_sig) {
        if (Src::kFix) {
            const double f = src.foff(row);
            NcoRunT<1> nco;
            if (f != 0.0) nco.init(js, f, src.fs_out);   // (out-of-line sincos first, while few registers are live)
            src.finish_lane(P, pref, yr, yi);
            if (f != 0.0) {
#pragma unroll
                for (int i = 0; i < La; ++i) {
                    double c = nco.ar, sn = nco.ai;
                    if (i > 0) nco.next(f, src.fs_out, c, sn);
                    const double a = yr[i], b = yi[i];
                    yr[i] = a * c - b * sn;
                    yi[i] = a * sn + b * c;
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < La; ++i) { yr[i] = 0; yi[i] = 0; }
    }
    // publish the finished samples; the odd extension (scipy odd_ext: 2 x[0] - x[-j], 2 x[n-1] - x[2n-2-j]) of the lanes
    // around the ends of the row reads them from there
#pragma unroll
    for (int i = 0; i < La; ++i) stage[lp2_slot(tid * La + i)] = f64x2{yr[i], yi[i]};
    cm.sync();
    LP2_T(1);
    auto Y = [&](int64_t j) { return stage[lp2_slot((int)(j - jc))]; };
    if (!inside) {
#pragma unroll
        for (int i = 0; i < La; ++i) {
            const int64_t j = js + i;
            if (j < 0 || j >= n) {
                f64x2 v{0.0, 0.0};
                if (j >= -(int64_t)edge && j < 0) {
                    const f64x2 a = Y(0), b = Y(-j);
                    v = f64x2{2 * a.x - b.x, 2 * a.y - b.y};
                } else if (j >= n && j < n + edge) {
                    const f64x2 a = Y(n - 1), b = Y(2 * n - 2 - j);
                    v = f64x2{2 * a.x - b.x, 2 * a.y - b.y};
                }
                yr[i] = v.x;
                yi[i] = v.y;
            }
        }
    }
    // the empty lanes next to the ends of the extended row carry the start states of the two banks (lp2_tables.hpp)
    const int64_t t_head = ((-(int64_t)edge - jc) >= 0 ? (-(int64_t)edge - jc) / La : -((jc + edge + La - 1) / La)) - 1;   // empty lane before position -edge
    const int64_t t_l1 = (n + edge - 1 - jc) / La;          // lane holding the last extended sample
    const int64_t t_tail = (n + edge - jc) / La + 1;        // empty lane after the lane holding position n + edge
    const bool has_head = (t_head >= 0 && t_head < kLp2Lanes);
    const bool has_tail = (n + edge - 1 - jc >= 0 && t_tail < kLp2Lanes && t_l1 >= 0);
    double e0r = 0, e0i = 0, xlr = 0, xli = 0;
    if (has_head && tid == t_head) {
        const f64x2 a = Y(0), b = Y(edge);
        e0r = 2 * a.x - b.x;
        e0i = 2 * a.y - b.y;
    }
    if (has_tail && tid == t_tail) {
        const f64x2 a = Y(n - 1), b = Y(n - 1 - edge);
        xlr = 2 * a.x - b.x;
        xli = 2 * a.y - b.y;
    }

    // (requested here, used by the scans: the latency overlaps pass 1)
    double lmf[NP][4], lmb[NP][4];   // C^(La (k+1)), k = lane's distance to the previous / next row of 16 lanes (scan)
    {
        const int r = lane & 15;
        const f64x2 *tf = (const f64x2 *)P.lane_m + (size_t)r * NP * 2;
        const f64x2 *tb = (const f64x2 *)P.lane_m + (size_t)(15 - r) * NP * 2;
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const f64x2 a0 = tf[s * 2], a1 = tf[s * 2 + 1], c0 = tb[s * 2], c1 = tb[s * 2 + 1];
            lmf[s][0] = a0.x; lmf[s][1] = a0.y; lmf[s][2] = a1.x; lmf[s][3] = a1.y;
            lmb[s][0] = c0.x; lmb[s][1] = c0.y; lmb[s][2] = c1.x; lmb[s][3] = c1.y;
        }
    }
    // ---------------- pass 1: recurrences from zero state, lane end states ----------------
    double zr[NP][2], zq[NP][2];   // causal end state (w[La-1], w[La-2]), re / im
    double ur[NP][2], uq[NP][2];   // anticausal end state (w'[0], w'[1])
#pragma unroll
    for (int s = 0; s < NP; ++s) {
        const double na1 = P.na1[s], na2 = P.na2[s];
        double f1r = 0, f2r = 0, f1q = 0, f2q = 0, a1r = 0, a2r = 0, a1q = 0, a2q = 0;
#pragma unroll
        for (int i = 0; i < La; ++i) {
            const int ib = La - 1 - i;
            const double wr = fma(na1, f1r, fma(na2, f2r, yr[i])), wq = fma(na1, f1q, fma(na2, f2q, yi[i]));
            f2r = f1r; f1r = wr; f2q = f1q; f1q = wq;
            const double vr = fma(na1, a1r, fma(na2, a2r, yr[ib])), vq = fma(na1, a1q, fma(na2, a2q, yi[ib]));
            a2r = a1r; a1r = vr; a2q = a1q; a1q = vq;
        }
        zr[s][0] = f1r; zr[s][1] = f2r; zq[s][0] = f1q; zq[s][1] = f2q;
        ur[s][0] = a1r; ur[s][1] = a2r; uq[s][0] = a1q; uq[s][1] = a2q;
    }
    if (has_head && tid == t_head) {
        const double *hv = P.cst + Lp2Cst::head_v;
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            zr[s][0] = hv[s * 2] * e0r; zr[s][1] = hv[s * 2 + 1] * e0r;
            zq[s][0] = hv[s * 2] * e0i; zq[s][1] = hv[s * 2 + 1] * e0i;
        }
    }
    LP2_T(2);
    // ---------------- scans.  dir 0: causal (inclusive from the left); dir 1: anticausal (from the right) ----------------
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        double(*vr)[2] = dir == 0 ? zr : ur;
        double(*vq)[2] = dir == 0 ? zq : uq;
        if (dir == 1) {
            // the anticausal bank starts from the causal bank's state at the end of the row (scipy: zi * forward[last])
            if (has_tail && tid == t_tail) {
                const double *cst = small + Lp2Lds::oTot + 128;   // causal state of lane t_l1, left there below
                const double *tm = P.cst + Lp2Cst::tail_m, *tx = P.cst + Lp2Cst::tail_x;
#pragma unroll
                for (int r = 0; r < kLp2D; ++r) {
                    double ar = tx[r] * xlr, ai = tx[r] * xli;
#pragma unroll
                    for (int k = 0; k < kLp2D; ++k) { ar = fma(tm[r * kLp2D + k], cst[2 * k], ar); ai = fma(tm[r * kLp2D + k], cst[2 * k + 1], ai); }
                    vr[r / 2][r % 2] = ar;
                    vq[r / 2][r % 2] = ai;
                }
            }
        }
        // inside a wavefront: four steps inside each row of 16 lanes, then the row totals are passed on twice
        const double *Msc = P.cst + Lp2Cst::Mscan;
        TDM_OPAQUE_SPTR(Msc);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double jr[NP][2], jq[NP][2];
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                if (dir == 0) cm.template row_shr2<2>(vr[s], vq[s], jr[s], jq[s], 1 << j);
                else cm.template row_shl2<2>(vr[s], vq[s], jr[s], jq[s], 1 << j);
            }
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                const auto M = TDM_CPTR(Msc + (s * 4 + j) * 4);
                const double o0 = vr[s][0], o1 = vr[s][1], q0 = vq[s][0], q1 = vq[s][1];
                vr[s][0] = fma(M[0], jr[s][0], fma(M[1], jr[s][1], o0));
                vr[s][1] = fma(M[2], jr[s][0], fma(M[3], jr[s][1], o1));
                vq[s][0] = fma(M[0], jq[s][0], fma(M[1], jq[s][1], q0));
                vq[s][1] = fma(M[2], jq[s][0], fma(M[3], jq[s][1], q1));
            }
        }
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            double jr[NP][2], jq[NP][2];
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                if (dir == 0) cm.template row_total_prev2<2>(vr[s], vq[s], jr[s], jq[s], step);
                else cm.template row_total_next2<2>(vr[s], vq[s], jr[s], jq[s], step);
            }
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                if (step == 1) {
                    const auto M = TDM_CPTR(P.cst + Lp2Cst::Mrow + s * 4);
                    const bool far = dir == 0 ? lane >= 48 : lane < 16;
                    const double a0 = fma(M[0], jr[s][0], M[1] * jr[s][1]), a1 = fma(M[2], jr[s][0], M[3] * jr[s][1]);
                    const double b0 = fma(M[0], jq[s][0], M[1] * jq[s][1]), b1 = fma(M[2], jq[s][0], M[3] * jq[s][1]);
                    jr[s][0] = far ? a0 : jr[s][0]; jr[s][1] = far ? a1 : jr[s][1];
                    jq[s][0] = far ? b0 : jq[s][0]; jq[s][1] = far ? b1 : jq[s][1];
                }
                const double *M = dir == 0 ? lmf[s] : lmb[s];
                const double o0 = vr[s][0], o1 = vr[s][1], q0 = vq[s][0], q1 = vq[s][1];
                vr[s][0] = fma(M[0], jr[s][0], fma(M[1], jr[s][1], o0));
                vr[s][1] = fma(M[2], jr[s][0], fma(M[3], jr[s][1], o1));
                vq[s][0] = fma(M[0], jq[s][0], fma(M[1], jq[s][1], q0));
                vq[s][1] = fma(M[2], jq[s][0], fma(M[3], jq[s][1], q1));
            }
        }
        // across the wavefronts of the workgroup: totals through LDS, each wavefront forms the prefix that enters it
        double *tot = small + Lp2Lds::oTot;   // [wave][pair][4]
        cm.sync();                            // (previous use of the area)
        if (lane == (dir == 0 ? kWave - 1 : 0)) {
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                double *o = tot + (wave * NP + s) * 4;
                o[0] = vr[s][0]; o[1] = vr[s][1]; o[2] = vq[s][0]; o[3] = vq[s][1];
            }
        }
        cm.sync();
        double pr[NP][2], pq[NP][2];   // state entering this wavefront
#pragma unroll
        for (int s = 0; s < NP; ++s) { pr[s][0] = 0; pr[s][1] = 0; pq[s][0] = 0; pq[s][1] = 0; }
#pragma unroll 1
        for (int k = 1; k < kLp2Waves; ++k) {
            // wavefronts in order of increasing distance ... processed from the farthest: P <- C^(64 La) P + T_v
            const int v = dir == 0 ? k - 1 : kLp2Waves - k;          // causal: v = 0 .. wave-1 ; anticausal: v = W-1 .. wave+1
            const bool use = dir == 0 ? (v < wave) : (v > wave);
            if (!use) continue;
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                const auto M = TDM_CPTR(P.cst + Lp2Cst::Mwave + s * 4);
                const double *t = tot + (v * NP + s) * 4;
                const double a0 = fma(M[0], pr[s][0], fma(M[1], pr[s][1], t[0]));
                const double a1 = fma(M[2], pr[s][0], fma(M[3], pr[s][1], t[1]));
                const double b0 = fma(M[0], pq[s][0], fma(M[1], pq[s][1], t[2]));
                const double b1 = fma(M[2], pq[s][0], fma(M[3], pq[s][1], t[3]));
                pr[s][0] = a0; pr[s][1] = a1; pq[s][0] = b0; pq[s][1] = b1;
            }
        }
        {
            // every lane: true inclusive state = in-wave value + C^(La (distance in lanes)) * prefix
            const int kk = dir == 0 ? lane : kWave - 1 - lane;
            const f64x2 *tm = (const f64x2 *)P.lane_m + (size_t)kk * NP * 2;
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                const f64x2 m0 = tm[s * 2], m1 = tm[s * 2 + 1];
                vr[s][0] = fma(m0.x, pr[s][0], fma(m0.y, pr[s][1], vr[s][0]));
                vr[s][1] = fma(m1.x, pr[s][0], fma(m1.y, pr[s][1], vr[s][1]));
                vq[s][0] = fma(m0.x, pq[s][0], fma(m0.y, pq[s][1], vq[s][0]));
                vq[s][1] = fma(m1.x, pq[s][0], fma(m1.y, pq[s][1], vq[s][1]));
            }
        }
        if (dir == 0 && has_tail && tid == t_l1) {
            double *cst = small + Lp2Lds::oTot + 128;
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                cst[(2 * s) * 2] = vr[s][0]; cst[(2 * s) * 2 + 1] = vq[s][0];
                cst[(2 * s + 1) * 2] = vr[s][1]; cst[(2 * s + 1) * 2 + 1] = vq[s][1];
            }
        }
        // start state of a lane = inclusive state of its neighbour (the wavefront's prefix at the wavefront's edge)
        double sr[NP][2], sq[NP][2];
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            if (dir == 0) cm.template wave_shr1<2>(vr[s], vq[s], sr[s], sq[s]);
            else cm.template wave_shl1<2>(vr[s], vq[s], sr[s], sq[s]);
            const bool at_edge = dir == 0 ? lane == 0 : lane == kWave - 1;
            vr[s][0] = at_edge ? pr[s][0] : sr[s][0]; vr[s][1] = at_edge ? pr[s][1] : sr[s][1];
            vq[s][0] = at_edge ? pq[s][0] : sq[s][0]; vq[s][1] = at_edge ? pq[s][1] : sq[s][1];
        }
        if (dir == 0) cm.sync();   // (cst visible before the anticausal start is formed)
    }
    LP2_T(3);
    // ---------------- pass 2: recurrences from the true start states, outputs accumulated ----------------
    double or_[La], oi[La];
    {
        const double dx = P.dx;
#pragma unroll
        for (int i = 0; i < La; ++i) { or_[i] = dx * yr[i]; oi[i] = dx * yi[i]; }
    }
#pragma unroll
    for (int s = 0; s < NP; ++s) {
        const double na1 = P.na1[s], na2 = P.na2[s], b0 = P.b0[s], b1 = P.b1[s];
        double f1r = zr[s][0], f2r = zr[s][1], f1q = zq[s][0], f2q = zq[s][1];
        double a1r = ur[s][0], a2r = ur[s][1], a1q = uq[s][0], a2q = uq[s][1];
#pragma unroll
        for (int i = 0; i < La; ++i) {
            const int ib = La - 1 - i;
            const double wr = fma(na1, f1r, fma(na2, f2r, yr[i])), wq = fma(na1, f1q, fma(na2, f2q, yi[i]));
            or_[i] = fma(b0, wr, fma(b1, f1r, or_[i]));
            oi[i] = fma(b0, wq, fma(b1, f1q, oi[i]));
            f2r = f1r; f1r = wr; f2q = f1q; f1q = wq;
            const double vr_ = fma(na1, a1r, fma(na2, a2r, yr[ib])), vq_ = fma(na1, a1q, fma(na2, a2q, yi[ib]));
            or_[ib] = fma(b0, vr_, fma(b1, a1r, or_[ib]));
            oi[ib] = fma(b0, vq_, fma(b1, a1q, oi[ib]));
            a2r = a1r; a1r = vr_; a2q = a1q; a1q = vq_;
        }
    }
    LP2_T(4);
    // ---------------- output: through LDS; one thread per (timing phase, group) stores its phase's samples and sums their powers ----------------
    cm.sync();   // (all lanes have taken their input out of the staging area)
#pragma unroll
    for (int i = 0; i < La; ++i) stage[lp2_slot(tid * La + i)] = f64x2{or_[i], oi[i]};
    cm.sync();
    LP2_T(5);
    const int64_t j_lo = (int64_t)chunk * P.U - P.off > 0 ? (int64_t)chunk * P.U - P.off : 0;
    int64_t j_hi = (int64_t)(chunk + 1) * P.U - P.off;
    if (j_hi > n) j_hi = n;
    const int sps = P.sps;
    if (sps > 0) {
        f64x2 *zt = (f64x2 *)P.zt + (int64_t)row * sps * P.zt_k;
        constexpr int NG = Lp2Lds::kPowGroups;
        const int ph = tid / NG, g = tid % NG;     // 512 threads: phases 0..12 x 39 groups (sps <= 13), else fewer groups per phase
        const int ngrp = kLp2Lanes / sps < NG ? kLp2Lanes / sps : NG;
        const int p2 = tid / ngrp, g2 = tid % ngrp;
        (void)ph; (void)g;
        double acc = 0;
        if (p2 < sps) {
            const int64_t np_ = (n - p2) / sps;                 // samples phase p owns: j = p + k*sps, k < np_
            const int64_t lim = p2 + np_ * (int64_t)sps;        // first j NOT owned (extract_symbols, processor.py:199-203)
            // first k with j = p2 + k*sps >= j_lo
            int64_t k0 = j_lo <= p2 ? 0 : (j_lo - p2 + sps - 1) / sps;
            for (int64_t k = k0 + g2; p2 + k * sps < j_hi; k += ngrp) {
                const int64_t j = p2 + k * sps;
                const f64x2 v = stage[lp2_slot((int)(j - jc))];
                zt[(int64_t)p2 * P.zt_k + k] = v;
                if (j < lim) acc += fma(v.x, v.x, v.y * v.y);
            }
            small[Lp2Lds::oPow + g2 * kMaxSps + p2] = acc;
        }
        cm.sync();
        if (tid < kMaxSps) {
            double t = 0;
            if (tid < sps)
                for (int gg = 0; gg < ngrp; ++gg) t += small[Lp2Lds::oPow + gg * kMaxSps + tid];
            P.partials[((int64_t)row * P.n_chunks + chunk) * kMaxSps + tid] = t;
        }
    } else {
        f64x2 *z = (f64x2 *)P.zt + (int64_t)row * P.zt_k;
        for (int64_t j = j_lo + tid; j < j_hi; j += kLp2Lanes) z[j] = stage[lp2_slot((int)(j - jc))];
    }
    LP2_T(6);
}

}  // namespace tdm
