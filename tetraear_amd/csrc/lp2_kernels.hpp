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
#include <type_traits>

#include "lp2_tables.hpp"
#include "pz_kernels.hpp"

// (TDM_SCHED_FENCE: pz_kernels.hpp)

namespace tdm {

// Staging area: one 16-byte slot per position of the workgroup's span, swizzled so that both access patterns are free of
// bank conflicts without padding: a lane's own run (slot = 16 lane + i: the low four bits become i ^ lane, sixteen
// neighbouring lanes hit sixteen different 16-byte bank groups) and the coalesced one (slot = base + lane with base a
// multiple of 16: XOR with a constant).  (One pad slot per 32 -- the cascade engine's layout, made for 8-sample lanes --
// left this kernel's 16-sample lanes with two-way conflicts: 37 % of its LDS cycles.)
TDM_HD int lp2_slot(int s) { return s ^ ((s >> 4) & (kLp2La - 1)); }
static_assert(kLp2La == 16, "lp2_slot swizzles inside a lane's run of sixteen slots");
static_assert(kLp2Lanes <= (1 << kLp2GBits), "item word layout");

struct Lp2Lds {
    static constexpr int kSlots = kLp2Span;
    static constexpr int kStage = 2 * kSlots;
    // small area (doubles): [0,256) wave totals and the causal state at the end of the row, [256, 256 + 40*32) power partials of the groups
    static constexpr int oTot = 0, oPow = 2 * (kLp2Lanes / 16) * kLp2Pairs * 4, kPowGroups = 39;
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
    template <class Comm>
    TDM_HD void prefetch_words(const Lp2Params &, int, Comm &, Pref &) const {}
    TDM_HD void prefetch_operands(const Lp2Params &, int, Pref &) const {}
    template <class Comm>
    TDM_HD void fix_phase(const Lp2Params &, int, int, f64x2 *, Comm &, const Pref &) const {}
};

struct Lp2SrcDec {     // block-local output of the parallel-form decimator + carries; freq_offset applied here
    ZpParams dec;
    const double *freq_offset;   // per row or null
    double fs_out;
    static constexpr bool kFix = true;
    TDM_HD double foff(int row) const { return freq_offset ? freq_offset[row] : 0.0; }
    TDM_HD const f64x2 *raw_row(int row) const { return (const f64x2 *)(dec.y0 + (int64_t)row * dec.n_out * 2); }
    // The decimator's carry responses (y = y0 + T1.Gf + T2.Hb), added to the staged samples in place.  For the La
    // consecutive outputs of a group they are, per pole pair and direction, a second-order recurrence at the decimated
    // rate seeded by two table rows.  They decay from the block's ends, so only the groups near a block boundary need them,
    // and only for the slowly decaying pairs: the plan lists the (group, direction, pairs) items of every chunk, costliest
    // first (Lp2Params::items), and the workgroup's threads work the list off -- thread k takes item k, whatever group
    // that is, so that a wavefront's 64 items cost the same.  (Round 2 ran all four pairs in both directions over every
    // lane's own samples: 768 of the kernel's 4076 vector instructions per lane, for responses that are below 1e-30 of the
    // signal in three groups out of four.)
    // operands of a thread's first item, requested before the samples are staged so that their latency overlaps the
    // staging: the item's two words, its seed rows and its block's carries
    struct Pref {
        int w0 = 0, w1 = 0, mine = 0;
        f64x2 sd[2 * PzLayout::kMaxPairs], cy[2 * PzLayout::kMaxPairs];
    };
    TDM_HD void item_operands(const Lp2Params &P, int row, int w0, int w1, f64x2 (&sdv)[2 * PzLayout::kMaxPairs], f64x2 (&cyv)[2 * PzLayout::kMaxPairs]) const
    {
        constexpr int ND = PzLayout::kMaxPairs, D = 2 * ND;
        const int dir = (w0 >> kLp2GBits) & 1, b = w1 >> 8, t = w1 & 255;
        const bool last = (b == dec.nb - 1);
        // seed rows of the group: outputs La t, La t + 1 (causal), La t + La - 2, La t + La - 1 (anticausal); the block's carries
        const f64x2 *sd = (const f64x2 *)(P.seeds + ((size_t)(last ? P.seed_groups : 0) + t) * kLp2SeedDoubles) + (dir ? 2 * ND : 0);   // (rows 16t, 16t+1 or 16t+14, 16t+15)
#pragma unroll
        for (int k = 0; k < 2 * ND; ++k) sdv[k] = sd[k];
        const f64x2 *cy = (const f64x2 *)((dir ? dec.Hb : dec.Gf) + ((int64_t)row * dec.nb + b) * D * 2);
#pragma unroll
        for (int k = 0; k < 2 * ND; ++k) cyv[k] = cy[k];
    }
    // Two steps around the issue of the sample loads: the item's two words first (unconditional: a chunk's list is padded
    // to one item per thread), so that they are the OLDEST loads in flight when the samples' sixteen follow; then, once
    // the words are back (the samples still on their way), the seed rows and carries they name.  (As one step ahead of
    // the sample loads the dependent pair of round trips delayed the staging of every workgroup.)
    template <class Comm>
    TDM_HD void prefetch_words(const Lp2Params &P, int chunk, Comm &cm, Pref &o) const
    {
        const int32_t *it = P.items + (size_t)chunk * P.items_stride;
        const int k = cm.tid();
        o.w0 = it[2 + 2 * k];
        o.w1 = it[3 + 2 * k];
        o.mine = k < it[0];
    }
    TDM_HD void prefetch_operands(const Lp2Params &P, int row, Pref &o) const
    {
        if (o.mine) item_operands(P, row, o.w0, o.w1, o.sd, o.cy);
    }
    // The decimator's carry responses (y = y0 + T1.Gf + T2.Hb), added to the staged samples in place.  For the La
    // consecutive outputs of a group they are, per pole pair and direction, a second-order recurrence at the decimated
    // rate seeded by two table rows.  They decay from the block's ends, so only the groups near a block boundary need them,
    // and only for the slowly decaying pairs: the plan lists the (group, direction, pairs) items of every chunk, costliest
    // first (Lp2Params::items), and the workgroup's threads work the list off -- thread k takes item k, whatever group
    // that is, so that a wavefront's 64 items cost the same.  (Round 2 ran all four pairs in both directions over every
    // lane's own samples: 768 of the kernel's 4076 vector instructions per lane, for responses that are below 1e-30 of the
    // signal in three groups out of four.)
    // Header of a chunk's list: it[0] = items of the first pass, it[1] = items of a second pass behind a barrier (only when
    // some group has a causal AND an anticausal item, i.e. blocks shorter than twice the reach: the two would update the
    // same samples), 0 otherwise.
    // one item: the group's samples leave LDS in the order the recurrence visits them (anticausal: last output first),
    // take the responses of the pairs in `mask` and go back
    TDM_HD void run_item(const Lp2Params &P, f64x2 *stage, int w0, const f64x2 (&sdv)[2 * PzLayout::kMaxPairs],
                         const f64x2 (&cyv)[2 * PzLayout::kMaxPairs]) const
    {
        constexpr int La = kLp2La, ND = PzLayout::kMaxPairs;
        const int g = w0 & ((1 << kLp2GBits) - 1), dir = (w0 >> kLp2GBits) & 1, mask = (w0 >> (kLp2GBits + 1)) & 15;
        const int base = g * La, sw = ((g * La) >> 4) & (La - 1);   // lp2_slot(La g + i) = La g + (i ^ sw)
        const int flip = dir ? La - 1 : 0;             // (La is a power of two: La - 1 - i == i ^ (La - 1))
        double vr[La], vi[La];
#pragma unroll
        for (int i = 0; i < La; ++i) {
            const f64x2 v = stage[base + ((i ^ flip) ^ sw)];
            vr[i] = v.x;
            vi[i] = v.y;
        }
#pragma unroll
        for (int s = 0; s < ND; ++s) {
            if (!((mask >> s) & 1)) continue;
            // pair s is components 2s, 2s+1 of a table row; cy holds (re, im) of each carry component.  Anticausal:
            // the rows arrive as (16t+14, 16t+15) and are visited last output first
            const f64x2 r0 = sdv[s], r1 = sdv[ND + s];
            const f64x2 ta = dir ? r1 : r0, tb = dir ? r0 : r1;
            const f64x2 g0 = cyv[2 * s], g1 = cyv[2 * s + 1];
            const double p1 = P.dec_p1[s], np2 = -P.dec_p2[s];
            double c0r = fma(ta.x, g0.x, ta.y * g1.x), c0i = fma(ta.x, g0.y, ta.y * g1.y);   // response at the first output visited
            double c1r = fma(tb.x, g0.x, tb.y * g1.x), c1i = fma(tb.x, g0.y, tb.y * g1.y);   // ... the second
            vr[0] += c0r; vi[0] += c0i; vr[1] += c1r; vi[1] += c1i;
#pragma unroll
            for (int i = 2; i < La; ++i) {
                const double nr = fma(p1, c1r, np2 * c0r), ni = fma(p1, c1i, np2 * c0i);
                c0r = c1r; c0i = c1i; c1r = nr; c1i = ni;
                vr[i] += nr; vi[i] += ni;
            }
        }
#pragma unroll
        for (int i = 0; i < La; ++i) stage[base + ((i ^ flip) ^ sw)] = f64x2{vr[i], vi[i]};
    }
    // The decimator's carry responses (y = y0 + T1.Gf + T2.Hb), added to the staged samples in place.  For the La
    // consecutive outputs of a group they are, per pole pair and direction, a second-order recurrence at the decimated
    // rate seeded by two table rows.  They decay from the block's ends, so only the groups near a block boundary need them,
    // and only for the slowly decaying pairs: the plan lists the (group, direction, pairs) items of every chunk, costliest
    // first (Lp2Params::items), and the workgroup's threads work the list off -- thread k takes item k, whatever group
    // that is, so that a wavefront's 64 items cost the same.  (Round 2 ran all four pairs in both directions over every
    // lane's own samples: 768 of the kernel's 4076 vector instructions per lane, for responses that are below 1e-30 of the
    // signal in three groups out of four.)
    // Header of a chunk's list: it[0] = items of the first pass, it[1] = items of a second pass behind a barrier (only when
    // some group has a causal AND an anticausal item, i.e. blocks shorter than twice the reach: the two would update the
    // same samples), 0 otherwise.
    template <class Comm>
    TDM_HD void fix_phase(const Lp2Params &P, int row, int chunk, f64x2 *stage, Comm &cm, const Pref &pf) const
    {
        constexpr int ND = PzLayout::kMaxPairs;
        const int32_t *it = P.items + (size_t)chunk * P.items_stride;
        const int cnt1 = it[0], cnt2 = it[1];
        if (pf.mine) run_item(P, stage, pf.w0, pf.sd, pf.cy);   // (operands requested while the samples were staged)
#pragma unroll 1
        for (int k = cm.tid() + kLp2Lanes; k < cnt1; k += kLp2Lanes) {
            f64x2 sdv[2 * ND], cyv[2 * ND];
            const int w0 = it[2 + 2 * k], w1 = it[3 + 2 * k];
            item_operands(P, row, w0, w1, sdv, cyv);
            run_item(P, stage, w0, sdv, cyv);
        }
        if (cnt2 == 0) return;
        cm.sync();
        // (the second pass hands its list out from the last thread downwards: its costly head falls to the wavefronts the
        // first pass left idle)
#pragma unroll 1
        for (int k = kLp2Lanes - 1 - cm.tid(); k < cnt2; k += kLp2Lanes) {
            f64x2 sdv[2 * ND], cyv[2 * ND];
            const int w0 = it[2 + 2 * (cnt1 + k)], w1 = it[3 + 2 * (cnt1 + k)];
            item_operands(P, row, w0, w1, sdv, cyv);
            run_item(P, stage, w0, sdv, cyv);
        }
    }
};

// What a thread has in flight for a chunk before the chunk's arithmetic starts: its La samples (coalesced: sample
// i * 64 + lane of the wavefront's run) and the operands of its first carry-response item.
template <class Src>
struct Lp2Loads {
    f64x2 v[kLp2La];
    typename Src::Pref pref;
};

// the item words of a chunk (two per thread, the oldest loads in flight), then the samples, then the operands the words name
template <class Src, class Comm>
TDM_HD void lp2_issue_words(const Lp2Params &P, const Src &src, Comm &cm, int chunk, Lp2Loads<Src> &L)
{
    if (Src::kFix) src.prefetch_words(P, chunk, cm, L.pref);
}
template <class Src, class Comm>
TDM_HD void lp2_issue_samples(const Lp2Params &P, const Src &src, Comm &cm, int chunk, int row, Lp2Loads<Src> &L)
{
    constexpr int La = kLp2La;
    const int tid = cm.tid(), lane = tid & 63, wave = tid >> 6;
    const int64_t jc = (int64_t)chunk * P.U - P.H - P.off;   // position of lane 0's first sample
    const f64x2 *rowp = src.raw_row(row);
    const int64_t jw = jc + (int64_t)wave * (kWave * La);
    // unconditional loads from a clamped 32-bit index (a load under a branch gets a wait of its own: sixteen serial
    // round trips), all of a lane's loads in flight at once, values outside the row zeroed afterwards
    const int jw32 = (int)jw + lane, n32 = (int)P.n;     // (|positions| < 2^31: tdm_plan_create bounds the chunk length)
#pragma unroll
    for (int i = 0; i < La; ++i) {
        const int j = jw32 + i * kWave;
        const int jj = j < 0 ? 0 : (j >= n32 ? n32 - 1 : j);
        L.v[i] = rowp[jj];
    }
    if (Src::kFix) src.prefetch_operands(P, row, L.pref);
}

// the chunk's arithmetic: the loaded samples to LDS, carry responses, NCO, channel filter; ends with the filter output in
// the staging area (behind a barrier).  nco_w: the NCO's step phasor in LDS
template <class Src, class Comm>
TDM_HD void lp2_compute(const Lp2Params &P, const Src &src, Comm &cm, int chunk, int row, Lp2Loads<Src> &L, const double *nco_w)
{
    constexpr int La = kLp2La, NP = kLp2Pairs;
    const int tid = cm.tid(), lane = tid & 63, wave = tid >> 6;
    const double *cst = P.cst, *lane_m = P.lane_m;
    const int64_t n = P.n;
    const int edge = P.edge;
    const int64_t jc = (int64_t)chunk * P.U - P.H - P.off;   // position of lane 0's first sample
    const int64_t js = jc + (int64_t)tid * La;
    f64x2 *stage = (f64x2 *)cm.stage();
    double *small = cm.small();
    const bool any_sig = (js + La > 0 && js < n);          // the lane holds at least one sample of the row
    // chunks that hold an end of the row: odd extension and start states (workgroup-uniform)
    const bool wg_edge = (jc < 0) || (jc + (int64_t)kLp2Span > n);
    typename Src::Pref &pref = L.pref;
    {
        const int64_t jw = jc + (int64_t)wave * (kWave * La);
        const int jw32 = (int)jw + lane, n32 = (int)n;
        f64x2 (&v)[La] = L.v;
        if (!wg_edge) {   // every position of the span lies inside the row: nothing to mask
#pragma unroll
            for (int i = 0; i < La; ++i) stage[lp2_slot(wave * (kWave * La) + i * kWave + lane)] = v[i];
        } else {
#pragma unroll
            for (int i = 0; i < La; ++i) {
                const int j = jw32 + i * kWave;
                const bool ok = (j >= 0 && j < n32);
                stage[lp2_slot(wave * (kWave * La) + i * kWave + lane)] = f64x2{ok ? v[i].x : 0.0, ok ? v[i].y : 0.0};
            }
        }
    }
    cm.sync();
    if (Src::kFix) {
        src.fix_phase(P, row, chunk, stage, cm, pref);
        cm.sync();
    }
    double yr[La], yi[La];
#pragma unroll
    for (int i = 0; i < La; ++i) {
        const f64x2 v = stage[lp2_slot(tid * La + i)];
        yr[i] = v.x;
        yi[i] = v.y;
    }
    // lanes that hold signal samples: NCO (for a lane that is only partly inside the row the values at positions outside
    // it are meaningless here and are replaced by the odd extension below; lanes outside the row hold zeros)
    const bool inside = (js >= 0 && js + La <= n);
    if (any_sig && Src::kFix) {
        const double f = src.foff(row);
        if (f != 0.0) {
            // the lane's first phasor from the workgroup's tables (lp2_body): anchor x W^(64 La wave) x W^(La lane), then one
            // complex multiplication by W per sample
            const double *t = nco_w;
            const double a0r = t[2], a0i = t[3], b0r = t[4 + 2 * wave], b0i = t[5 + 2 * wave];
            const double m0r = a0r * b0r - a0i * b0i, m0i = a0r * b0i + a0i * b0r;
            const double l0r = t[16 + 2 * lane], l0i = t[17 + 2 * lane];
            double c = m0r * l0r - m0i * l0i, sn = m0r * l0i + m0i * l0r;
            const double wr = t[0], wi = t[1];
#pragma unroll
            for (int i = 0; i < La; ++i) {
                const double a = yr[i], b = yi[i];
                yr[i] = a * c - b * sn;
                yi[i] = a * sn + b * c;
                const double nc = c * wr - sn * wi, ns = c * wi + sn * wr;
                c = nc;
                sn = ns;
            }
        }
    }
    auto Y = [&](int64_t j) { return stage[lp2_slot((int)(j - jc))]; };
    // the empty lanes next to the ends of the extended row carry the start states of the two banks (lp2_tables.hpp)
    const int64_t t_head = ((-(int64_t)edge - jc) >= 0 ? (-(int64_t)edge - jc) / La : -((jc + edge + La - 1) / La)) - 1;   // empty lane before position -edge
    const int64_t t_l1 = (n + edge - 1 - jc) / La;          // lane holding the last extended sample
    const int64_t t_tail = (n + edge - jc) / La + 1;        // empty lane after the lane holding position n + edge
    const bool has_head = wg_edge && (t_head >= 0 && t_head < kLp2Lanes);
    const bool has_tail = wg_edge && (n + edge - 1 - jc >= 0 && t_tail < kLp2Lanes && t_l1 >= 0);
    double e0r = 0, e0i = 0, xlr = 0, xli = 0;
    if (wg_edge) {
    // a chunk that holds an end of the row: publish the finished samples; the odd extension (scipy odd_ext: 2 x[0] - x[-j],
    // 2 x[n-1] - x[2n-2-j]) of the lanes around the end reads them from there.  (Interior chunks -- six of the eight of a
    // 26 215-sample row -- skip the sixteen stores, the barrier and everything below.)
#pragma unroll
    for (int i = 0; i < La; ++i) stage[lp2_slot(tid * La + i)] = f64x2{yr[i], yi[i]};
    cm.sync();
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
    }   // wg_edge

    // C^(La (k+1)), k = lane's distance to the previous / next row of 16 lanes: what the scans' last step multiplies with
    double lmf[NP][4], lmb[NP][4];
    auto load_lm = [&](auto dir_c) __attribute__((always_inline)) {
        constexpr int dir = decltype(dir_c)::value;
        const int r = lane & 15;
        const f64x2 *t = (const f64x2 *)lane_m + (size_t)(dir == 0 ? r : 15 - r) * NP * 2;
        double(*lm)[4] = dir == 0 ? lmf : lmb;
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const f64x2 a0 = t[s * 2], a1 = t[s * 2 + 1];
            lm[s][0] = a0.x; lm[s][1] = a0.y; lm[s][2] = a1.x; lm[s][3] = a1.y;
        }
    };
    // (requested here, used by the scans: the latency overlaps pass 1)
    load_lm(std::integral_constant<int, 0>{});
    load_lm(std::integral_constant<int, 1>{});
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
        const double *hv = cst + Lp2Cst::head_v;
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            zr[s][0] = hv[s * 2] * e0r; zr[s][1] = hv[s * 2 + 1] * e0r;
            zq[s][0] = hv[s * 2] * e0i; zq[s][1] = hv[s * 2 + 1] * e0i;
        }
    }
    // ---------------- scans.  dir 0: causal (inclusive from the left); dir 1: anticausal (from the right) ----------------
    // Per direction: (1) four Kogge-Stone steps inside each row of 16 lanes (DPP row shifts), the row's total to LDS;
    // (2) after ONE barrier every lane forms the state that enters its row from the totals of the P.scan_rows rows before
    // it (Horner in C^(16 La); the filter's memory has decayed below 1e-24 beyond that: two rows at the usual cutoffs),
    // adds its share C^(La (k+1)) of it and takes its start state from its neighbour inside the row.
    // Round 2 passed row totals on with two more cross-lane steps (the anticausal ones through ds_bpermute), then wave
    // totals through LDS and a serial loop over the other wavefronts with a scalar load per turn: six barriers and a long
    // chain of dependent round trips -- a quarter of the kernel's time for a seventh of its instructions.
    // The two directions are independent EXCEPT in the chunk that holds the end of the row, where the anticausal bank
    // starts from the causal bank's state there (scipy: zi * forward[last]): that chunk runs them one after the other.
    constexpr int kRows = kLp2Lanes / 16;
    static_assert(2 * kRows * NP * 4 <= Lp2Lds::oPow, "row totals of both directions");
    double *tail_cst = small + Lp2Lds::oPow;     // (the power partials' area is free until the output stage)
    auto scan_rows = [&](auto dir_c) __attribute__((always_inline)) {
        constexpr int dir = decltype(dir_c)::value;
        double(*vr)[2] = dir == 0 ? zr : ur;
        double(*vq)[2] = dir == 0 ? zq : uq;
        const double *Msc = cst + Lp2Cst::Mscan;
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
        // the row's total: [dir][row of the workgroup][pair][4]
        if ((lane & 15) == (dir == 0 ? 15 : 0)) {
            double *o = small + Lp2Lds::oTot + ((dir * kRows + (tid >> 4)) * NP) * 4;
#pragma unroll
            for (int s = 0; s < NP; ++s) { o[s * 4] = vr[s][0]; o[s * 4 + 1] = vr[s][1]; o[s * 4 + 2] = vq[s][0]; o[s * 4 + 3] = vq[s][1]; }
        }
    };
    auto scan_apply = [&](auto dir_c) __attribute__((always_inline)) {
        constexpr int dir = decltype(dir_c)::value;
        double(*vr)[2] = dir == 0 ? zr : ur;
        double(*vq)[2] = dir == 0 ? zq : uq;
        const double *tot = small + Lp2Lds::oTot + dir * (kRows * NP * 4);
        const int g = tid >> 4;                  // the lane's row in the workgroup
        double pr[NP][2], pq[NP][2];             // state entering the row
#pragma unroll
        for (int s = 0; s < NP; ++s) { pr[s][0] = 0; pr[s][1] = 0; pq[s][0] = 0; pq[s][1] = 0; }
        double mrow[NP][4];
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const auto M = TDM_CPTR(cst + Lp2Cst::Mrow + s * 4);
            mrow[s][0] = M[0]; mrow[s][1] = M[1]; mrow[s][2] = M[2]; mrow[s][3] = M[3];
        }
#pragma unroll 1
        for (int i = P.scan_rows; i >= 1; --i) {          // the farthest row first: P <- C^(16 La) P + T
            const int src = dir == 0 ? g - i : g + i;
            const bool in = src >= 0 && src < kRows;      // (rows beyond the workgroup: zero -- that is what the halo is for)
            const double *t = tot + (in ? src : g) * (NP * 4);
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                const double t0 = in ? t[s * 4] : 0.0, t1 = in ? t[s * 4 + 1] : 0.0, t2 = in ? t[s * 4 + 2] : 0.0, t3 = in ? t[s * 4 + 3] : 0.0;
                const double a0 = fma(mrow[s][0], pr[s][0], fma(mrow[s][1], pr[s][1], t0));
                const double a1 = fma(mrow[s][2], pr[s][0], fma(mrow[s][3], pr[s][1], t1));
                const double b0 = fma(mrow[s][0], pq[s][0], fma(mrow[s][1], pq[s][1], t2));
                const double b1 = fma(mrow[s][2], pq[s][0], fma(mrow[s][3], pq[s][1], t3));
                pr[s][0] = a0; pr[s][1] = a1; pq[s][0] = b0; pq[s][1] = b1;
            }
        }
        // every lane: true inclusive state = in-row value + C^(La (position in the row + 1)) * (state entering the row)
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            const double *M = dir == 0 ? lmf[s] : lmb[s];
            vr[s][0] = fma(M[0], pr[s][0], fma(M[1], pr[s][1], vr[s][0]));
            vr[s][1] = fma(M[2], pr[s][0], fma(M[3], pr[s][1], vr[s][1]));
            vq[s][0] = fma(M[0], pq[s][0], fma(M[1], pq[s][1], vq[s][0]));
            vq[s][1] = fma(M[2], pq[s][0], fma(M[3], pq[s][1], vq[s][1]));
        }
        if (dir == 0 && has_tail && tid == t_l1) {
#pragma unroll
            for (int s = 0; s < NP; ++s) {
                tail_cst[(2 * s) * 2] = vr[s][0]; tail_cst[(2 * s) * 2 + 1] = vq[s][0];
                tail_cst[(2 * s + 1) * 2] = vr[s][1]; tail_cst[(2 * s + 1) * 2 + 1] = vq[s][1];
            }
        }
        // start state of a lane = inclusive state of its neighbour in the row; the row's first lane starts from the state
        // that enters the row
        double sr[NP][2], sq[NP][2];
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            if (dir == 0) cm.template row_shr2<2>(vr[s], vq[s], sr[s], sq[s], 1);
            else cm.template row_shl2<2>(vr[s], vq[s], sr[s], sq[s], 1);
            const bool at_edge = (lane & 15) == (dir == 0 ? 0 : 15);
            vr[s][0] = at_edge ? pr[s][0] : sr[s][0]; vr[s][1] = at_edge ? pr[s][1] : sr[s][1];
            vq[s][0] = at_edge ? pq[s][0] : sq[s][0]; vq[s][1] = at_edge ? pq[s][1] : sq[s][1];
        }
    };
    using D0 = std::integral_constant<int, 0>;
    using D1 = std::integral_constant<int, 1>;
    if (!has_tail) {
        scan_rows(D0{});
        TDM_SCHED_FENCE();   // (one direction after the other: interleaved, their temporaries overflow the register file)
        scan_rows(D1{});
        cm.sync();
        scan_apply(D0{});
        TDM_SCHED_FENCE();
        scan_apply(D1{});
    } else {
        scan_rows(D0{});
        cm.sync();
        scan_apply(D0{});
        cm.sync();   // (the causal state at the end of the row visible)
        if (tid == t_tail) {
            // the anticausal bank starts from the causal bank's state at the end of the row (scipy: zi * forward[last])
            const double *tm = cst + Lp2Cst::tail_m, *tx = cst + Lp2Cst::tail_x;
#pragma unroll
            for (int r = 0; r < kLp2D; ++r) {
                double ar = tx[r] * xlr, ai = tx[r] * xli;
#pragma unroll
                for (int k = 0; k < kLp2D; ++k) { ar = fma(tm[r * kLp2D + k], tail_cst[2 * k], ar); ai = fma(tm[r * kLp2D + k], tail_cst[2 * k + 1], ai); }
                ur[r / 2][r % 2] = ar;
                uq[r / 2][r % 2] = ai;
            }
        }
        scan_rows(D1{});
        cm.sync();
        scan_apply(D1{});
    }
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
    // ---------------- output: through LDS; one thread per (timing phase, group) stores its phase's samples and sums their powers ----------------
    // (No barrier here: a lane writes the slots only it has read since the carry responses were added; the one place
    // where lanes read their neighbours' slots -- the odd extension in a chunk that holds an end of the row -- lies before
    // the scans' barrier.  Round 2 and the first half of round 3 had one, "all lanes have taken their input out".)
#pragma unroll
    for (int i = 0; i < La; ++i) stage[lp2_slot(tid * La + i)] = f64x2{or_[i], oi[i]};
    cm.sync();
}

// the chunk's filter output from the staging area to memory, phase-major, with the chunk's power sums per timing phase;
// ends behind a barrier after the last read of the staging area (the few threads that add up the partial sums after it
// touch only the small area's power slots)
template <class Comm>
TDM_HD void lp2_store(const Lp2Params &P, Comm &cm, int chunk, int row)
{
    const int tid = cm.tid();
    const int64_t n = P.n;
    const int64_t jc = (int64_t)chunk * P.U - P.H - P.off;
    f64x2 *stage = (f64x2 *)cm.stage();
    double *small = cm.small();
    const int64_t j_lo = (int64_t)chunk * P.U - P.off > 0 ? (int64_t)chunk * P.U - P.off : 0;
    int64_t j_hi = (int64_t)(chunk + 1) * P.U - P.off;
    if (j_hi > n) j_hi = n;
    const int sps = P.sps;
    if (sps > 0) {
        f64x2 *zt = (f64x2 *)P.zt + (int64_t)row * sps * P.zt_k;
        constexpr int NG = Lp2Lds::kPowGroups;
        const int ngrp = kLp2Lanes / sps < NG ? kLp2Lanes / sps : NG;
        const int p2 = tid / ngrp, g2 = tid % ngrp;   // 256 threads: phases 0..12 x 19 groups (sps = 13)
        double acc = 0;
        if (p2 < sps) {
            // 32-bit index arithmetic (positions are below 2^31: tdm_plan_create bounds the chunk length), strength-reduced:
            // the staging offset and the position advance by ngrp * sps per turn, the output pointer by ngrp
            const int n32 = (int)n, jlo = (int)j_lo, jhi = (int)j_hi, jc32 = (int)jc;
            const int np_ = (n32 - p2) / sps;                   // samples phase p owns: j = p + k*sps, k < np_
            const int lim = p2 + np_ * sps;                     // first j NOT owned (extract_symbols, processor.py:199-203)
            const int k0 = jlo <= p2 ? 0 : (jlo - p2 + sps - 1) / sps;   // first k with j = p2 + k*sps >= j_lo
            const int step = ngrp * sps;
            int j = p2 + (k0 + g2) * sps;
            f64x2 *po = zt + (int64_t)p2 * P.zt_k + (k0 + g2);
            for (; j < jhi; j += step, po += ngrp) {
                const f64x2 v = stage[lp2_slot(j - jc32)];
                *po = v;
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
        cm.sync();
    }
}

// one chunk of one row, start to end (the kernel with one workgroup per chunk)
template <class Src, class Comm>
TDM_HD void lp2_body(const Lp2Params &P, const Src &src, Comm &cm, int chunk, int row)
{
    Lp2Loads<Src> L;
    lp2_issue_words(P, src, cm, chunk, L);
    lp2_issue_samples(P, src, cm, chunk, row, L);
    // ---- while the samples are on their way: the NCO's step phasor exp(i Dd), which is the same for the whole row: the
    // last wavefront (the one with the fewest items) forms it and leaves it in LDS for all lanes (round 2: every lane's
    // own sincos, 120 instructions in each of the four wavefronts)
    double *nco_w = cm.small() + Lp2Lds::oPow + 16;     // (the power partials' area is free until the output stage)
    if (Src::kFix) {
        // frequency_shift(samples, freq_offset) at the low rate (processor.py:260-261) as the ideal phase ramp from ONE exactly
        // anchored sample per workgroup: exp(i theta_j) = A0 W^(j - jc), A0 = the reference's own exp(i theta_jc)
        // (nco_phasor: theta = fl(ci fl(j / fs))), W = exp(i ci / fs).  The reference's theta_j differs from the ramp by its
        // own rounding -- at most 1.5e-13 rad at 1.2 kHz x 0.11 s, the level of the filters' arithmetic noise (reproducing it
        // sample by sample, NcoRunT, cost 25 instructions per sample and a sincos per lane: 600 of this kernel's 2947
        // instructions per lane; rounds 1-4 did).  Two wavefronts form the tables while the samples are on
        // their way: [0,1] W, [2,3] A0, [4 + 2 w] W^(64 La w) per wavefront, [16 + 2 l] W^(La l) per lane.
        const int wave = cm.tid() >> 6, lane = cm.tid() & 63;
        const double f = src.foff(row);
        if (f != 0.0 && wave >= kLp2Waves - 2) {
            const double dth = -(2.0 * M_PI) * f / src.fs_out;
            double sn, c;
            if (wave == kLp2Waves - 1) {
                sincos((double)(kLp2La * lane) * dth, &sn, &c);
                nco_w[16 + 2 * lane] = c;
                nco_w[17 + 2 * lane] = sn;
                sincos(dth, &sn, &c);
                if (lane == 0) { nco_w[0] = c; nco_w[1] = sn; }
            } else {
                sincos((double)(kWave * kLp2La * (lane < kLp2Waves ? lane : 0)) * dth, &sn, &c);
                if (lane < kLp2Waves) { nco_w[4 + 2 * lane] = c; nco_w[5 + 2 * lane] = sn; }
                const int64_t jc = (int64_t)chunk * P.U - P.H - P.off;
                const phasor a = nco_phasor(jc, f, src.fs_out);
                if (lane == 0) { nco_w[2] = a.c; nco_w[3] = a.s; }
            }
        }
    }
    lp2_compute(P, src, cm, chunk, row, L, nco_w);
    lp2_store(P, cm, chunk, row);
}

}  // namespace tdm
