// Host-side plan of the low-rate stage in parallel form (lp2_kernels.hpp): the channel filter
// filter_signal(samples, 25000, current_rate) (processor.py:51-83, scipy butter(4) + filtfilt, pad 15) as the
// partial-fraction expansion of H(z)H(1/z) (see pz_tables.hpp), evaluated by workgroups that own a chunk of a
// carrier's low-rate samples plus a halo on either side.  The halo is long enough for the filter's memory to decay
// below 1e-21 (five decades under the rounding of the samples themselves), so chunks need no carries from their neighbours and the filter output is final when it is written.
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>

#include "pz_tables.hpp"

namespace tdm {

// A lane owns 16 consecutive low-rate samples, a workgroup of 4 wavefronts a span of 4096 positions (64 KB of staging).
// (Measured alternatives, all slower or level and no longer in the source -- docs/HISTORY.md, profiles/r05_ab/: eight-sample
// lanes in eight wavefronts, 0.341 ms against 0.330-0.334 once their spills were trimmed; eight wavefronts on a span of 8192
// positions, 0.336 against 0.305; the carry-response items forming the decimator's block carries themselves instead of a
// carry launch, 0.481 against 0.361 + 0.015; the first dispatch round's workgroups started staggered: no change.)
constexpr int kLp2La = 16;                    // samples per lane
constexpr int kLp2Waves = 4;                  // wavefronts per workgroup
constexpr int kLp2GBits = 9;                  // item word 0: group in the span (< 512) | direction << 9 | pair mask << 10
constexpr int kLp2Lanes = kLp2Waves * kWave;
constexpr int kLp2Span = kLp2Lanes * kLp2La;   // positions a workgroup covers (chunk + both halos)
static_assert(kLp2Waves >= 2 && 4 + 2 * kLp2Waves <= 16, "two wavefronts form the NCO's tables; the per-wavefront entries end where the per-lane ones start (lp2_body)");
constexpr int kLp2Pairs = 2;
constexpr int kLp2D = 2 * kLp2Pairs;
// what a chunk may ignore of its neighbours: the states are O(10) per unit input, so the neglected part is below 1e-20 of
// the signal -- four decades under one rounding of a sample (1.1e-16); 1e-30 cost a ninth chunk per 26 215-sample row
constexpr long double kLp2HaloTol = 1e-21L;

struct Lp2Params {
    // ---- geometry: lane t of chunk c covers positions j = c*U - H - off + La*t ... + La - 1 of the low-rate row
    int64_t n;          // samples per row
    int32_t H, U, off;  // halo and usable outputs per chunk (multiples of La), offset of the lane grid (< La)
    int32_t n_chunks;
    int32_t sps;        // timing phases for the power partials (extract_symbols), 0 = none
    int32_t edge;       // odd-extension length (15)
    int32_t scan_rows;  // rows of 16 lanes whose totals still reach a lane's start state (|C^(16 La k)| < 1e-24 beyond)
    // ---- channel filter in parallel form
    double na1[kLp2Pairs], na2[kLp2Pairs], b0[kLp2Pairs], b1[kLp2Pairs], dx;
    // constant block in device memory (read with scalar loads where it is used; as kernel arguments the matrices would
    // all be fetched at the top of the kernel and spilled), layout Lp2Cst
    const double *cst;
    const double *lane_m;            // [64][pairs][4]  C^(La (k+1))
    // ---- decimator fix-up by recurrence (when the input is a parallel-form decimator's block-local output)
    double dec_p1[PzLayout::kMaxPairs], dec_p2[PzLayout::kMaxPairs];   // x[k+1] = p1 x[k] - p2 x[k-1] at the decimated rate
    // seeds of those recurrences: for every group of La outputs of a decimator block (group t = outputs 16t .. 16t+15)
    // the table rows of outputs 16t, 16t+1 (causal) and 16t+14, 16t+15 (anticausal): [2][seed_groups][kLp2SeedDoubles],
    // first the regular blocks' tables, then the last block's
    const double *seeds;
    int32_t seed_groups;   // groups per decimator block + 1
    // The carry responses decay from the block's ends (pole pair s by |lambda_s|^q per output): a group of La outputs
    // needs pair s's causal response only within reach_s outputs of its block's start and the anticausal one within
    // reach_s of its end (beyond, the response is below kLp2FixTol of the signal).  The (group, direction, pairs) items a
    // chunk needs are listed on the host, costliest first, so that the wavefronts of the workgroup that work them off
    // hold items of equal cost:  two words per item after a two-word header:
    //   w0 = group in the workgroup's span | direction << kLp2GBits | pair mask << (kLp2GBits + 1),   w1 = decimator block << 8 | group in block
    // with header items[chunk * items_stride + 0 / 1] = items of the first pass / of a second pass behind a barrier (used
    // only when some group has an item in both directions: then causal items first, anticausal second)
    const int32_t *items;
    int32_t items_stride;
    // ---- outputs
    double *zt;          // [rows][sps][zt_k] c128: filter output, phase-major (sample p + sps*k at [p][k]); plain [n] when sps == 0
    int64_t zt_k;
    double *partials;    // [rows][n_chunks][kMaxSps]
};

constexpr int kLp2SeedDoubles = 32;   // four rows of D = 8 doubles

struct Lp2Cst {   // offsets (doubles) into Lp2Params::cst
    static constexpr int Mscan = 0;     // [pairs][4][4]  C^(La 2^j), j = 0..3 (steps inside a row of 16 lanes)
    static constexpr int Mrow = 32;     // [pairs][4]     C^(16 La)
    static constexpr int Mwave = 40;    // [pairs][4]     C^(64 La)
    // edges (scipy filtfilt: odd extension, start states zi*ext[0] and zi*forward[last])
    static constexpr int head_v = 48;   // [pairs][2]  causal end state of the empty lane before the lane holding position -edge, per unit ext[0]
    static constexpr int tail_m = 52;   // [D][D]      anticausal end state of the empty lane after the lane holding position n+edge:
    static constexpr int tail_x = 68;   // [D]           tail_m * (causal state at the end of the lane holding n+edge-1) + tail_x * ext[last]
    static constexpr int size = 72;
};

struct Lp2Host {
    Lp2Params p;
    std::vector<double> lane_m, seeds, cst;
    std::vector<double> items;   // int32 words, two per double slot (kept in the plan's table blob of doubles)
    bool ok = false;
};
// what a group may ignore of a carry response (relative to a unit carry; the carries are O(100) x the signal, the
// responses start at O(10)): twelve decades under one rounding of a sample
constexpr long double kLp2FixTol = 1e-30L;

// sos: the channel filter as biquads g*[1,2,1]/a (Tf4::sos); dec: the decimator's tables when its output feeds this
// stage (parallel form only), else null
inline Lp2Host build_lp2(const double (*sos)[6], int64_t n, int edge, int sps, const ZpHostTables *dec_t, const double (*dec_sos)[6])
{
    using namespace detail;
    Lp2Host h;
    std::memset(&h.p, 0, sizeof(h.p));
    Lp2Params &p = h.p;
    constexpr int La = kLp2La, NP = kLp2Pairs, D = kLp2D;
    const PzDesign dz = design_pz(sos, NP);
    // halo: |largest pole|^H < kLp2HaloTol
    ldbl rmax = 0;
    for (int s = 0; s < NP; ++s) rmax = std::fmax(rmax, std::sqrt(dz.a2[s]));
    int H = (int)std::ceil(std::log(kLp2HaloTol) / std::log(rmax));
    H = ((H + La - 1) / La) * La;
    if (H < 3 * La) H = 3 * La;
    if (2 * H > kLp2Span / 2) return h;   // (a filter this narrow runs on the cascade engine with its block carries)
    p.n = n;
    p.H = H;
    p.U = kLp2Span - 2 * H;
    p.edge = edge;
    p.sps = sps;
    p.off = 0;
    const ZpParams *dec = dec_t ? &dec_t->p : nullptr;
    if (dec) {
        // lanes must not straddle decimator blocks: block b starts at output b*(Bn/q) - k0L/q
        const int q = dec->out_stride;
        if (!dec->pform || (kWave * dec->L) % (q * La) != 0 || dec->k0L % q != 0) return h;
        p.off = (dec->k0L / q) % La;
        const PzDesign dd = design_pz(dec_sos, dec->nsec);
        for (int s = 0; s < dec->nsec; ++s) {
            const M2 cq = m2pow(dd.C[s], q);
            p.dec_p1[s] = (double)(cq.a + cq.d);
            p.dec_p2[s] = (double)(cq.a * cq.d - cq.b * cq.c);
        }
        // seed rows, gathered from the decimator's own (phase-major) tables
        const int Dd = 2 * dec->nsec;
        const int groups = (kWave * dec->L) / (q * La);
        p.seed_groups = groups + 1;
        h.seeds.assign((size_t)2 * p.seed_groups * kLp2SeedDoubles, 0.0);
        for (int v = 0; v < 2; ++v) {
            const double *T1 = v ? dec_t->blob.data() + dec_t->off_T1last : dec_t->reg_tables() + dec_t->off_T1reg;
            const double *T2 = v ? dec_t->blob.data() + dec_t->off_T2last : dec_t->reg_tables() + dec_t->off_T2reg;
            const int R = v ? dec->R_last : dec->R_reg;
            for (int t = 0; t < p.seed_groups; ++t) {
                double *o = &h.seeds[((size_t)v * p.seed_groups + t) * kLp2SeedDoubles];
                const int rows[4] = {La * t, La * t + 1, La * t + La - 2, La * t + La - 1};   // (q | offsets: phase 0, row = output index)
                for (int k = 0; k < 4; ++k) {
                    if (rows[k] >= R) continue;   // (past the table: outputs the row does not have)
                    const double *src = (k < 2 ? T1 : T2) + (size_t)rows[k] * Dd;
                    for (int d = 0; d < Dd; ++d) o[k * 8 + d] = src[d];
                }
            }
        }
    }
    p.n_chunks = (int32_t)((n + p.off + p.U - 1) / p.U);
    p.dx = (double)dz.dx;
    if (dec) {
        // ---- fix-up items of every chunk (see Lp2Params::items)
        const int q = dec->out_stride, nsec = dec->nsec;
        const int64_t Bn = (int64_t)kWave * dec->L;
        int reach[PzLayout::kMaxPairs] = {0, 0, 0, 0};   // outputs
        for (int s = 0; s < nsec; ++s) {
            const ldbl mu = std::sqrt(std::fabs((ldbl)p.dec_p2[s]));   // |lambda_s|^q
            reach[s] = mu < 1 ? (int)std::ceil(std::log(kLp2FixTol) / std::log(mu)) : (1 << 30);
        }
        std::vector<std::vector<int32_t>> per_chunk(p.n_chunks);
        size_t max_words = 0;
        for (int c = 0; c < p.n_chunks; ++c) {
            const int64_t jc = (int64_t)c * p.U - p.H - p.off;
            std::vector<std::pair<int, std::pair<int32_t, int32_t>>> it;   // (cost key, words)
            bool both = false;
            for (int g = 0; g < kLp2Lanes; ++g) {
                const int64_t js = jc + (int64_t)g * La;
                if (!(js + La > 0 && js < n)) continue;                 // no sample of the row in this group
                const int64_t pos = dec->k0L + js * q;                  // extended-domain index of the group's first output
                if (pos < 0) continue;                                  // (a group wholly before the row is excluded above; partly: its row samples start at 0)
                const int b = (int)(pos / Bn);
                const int64_t m = pos - (int64_t)b * Bn;                // in-block offset (input samples), a multiple of q
                const int t = (int)(m / ((int64_t)q * La));
                const int64_t len_b = b == dec->nb - 1 ? dec->len_last : Bn;
                const int64_t d_c = m / q;                               // outputs between the block's start and the group
                int64_t d_a = (len_b - 1 - (m + (int64_t)(La - 1) * q)) / q;   // ... between the group's last output and the block's end
                if (d_a < 0) d_a = 0;
                int mc = 0, ma = 0;
                for (int s = 0; s < nsec; ++s) {
                    if (d_c < reach[s]) mc |= 1 << s;
                    if (d_a < reach[s]) ma |= 1 << s;
                }
                auto bits = [](int v) { int k = 0; for (; v; v &= v - 1) ++k; return k; };
                if (mc && ma) both = true;
                // sort key (filled in below): pass, then costliest first
                if (mc) it.push_back({16 - bits(mc), {g | (0 << kLp2GBits) | (mc << (kLp2GBits + 1)), (b << 8) | t}});
                if (ma) it.push_back({16 - bits(ma) + 1000, {g | (1 << kLp2GBits) | (ma << (kLp2GBits + 1)), (b << 8) | t}});
            }
            // one pass when no group has an item in both directions (long decimator blocks: the usual case); otherwise the
            // causal items, a barrier, the anticausal items
            if (!both)
                for (auto &e : it) e.first %= 1000;
            std::stable_sort(it.begin(), it.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
            auto &w = per_chunk[c];
            int n_1 = 0;
            for (auto &e : it) n_1 += e.first < 1000;
            w.push_back(n_1);
            w.push_back((int32_t)it.size() - n_1);
            for (auto &e : it) { w.push_back(e.second.first); w.push_back(e.second.second); }
            if (w.size() > max_words) max_words = w.size();
        }
        // (at least one item slot per thread: the kernel loads its first item's words before it knows the count)
        if (max_words < 2 + 2 * (size_t)kLp2Lanes) max_words = 2 + 2 * (size_t)kLp2Lanes;
        p.items_stride = (int32_t)((max_words + 1) & ~(size_t)1);
        std::vector<int32_t> flat((size_t)p.items_stride * p.n_chunks, 0);
        for (int c = 0; c < p.n_chunks; ++c) std::copy(per_chunk[c].begin(), per_chunk[c].end(), flat.begin() + (size_t)c * p.items_stride);
        h.items.assign(flat.size() / 2, 0.0);
        std::memcpy(h.items.data(), flat.data(), flat.size() * sizeof(int32_t));
    }
    h.lane_m.assign((size_t)kWave * NP * 4, 0.0);
    h.cst.assign(Lp2Cst::size, 0.0);
    double *cst = h.cst.data();
    for (int s = 0; s < NP; ++s) {
        p.na1[s] = (double)-dz.a1[s];
        p.na2[s] = (double)-dz.a2[s];
        p.b0[s] = (double)dz.b0[s];
        p.b1[s] = (double)dz.b1[s];
        auto put = [](double *o, const M2 &m) { o[0] = (double)m.a; o[1] = (double)m.b; o[2] = (double)m.c; o[3] = (double)m.d; };
        for (int j = 0; j < 4; ++j) put(cst + Lp2Cst::Mscan + (s * 4 + j) * 4, m2pow(dz.C[s], (long)La << j));
        put(cst + Lp2Cst::Mrow + s * 4, m2pow(dz.C[s], 16L * La));
        put(cst + Lp2Cst::Mwave + s * 4, m2pow(dz.C[s], 64L * La));
        for (int k = 0; k < kWave; ++k) put(&h.lane_m[((size_t)k * NP + s) * 4], m2pow(dz.C[s], (long)La * (k + 1)));
    }
    // ---- edges.  Positions are congruent to -off modulo La at every lane start, in every chunk.
    auto pmod = [&](int64_t j) { return (int)(((j + p.off) % La + La) % La); };   // index of position j inside its lane
    {
        // causal start: the state before position -edge is g*ext[0] in both components (constant history).  The lane
        // holding that position has zeros before it, so the same run results from starting the LANE in C^(-ki) (g,g)
        // -- which is what the scan delivers if the empty lane before it ends in that state.
        const int ki = pmod(-(int64_t)edge);
        for (int s = 0; s < NP; ++s) {
            const ldbl g = 1 / (1 + dz.a1[s] + dz.a2[s]);
            const M2 ci = m2pow(m2inv(dz.C[s]), ki);
            cst[Lp2Cst::head_v + s * 2] = (double)(ci.a * g + ci.b * g);
            cst[Lp2Cst::head_v + s * 2 + 1] = (double)(ci.c * g + ci.d * g);
        }
        // anticausal start at position n+edge (first one past the extension): V = AE * (causal state at n+edge-1) +
        // wx * ext[last].  The kernel has the causal state at the END of the lane holding n+edge-1 (kinv zero steps
        // later) and delivers V by letting the empty lane after the lane holding n+edge end in C^-(La - kv) V.
        const int kinv = La - 1 - pmod(n + edge - 1);
        const int kv = pmod(n + edge);
        ldbl pre[D][D + 1];
        for (int r = 0; r < D; ++r) {
            for (int s = 0; s < NP; ++s) {
                const M2 ci = m2pow(m2inv(dz.C[s]), kinv);
                const ldbl e0 = dz.AE[(size_t)r * (D + 1) + 2 * s], e1 = dz.AE[(size_t)r * (D + 1) + 2 * s + 1];
                pre[r][2 * s] = e0 * ci.a + e1 * ci.c;
                pre[r][2 * s + 1] = e0 * ci.b + e1 * ci.d;
            }
            pre[r][D] = dz.AE[(size_t)r * (D + 1) + D];
        }
        for (int s = 0; s < NP; ++s) {
            const M2 cb = m2pow(m2inv(dz.C[s]), La - kv);
            for (int c = 0; c <= D; ++c) {
                const ldbl v0 = cb.a * pre[2 * s][c] + cb.b * pre[2 * s + 1][c];
                const ldbl v1 = cb.c * pre[2 * s][c] + cb.d * pre[2 * s + 1][c];
                if (c < D) {
                    cst[Lp2Cst::tail_m + (2 * s) * D + c] = (double)v0;
                    cst[Lp2Cst::tail_m + (2 * s + 1) * D + c] = (double)v1;
                } else {
                    cst[Lp2Cst::tail_x + 2 * s] = (double)v0;
                    cst[Lp2Cst::tail_x + 2 * s + 1] = (double)v1;
                }
            }
        }
    }
    {
        // how many rows of 16 lanes back the scan has to look: the state entering a row from k rows away has passed
        // through C^(16 La (k-1)) at least
        int k = 1;
        for (; k < kLp2Lanes / 16; ++k) {
            ldbl worst = 0;
            for (int s = 0; s < NP; ++s) {
                const M2 m = m2pow(dz.C[s], 16L * La * k);
                worst = std::fmax(worst, std::fmax(std::fmax(std::fabs(m.a), std::fabs(m.b)), std::fmax(std::fabs(m.c), std::fabs(m.d))));
            }
            if (worst < 1e-24L) break;
        }
        p.scan_rows = k;
    }
    p.zt_k = sps > 0 ? (n + sps - 1) / sps : n;
    h.ok = true;
    return h;
}

}  // namespace tdm
