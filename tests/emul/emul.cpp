// TEST INFRASTRUCTURE ONLY -- CPU lock-step emulation of the HIP kernels.
//
// Compiles the kernel bodies of tetraear_amd/csrc/zp_kernels.hpp with g++ and runs each
// "workgroup" as a set of std::threads that exchange data through barriers where the GPU uses
// wavefront shuffles / LDS reductions.  It exists so that the table/index logic of the device
// code can be validated against the oracle in the CPU test tier (no GPU in the build container).
// It is NOT part of the product: tetraear_amd never loads it and libtetrahip.so has no CPU path.
#include <barrier>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <thread>
#include <vector>

#include "../../tetraear_amd/csrc/ref_pipeline.hpp"
#include "../../tetraear_amd/csrc/resample_plan.hpp"
#include "../../tetraear_amd/csrc/sync_kernels.hpp"
#include "../../tetraear_amd/csrc/gate_kernels.hpp"
#include "../../tetraear_amd/csrc/detect_kernels.hpp"
#include "../../tetraear_amd/csrc/small_dft.hpp"
#include "../../tetraear_amd/csrc/tetra_taps.hpp"

using namespace tdm;

namespace {

struct Group {
    int nt;
    std::barrier<> bar;
    std::vector<double> slots;
    explicit Group(int n) : nt(n), bar(n), slots((size_t)n * 16 + 1024 + 2 * 2200 + 4 * 2048 + (n > 64 ? Lp2Lds::kStage + Lp2Lds::kSmall + 64 : 0)) {}
};

struct EmuWaveComm {
    Group *g;
    int lane;
    double *stage() { return &g->slots[(size_t)g->nt * 16 + 1024]; }
    double *edge_slots() { return stage(); }   // (the parallel-form kernel has no staged loader)
    void wave_sync() { g->bar.arrive_and_wait(); }
    template <int K>
    void xchg(const double *a, const double *b, double *oa, double *ob, int src)
    {
        double *mine = &g->slots[(size_t)lane * 16];
        for (int k = 0; k < K; ++k) { mine[k] = a[k]; mine[8 + k] = b[k]; }
        g->bar.arrive_and_wait();
        const int s = (src >= 0 && src < g->nt) ? src : lane;
        const double *from = &g->slots[(size_t)s * 16];
        for (int k = 0; k < K; ++k) { oa[k] = from[k]; ob[k] = from[8 + k]; }
        g->bar.arrive_and_wait();
    }
    // source lane or -1 -> zeros (the DPP / bpermute shuffles of the device build)
    template <int K>
    void gather0(const double *a, const double *b, double *oa, double *ob, int src)
    {
        xchg<K>(a, b, oa, ob, src < 0 ? lane : src);
        if (src < 0)
            for (int k = 0; k < K; ++k) { oa[k] = 0; ob[k] = 0; }
    }
    template <int K>
    void row_shr2(const double *a, const double *b, double *oa, double *ob, int d) { gather0<K>(a, b, oa, ob, (lane & 15) >= d ? lane - d : -1); }
    template <int K>
    void row_shl2(const double *a, const double *b, double *oa, double *ob, int d) { gather0<K>(a, b, oa, ob, (lane & 15) + d <= 15 ? lane + d : -1); }
    template <int K>
    void row_total_prev2(const double *a, const double *b, double *oa, double *ob, int step)
    {
        const int row = lane >> 4;
        gather0<K>(a, b, oa, ob, step == 0 ? ((row & 1) ? 16 * row - 1 : -1) : (row >= 2 ? 31 : -1));
    }
    template <int K>
    void row_total_next2(const double *a, const double *b, double *oa, double *ob, int step)
    {
        const int row = lane >> 4;
        gather0<K>(a, b, oa, ob, step == 0 ? ((row & 1) ? -1 : 16 * (row + 1)) : (row < 2 ? 32 : -1));
    }
    template <int K>
    void wave_shr1(const double *a, const double *b, double *oa, double *ob) { gather0<K>(a, b, oa, ob, lane > 0 ? lane - 1 : -1); }
    template <int K>
    void wave_shl1(const double *a, const double *b, double *oa, double *ob) { gather0<K>(a, b, oa, ob, lane < 63 ? lane + 1 : -1); }
    template <int K>
    void shfl_up2(const double *a, const double *b, double *oa, double *ob, int d) { xchg<K>(a, b, oa, ob, lane - d); }
    template <int K>
    void shfl_down2(const double *a, const double *b, double *oa, double *ob, int d) { xchg<K>(a, b, oa, ob, lane + d); }
};

// workgroup of several wavefronts (lp2_kernels.hpp): shuffles stay inside a thread's own 64-lane wavefront, the barrier
// is the whole group's (every thread runs the same sequence of exchanges)
struct EmuWgComm {
    Group *g;
    int t;
    int tid() const { return t; }
    void sync() { g->bar.arrive_and_wait(); }
    double *stage() { return &g->slots[(size_t)g->nt * 16 + 1024 + 2 * 2200 + 4 * 2048]; }
    double *small() { return stage() + Lp2Lds::kStage; }
    template <int K>
    void gather0(const double *a, const double *b, double *oa, double *ob, int src_lane)
    {
        double *mine = &g->slots[(size_t)t * 16];
        for (int k = 0; k < K; ++k) { mine[k] = a[k]; mine[8 + k] = b[k]; }
        g->bar.arrive_and_wait();
        if (src_lane >= 0) {
            const double *from = &g->slots[(size_t)((t & ~63) + src_lane) * 16];
            for (int k = 0; k < K; ++k) { oa[k] = from[k]; ob[k] = from[8 + k]; }
        } else {
            for (int k = 0; k < K; ++k) { oa[k] = 0; ob[k] = 0; }
        }
        g->bar.arrive_and_wait();
    }
    template <int K>
    void row_shr2(const double *a, const double *b, double *oa, double *ob, int d) { const int l = t & 63; gather0<K>(a, b, oa, ob, (l & 15) >= d ? l - d : -1); }
    template <int K>
    void row_shl2(const double *a, const double *b, double *oa, double *ob, int d) { const int l = t & 63; gather0<K>(a, b, oa, ob, (l & 15) + d <= 15 ? l + d : -1); }
    template <int K>
    void row_total_prev2(const double *a, const double *b, double *oa, double *ob, int step)
    {
        const int row = (t & 63) >> 4;
        gather0<K>(a, b, oa, ob, step == 0 ? ((row & 1) ? 16 * row - 1 : -1) : (row >= 2 ? 31 : -1));
    }
    template <int K>
    void row_total_next2(const double *a, const double *b, double *oa, double *ob, int step)
    {
        const int row = (t & 63) >> 4;
        gather0<K>(a, b, oa, ob, step == 0 ? ((row & 1) ? -1 : 16 * (row + 1)) : (row < 2 ? 32 : -1));
    }
    template <int K>
    void wave_shr1(const double *a, const double *b, double *oa, double *ob) { const int l = t & 63; gather0<K>(a, b, oa, ob, l > 0 ? l - 1 : -1); }
    template <int K>
    void wave_shl1(const double *a, const double *b, double *oa, double *ob) { const int l = t & 63; gather0<K>(a, b, oa, ob, l < 63 ? l + 1 : -1); }
};

struct EmuBlockComm {
    Group *g;
    int t;
    int tid() const { return t; }
    int nthreads() const { return g->nt; }
    void sync() { g->bar.arrive_and_wait(); }
    double &lds(int i) { return g->slots[(size_t)g->nt * 2 + i]; }
    double *smem() { return &g->slots[(size_t)g->nt * 16 + 1024 + 2 * 2200]; }
    template <class F>
    double reduce(double v, F f)
    {
        g->slots[t] = v;
        g->bar.arrive_and_wait();
        double r = g->slots[0];
        for (int i = 1; i < g->nt; ++i) r = f(r, g->slots[i]);
        g->bar.arrive_and_wait();
        return r;
    }
    double reduce_sum(double v) { return reduce(v, [](double a, double b) { return a + b; }); }
    double reduce_max(double v) { return reduce(v, [](double a, double b) { return std::fmax(a, b); }); }
    double reduce_min(double v) { return reduce(v, [](double a, double b) { return std::fmin(a, b); }); }
    double reduce_max_nan(double v) { return reduce(v, [](double a, double b) { return (a != a) ? a : ((b != b) ? b : std::fmax(a, b)); }); }
};

template <class F>
void run_group(int nt, F fn)
{
    Group g(nt);
    std::vector<std::thread> th;
    th.reserve(nt);
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { fn(t, &g); });
    for (auto &x : th) x.join();
}

struct EmuBackend {
    template <int K, int NSEC, int L, int EDGE, class Loader>
    void zp_block(const ZpParams &P, Loader ld, int nb, int rows)
    {
        for (int row = 0; row < rows; ++row)
            for (int b = 0; b < nb; ++b)
                run_group(kWave, [&](int lane, Group *g) {
                    EmuWaveComm cm{g, lane};
                    zp_block_body<K, NSEC, L, EDGE>(P, ld, cm, lane, b, row);
                });
    }
    template <int Q, int S, int EDGE, class Loader>
    void pz_block(const ZpParams &P, Loader ld, int nb, int rows)
    {
        for (int row = 0; row < rows; ++row)
            for (int b = 0; b < nb; ++b)
                run_group(kWave, [&](int lane, Group *g) {
                    EmuWaveComm cm{g, lane};
                    pz_block_body<Q, S, EDGE>(P, ld, cm, lane, b, row);
                });
    }
    template <int Q, int S, int EDGE, int FMT8>
    void pz_raw(const ZpParams &P, const void *iq, int64_t stride, int b_tail, int rows)
    {
        for (int row = 0; row < rows; ++row)
            for (int blk = 0; blk < P.nb; ++blk)
                run_group(kWave, [&](int lane, Group *g) {
                    EmuWaveComm cm{g, lane};
                    if (blk == 0 || blk >= b_tail)
                        pz_raw_body<Q, S, EDGE, FMT8, true>(P, iq, stride, cm, lane, blk, row);
                    else
                        pz_raw_body<Q, S, EDGE, FMT8, false>(P, iq, stride, cm, lane, blk, row);
                });
    }
    template <class Src>
    void lp2(const Lp2Params &P, const Src &src, int rows)
    {
        for (int row = 0; row < rows; ++row)
            for (int c = 0; c < P.n_chunks; ++c)
                run_group(kLp2Lanes, [&](int t, Group *g) {
                    EmuWgComm cm{g, t};
                    lp2_body(P, src, cm, c, row);
                });
    }
    template <int K, int NSEC>
    void zp_carry(const ZpParams &P, int nb, int rows)
    {
        if (P.pform) {
            for (int row = 0; row < rows; ++row)
                for (int b = nb - 1; b >= 0; --b)   // (any order: every thread is self-contained)
                    for (int ch = 0; ch < 2; ++ch) pz_carry_body<NSEC>(P, row, b, ch);
            return;
        }
        for (int row = 0; row < rows; ++row)
            for (int b = 0; b < nb; ++b)
                for (int ch = 0; ch < 2; ++ch) zp_carry_fwd_body<K, NSEC>(P, row, b, ch);
        for (int row = 0; row < rows; ++row)
            for (int b = 0; b < nb; ++b)
                for (int ch = 0; ch < 2; ++ch) zp_carry_bwd_body<K, NSEC>(P, row, b, ch);
    }
    template <int D, int L>
    void zp_fixup(const ZpParams &P, int nb, int rows, double *out, int64_t out_row_stride,
                  const double *freq_offset, double fs_out)
    {
        const int nt = 7;  // any thread count must give the same result
        for (int row = 0; row < rows; ++row)
            for (int b = 0; b < nb; ++b)
                for (int t = 0; t < nt; ++t)
                    zp_fixup_body<D, L>(P, row, b, t, nt, out + (int64_t)row * out_row_stride * 2, freq_offset, fs_out);
    }
    template <class Loader>
    void convert(Loader ld, int rows, int64_t n, double *out, const double *freq_offset, double fs)
    {
        for (int row = 0; row < rows; ++row)
            for (int64_t j = 0; j < n; ++j) convert_body(ld, row, j, out + (int64_t)row * n * 2, freq_offset, fs);
    }
    template <int D, int L>
    void power_fixup(const ZpParams &P, int rows, int64_t n, double *z, int sps, double *partials, int n_pblk)
    {
        for (int row = 0; row < rows; ++row)
            for (int b = 0; b < (n_pblk + kPowSub - 1) / kPowSub; ++b)
                run_group(kPowThreads, [&](int t, Group *g) {
                    EmuBlockComm cm{g, t};
                    power_fixup_body<D, L>(P, cm, row, b, z ? z + (int64_t)row * n * 2 : nullptr, n, sps,
                                           partials + (int64_t)row * n_pblk * kMaxSps, n_pblk);
                });
    }
    template <class Src>
    void lp2_finish(const Lp2Params &P, const Src &src, const FinishArgs &fa, int rows)
    {
        lp2(P, src, rows);
        finish(fa, rows);
    }
    void finish(const FinishArgs &fa, int rows)
    {
        for (int row = 0; row < rows; ++row)
            run_group(64, [&](int t, Group *g) {
                EmuBlockComm cm{g, t};
                finish_body(fa, cm, row);
            });
    }
};

struct HostZp {
    ZpHostTables t;
    std::vector<double> y0, Ef, Eb, flast, Gf, Hb, Elast;
    void bind(int rows)
    {
        ZpParams &p = t.p;
        const int D = p.nsec * p.K;
        t.bind(p, t.blob.data(), t.shared ? t.shared->blob.data() : nullptr);
        const double nan = std::numeric_limits<double>::quiet_NaN();
        y0.assign((size_t)rows * p.n_out * 2 + 2, nan);
        Ef.assign((size_t)rows * p.nb * D * 2, nan);
        Eb.assign((size_t)rows * p.nb * D * 2, nan);
        Gf.assign((size_t)rows * p.nb * D * 2, nan);
        Hb.assign((size_t)rows * p.nb * D * 2, nan);
        flast.assign((size_t)rows * 2, nan);
        Elast.assign((size_t)rows * D * 2, nan);
        p.y0 = y0.data(); p.Ef = Ef.data(); p.Eb = Eb.data();
        p.Gf = Gf.data(); p.Hb = Hb.data(); p.flast = flast.data(); p.Elast = Elast.data();
    }
};

}  // namespace


// register-resident small DFTs of the channeliser (small_dft.hpp); in/out interleaved float re,im
template <int N>
static void small_dft_host(const float *in, float *out)
{
    cf32v x[N];
    for (int i = 0; i < N; ++i) x[i] = cv(in[2 * i], in[2 * i + 1]);
    SmallDft<N>::run(x);
    for (int i = 0; i < N; ++i) {
        out[2 * i] = x[i].x;
        out[2 * i + 1] = x[i].y;
    }
}

static bool g_allow_pz = true, g_allow_raw = true, g_fast_shift = false;
static int g_rows_per_chunk = 1;

extern "C" {

// the TETRA-mode plan's matched-filter operand table (tetra_taps.hpp), as the library uploads it
int64_t emu_tetra_tap_operands(const float *taps, int ntaps, uint32_t *out)
{
    if (out) tdm::tetra_tap_operands(taps, ntaps, out);
    return (int64_t)tdm::tetra_tap_operand_words(ntaps);
}

// tests: 0 forces the cascade engine for every decimation factor (the generic fallback of the library)
void emu_allow_parallel_form(int on) { g_allow_pz = on != 0; }
// tests: 0 keeps cu8 plans on the kernel that holds its samples as doubles
void emu_allow_raw_integer(int on) { g_allow_raw = on != 0; }
// tests: 1 = the plan option "fast_pre_shift" (the input-rate shift's phase as the ideal ramp)
void emu_fast_pre_shift(int on) { g_fast_shift = on != 0; }
void emu_rows_per_chunk(int c) { g_rows_per_chunk = c > 0 ? c : 1; }

// whole pipeline == tdm_process with host pointers
int emu_process(double sample_rate, int64_t n, int rows, int fmt, const void *iq, int64_t stride,
                const double *pre_shift, const double *freq_offset, uint8_t *hard, double *soft,
                int32_t *n_soft, int32_t *best_phase, double *min_margin, int32_t *max_soft_out)
{
    RefPlanHost h = build_ref_plan(sample_rate, n, 25000.0, g_allow_pz, g_allow_raw ? fmt : -1);
    if (max_soft_out) *max_soft_out = (int32_t)h.max_soft;
    if (!iq) return 0;  // query only
    HostZp dec, lpf, dec_raw;
    RefBuffers B;
    if (h.decimated) { dec.t = h.dec; dec.bind(rows); B.dec_params = dec.t.p; }
    if (h.lpf && !h.lp2.ok) { lpf.t = h.lpf_t; lpf.bind(rows); B.lpf_params = lpf.t.p; }
    std::vector<double> zt, lp2p;
    if (h.lp2.ok) {
        B.lp2 = h.lp2.p;
        zt.assign((size_t)rows * h.sps * B.lp2.zt_k * 2 + 2, std::numeric_limits<double>::quiet_NaN());
        lp2p.assign((size_t)rows * B.lp2.n_chunks * kMaxSps + 2, std::numeric_limits<double>::quiet_NaN());
        B.lp2.zt = zt.data();
        B.lp2.partials = lp2p.data();
        B.lp2.lane_m = h.lp2.lane_m.data();
        B.lp2.cst = h.lp2.cst.data();
        B.lp2.seeds = h.lp2.seeds.data();
        B.lp2.items = (const int32_t *)h.lp2.items.data();
        if (h.raw_S) {
            dec_raw.t = h.dec_raw;
            dec_raw.bind(rows);
            B.dec_raw_params = dec_raw.t.p;
            B.lp2_raw = h.lp2_raw.p;
            if (B.lp2_raw.n_chunks > B.lp2.n_chunks)
                lp2p.assign((size_t)rows * B.lp2_raw.n_chunks * kMaxSps + 2, std::numeric_limits<double>::quiet_NaN());
            B.lp2.partials = lp2p.data();
            B.lp2_raw.zt = zt.data();
            B.lp2_raw.partials = lp2p.data();
            B.lp2_raw.lane_m = h.lp2.lane_m.data();
            B.lp2_raw.cst = h.lp2_raw.cst.data();
            B.lp2_raw.seeds = h.lp2_raw.seeds.data();
            B.lp2_raw.items = (const int32_t *)h.lp2_raw.items.data();
        }
    }
    const double nan = std::numeric_limits<double>::quiet_NaN();
    std::vector<double> y((size_t)rows * h.n_dec * 2 + 2, nan), z((size_t)rows * h.n_dec * 2 + 2, nan);
    B.y = y.data();
    B.z = z.data();
    std::vector<double> partials((size_t)rows * (h.n_dec / kPowThreads + 16) * kMaxSps, nan);
    B.partials = partials.data();
    RefIO io{iq, stride, pre_shift, freq_offset, hard, soft, n_soft, best_phase, min_margin, g_fast_shift ? 1 : 0, g_rows_per_chunk};
    EmuBackend be;
    run_ref(be, h, rows, fmt, B, io);
    return 0;
}

// one zero-phase stage on c128 data: kind 0 = decimator sos (q), kind 1 = butter tf (bandwidth, fs)
int emu_zp_stage(int kind, const double *x, int64_t n, int q, double bandwidth, double fs, double *y)
{
    EmuBackend be;
    HostZp hz;
    RawLoader<FMT_CF64, false> ld{x, n, nullptr, fs};
    if (kind == 0) {
        if (n <= kEdgeSos) return -1;
        Sos4 s = design_cheby1_8(0.05, 0.8 / q);
        int64_t n_out = (n + q - 1) / q;
        const int S = g_allow_pz ? pz_outputs_per_lane(q) : 0;
        if (S) {
            RefPlanHost h;
            h.q = q;
            h.dec = build_pz_tables(s.sos, 4, n, kEdgeSos, q * S, S, n_out, q);
            hz.t = h.dec;
            hz.bind(1);
            h.dec.p = hz.t.p;
            RawLoaderRT<false> lr{x, n, nullptr, fs, FMT_CF64, 0};
            run_pz_block(be, h, hz.t.p, lr, 1);
        } else {
            hz.t = build_zp_tables(desc_from_sos(s), n, kEdgeSos, kLDec, n_out, q);
            hz.bind(1);
            be.zp_block<2, 4, kLDec, kEdgeSos>(hz.t.p, ld, hz.t.p.nb, 1);
        }
        be.zp_carry<2, 4>(hz.t.p, hz.t.p.nb, 1);
        be.zp_fixup<8, kLDec>(hz.t.p, hz.t.p.nb, 1, y, n_out, nullptr, fs);
    } else {
        if (n <= kEdgeTf) return -1;
        Tf4 t = design_butter4(butter_cutoff(bandwidth, fs));
        hz.t = build_zp_tables(desc_from_tf(t), n, kEdgeTf, kLLpf, n, 1);
        hz.bind(1);
        StagedLoader<PlainC128Src> ls{{x, n}};
        be.zp_block<2, 2, kLLpf, kEdgeTf>(hz.t.p, ls, hz.t.p.nb, 1);
        be.zp_carry<2, 2>(hz.t.p, hz.t.p.nb, 1);
        be.zp_fixup<4, kLLpf>(hz.t.p, hz.t.p.nb, 1, y, n, nullptr, fs);
    }
    return 0;
}

int emu_resample(const double *x, int64_t n, int64_t num, double *y)
{
    ResamplePlan rp = build_resample_plan(n, num);
    std::vector<double> X(rp.src_bins.size() * 2 + 2);
    for (size_t o = 0; o < rp.src_bins.size(); ++o)
        run_group(8, [&](int t, Group *g) {
            EmuBlockComm cm{g, t};
            dft_terms_body(cm, (int64_t)o, rp.src_bins.data(), x, n, nullptr, nullptr, nullptr, n, -1.0, 1.0, X.data());
        });
    for (int64_t o = 0; o < num; ++o)
        run_group(8, [&](int t, Group *g) {
            EmuBlockComm cm{g, t};
            dft_terms_body(cm, o, nullptr, X.data(), (int64_t)rp.term_src.size(), rp.term_src.data(),
                           rp.term_dst.data(), rp.term_w.data(), num, 1.0, 1.0 / (double)n, y);
        });
    return 0;
}

int emu_find_sync(const uint8_t *units, int64_t n_units, int from_bits, double threshold, int max_pos,
                  int32_t *positions, int32_t *n_pos, double *max_corr)
{
    const int64_t n_bits = from_bits ? n_units : 2 * n_units;
    std::vector<uint16_t> counts((size_t)n_bits + 1, 0xffff);
    for (int64_t pos = 0; pos < n_bits; ++pos) sync_count_body(units, n_bits, pos, from_bits, counts.data());
    *n_pos = sync_walk_body(counts.data(), n_bits, threshold, positions, max_pos, max_corr);
    return 0;
}

int emu_gate(const void *iq, int64_t n, int rows, int fmt, double fs, double *out, double *afc)
{
    GateArgs A{iq, n, n, fmt, 0, fs, out, afc};
    for (int row = 0; row < rows; ++row)
        run_group(32, [&](int t, Group *g) {
            EmuBlockComm cm{g, t};
            gate_body(A, cm, row);
        });
    return 0;
}

int emu_detect(const double *x, int64_t n, double fs, double *out)
{
    std::vector<double> ang((size_t)n + 1);
    std::vector<uint8_t> bits((size_t)n + 1);
    DetectArgs A{x, n, fs, -85.0, ang.data(), bits.data(), out};
    run_group(16, [&](int t, Group *g) {
        EmuBlockComm cm{g, t};
        detect_body(A, cm, 0);
    });
    return 0;
}

int emu_carry_terms(double sample_rate, int64_t n, int32_t *dec_terms, int32_t *lpf_terms, int32_t *nb_dec)
{
    RefPlanHost h = build_ref_plan(sample_rate, n);
    *dec_terms = h.decimated ? h.dec.p.carry_terms : 0;
    *lpf_terms = (h.lpf && !h.lp2.ok) ? h.lpf_t.p.carry_terms : 0;   // (the one-kernel low-rate stage has no carries)
    *nb_dec = h.decimated ? h.dec.p.nb : 0;
    return 0;
}
int emu_small_dft(int n, const float *in, float *out)
{
    switch (n) {
    case 2: small_dft_host<2>(in, out); break;
    case 3: small_dft_host<3>(in, out); break;
    case 4: small_dft_host<4>(in, out); break;
    case 5: small_dft_host<5>(in, out); break;
    case 8: small_dft_host<8>(in, out); break;
    case 9: small_dft_host<9>(in, out); break;
    case 10: small_dft_host<10>(in, out); break;
    case 12: small_dft_host<12>(in, out); break;
    case 16: small_dft_host<16>(in, out); break;
    case 20: small_dft_host<20>(in, out); break;
    case 25: small_dft_host<25>(in, out); break;
    default: return -1;
    }
    return 0;
}
}
