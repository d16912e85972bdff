// Launch sequence of the reference-parity pipeline (SignalProcessor.process, processor.py:221-273),
// written against a Backend so that the gfx950 build (tdm_hip.hip) and the CPU lock-step test
// harness (tests/emul) run the SAME sequence of the SAME kernel bodies.
//
// Backend concept (all calls enqueue asynchronously on the backend's stream):
//   (ZpParams are passed by value with their pointers already valid where the kernels run)
//   template<int K,int NSEC,int L,int EDGE,class Loader> void zp_block(const ZpParams&, Loader, int nb, int rows);
//   template<int K,int NSEC> void zp_carry(const ZpParams&, int nb, int rows);
//   template<int D,int L> void zp_fixup(const ZpParams&, int nb, int rows, double* out,
//                                       int64_t out_row_stride, const double* freq_offset, double fs_out);
//   template<class Loader> void convert(Loader, int rows, int64_t n, double* out,
//                                       const double* freq_offset, double fs);
//   template<int D,int L> void power_fixup(const ZpParams&, int rows, int64_t n, double* z, int sps,
//                                           double* partials, int n_pblk);
//   void finish(const FinishArgs&, int rows);
//   template<class Src> void lp2_finish(const Lp2Params&, const Src&, const FinishArgs&, int rows);   (low-rate kernel, then finish)
#pragma once
#include "ref_plan.hpp"
#include "zp_kernels.hpp"
#include "pz_kernels.hpp"
#include "lp2_kernels.hpp"

namespace tdm {

struct RefBuffers {
    ZpParams dec_params{};  // h.dec.p / h.lpf_t.p with table + work pointers bound for the backend
    ZpParams lpf_params{};
    double *y = nullptr;  // [rows][n_dec] c128: decimated (+freq_offset) signal
    double *z = nullptr;  // [rows][n_dec] c128: channel-filtered signal
    double *partials = nullptr;  // [rows][ceil(n_dec/kPowThreads)][kMaxSps] partial phase powers
    Lp2Params lp2{};             // (h.lp2.ok) pointers bound for the backend
    ZpParams dec_raw_params{};   // (h.raw_S) the raw-integer decimator's geometry and its low-rate stage
    Lp2Params lp2_raw{};
};

struct RefIO {
    const void *iq;
    int64_t carrier_stride;
    const double *pre_shift;
    const double *freq_offset;
    uint8_t *hard;
    double *soft;
    int32_t *n_soft;
    int32_t *best_phase;
    double *min_margin;
    int32_t fast_shift = 0;   // pre_shift as the ideal phase ramp (RawLoaderRT::fast_shift)
    int32_t rows_per_chunk = 1;   // > 1: that many consecutive plan rows read the same input row (tdm_plan_option "rows_per_chunk")
};

// parallel-form decimator: one kernel per decimation factor, wire format as a run-time switch
template <class BE, bool SHIFT>
void run_pz_block(BE &be, const RefPlanHost &h, const ZpParams &P, const RawLoaderRT<SHIFT> &ld, int rows)
{
    const int nb = h.dec.p.nb;
    switch (h.q) {
#define TDM_PZ_CASE(Q, S) case Q: be.template pz_block<Q, S, kEdgeSos>(P, ld, nb, rows); break;
        TDM_PZ_CASES(TDM_PZ_CASE)
#undef TDM_PZ_CASE
    default: break;   // (build_ref_plan only sets pz_S for the factors above)
    }
}

// raw-integer decimator (cu8): blocks that hold no extension sample run on the bytes as they are, the first block and
// the block(s) with the tail extension (from b_tail on) on int16 pairs
template <class BE>
void run_pz_raw(BE &be, const RefPlanHost &h, const ZpParams &P, const void *iq, int64_t stride, int rows)
{
    const int nb = P.nb, Bn = kWave * P.L;
    int b_tail = (int)((P.k0L + P.n) / Bn);   // block of the first position past the signal
    if (b_tail > nb - 1) b_tail = nb - 1;
    switch (h.q) {
#define TDM_PZR_CASE(Q, S) case Q: be.template pz_raw<Q, S, kEdgeSos, FMT_CU8>(P, iq, stride, b_tail, rows); break;
        TDM_PZR_CASES(TDM_PZR_CASE)
#undef TDM_PZR_CASE
    default: break;
    }
}

template <class BE, int FMT, bool SHIFT>
void run_ref_fmt(BE &be, const RefPlanHost &h, int rows, const RefBuffers &B, const RefIO &io)
{
    RawLoader<FMT, SHIFT> ld{io.iq, io.carrier_stride, io.pre_shift, h.sample_rate, io.rows_per_chunk};
    // (the raw-byte kernel addresses its rows itself: a plan whose rows share input rows stays on the loaders)
    const bool use_raw = h.raw_S > 0 && FMT == FMT_CU8 && !SHIFT && io.rows_per_chunk <= 1 && (int64_t)rows * h.dec.p.nb >= h.raw_min_blocks;
    if (use_raw) {
        run_pz_raw(be, h, B.dec_raw_params, io.iq, io.carrier_stride, rows);
        be.template zp_carry<2, 4>(B.dec_raw_params, h.dec_raw.p.nb, rows);
    } else if (h.decimated) {
        // scipy.signal.decimate(samples, q)  (processor.py:254): block-local part + carries
        if (h.pz_S) {
            RawLoaderRT<SHIFT> lr{io.iq, io.carrier_stride, SHIFT ? io.pre_shift : nullptr, h.sample_rate, FMT, io.fast_shift, io.rows_per_chunk};
            run_pz_block(be, h, B.dec_params, lr, rows);
        } else {
            be.template zp_block<2, 4, kLDec, kEdgeSos>(B.dec_params, ld, h.dec.p.nb, rows);
        }
        be.template zp_carry<2, 4>(B.dec_params, h.dec.p.nb, rows);
        if (!h.lpf)  // (n_dec <= 15) nothing downstream finishes the decimator output: do it here
            be.template zp_fixup<8, kLDec>(B.dec_params, h.dec.p.nb, rows, B.y, h.n_dec, io.freq_offset, h.rate_dec);
    } else {
        be.convert(ld, rows, h.n, B.y, io.freq_offset, h.sample_rate);
    }
    // extract_symbols + demodulate_dqpsk  (processor.py:267-271)
    FinishArgs fa{};
    fa.z = B.y;
    fa.n = h.n_dec;
    fa.row_stride = h.n_dec;
    fa.sps = h.sps;
    fa.do_extract = 1;
    fa.do_demod = 1;
    fa.max_soft = (int32_t)h.max_soft;
    fa.soft = io.soft;
    fa.hard = io.hard;
    fa.n_soft = io.n_soft;
    fa.best_phase = io.best_phase;
    fa.min_margin = io.min_margin;
    fa.smear = (h.decimated || h.lpf) ? 1 : 0;   // a zero-phase filter ran: one non-finite sample is the reference's all-NaN chunk
    if (h.lp2.ok) {
        // fix-up + frequency_shift + filter_signal + phase powers in one kernel, output final and phase-major; then the
        // finish stage
        const Lp2Params &L = use_raw ? B.lp2_raw : B.lp2;
        fa.partials = L.partials;
        fa.n_pblk = L.n_chunks;
        fa.zt = B.lp2.zt;
        fa.zt_k = B.lp2.zt_k;
        if (use_raw) {
            Lp2SrcDec src{B.dec_raw_params, io.freq_offset, h.rate_dec};
            be.lp2_finish(L, src, fa, rows);
        } else if (h.decimated) {
            Lp2SrcDec src{B.dec_params, io.freq_offset, h.rate_dec};
            be.lp2_finish(L, src, fa, rows);
        } else {
            Lp2SrcPlain src{B.y, h.n_dec};
            be.lp2_finish(L, src, fa, rows);
        }
        return;
    }
    if (h.lpf) {
        // filter_signal(samples, 25000, current_rate)  (processor.py:264).  When decimated, the
        // loader finishes the decimator output (carry responses) and applies
        // frequency_shift(samples, freq_offset, current_rate) (processor.py:260-261) on the fly.
        if (h.decimated) {
            StagedLoader<DecFixSrc<kLDec>> l2{{B.dec_params, io.freq_offset, h.rate_dec}};
            be.template zp_block<2, 2, kLLpf, kEdgeTf>(B.lpf_params, l2, h.lpf_t.p.nb, rows);
        } else {
            StagedLoader<PlainC128Src> l2{{B.y, h.n_dec}};
            be.template zp_block<2, 2, kLLpf, kEdgeTf>(B.lpf_params, l2, h.lpf_t.p.nb, rows);
        }
        be.template zp_carry<2, 2>(B.lpf_params, h.lpf_t.p.nb, rows);
        if (h.sps > 1 && h.sps <= kMaxSps) {
            fa.n_pblk = h.lpf_t.p.nb * (kWave * kLLpf / kPowThreads);
            // z itself is not written: the finish stage evaluates it at the symbols it gathers
            be.template power_fixup<4, kLLpf>(B.lpf_params, rows, h.n_dec, nullptr, h.sps, B.partials, fa.n_pblk);
            fa.partials = B.partials;
            fa.use_fix = 1;
            fa.fix = B.lpf_params;
        } else {
            be.template zp_fixup<4, kLLpf>(B.lpf_params, h.lpf_t.p.nb, rows, B.z, h.n_dec, nullptr, h.rate_dec);
        }
        fa.z = B.z;
    }
    static_assert(kFixBn == kWave * kLLpf, "finish evaluates the channel filter's fix-up with its block length");
    be.finish(fa, rows);
}

template <class BE>
void run_ref(BE &be, const RefPlanHost &h, int rows, int fmt, const RefBuffers &B, const RefIO &io)
{
    const bool sh = io.pre_shift != nullptr;
    switch (fmt) {
    case FMT_CU8: sh ? run_ref_fmt<BE, FMT_CU8, true>(be, h, rows, B, io) : run_ref_fmt<BE, FMT_CU8, false>(be, h, rows, B, io); break;
    case FMT_CS8: sh ? run_ref_fmt<BE, FMT_CS8, true>(be, h, rows, B, io) : run_ref_fmt<BE, FMT_CS8, false>(be, h, rows, B, io); break;
    case FMT_CF32: sh ? run_ref_fmt<BE, FMT_CF32, true>(be, h, rows, B, io) : run_ref_fmt<BE, FMT_CF32, false>(be, h, rows, B, io); break;
    default: sh ? run_ref_fmt<BE, FMT_CF64, true>(be, h, rows, B, io) : run_ref_fmt<BE, FMT_CF64, false>(be, h, rows, B, io); break;
    }
}

// convert body: raw sample (+ input-rate pre-shift) then process()'s freq_offset at rate fs
template <class Loader>
TDM_HD void convert_body(const Loader &ld, int row, int64_t j, double *out_row, const double *freq_offset,
                         double fs)
{
    double re, im;
    const double f0 = ld.row_shift(row);
    ld.sample(ld.row_ptr(row), j, f0, re, im);
    if (freq_offset) {
        const double f = freq_offset[row];
        if (f != 0.0) nco_rotate(re, im, j, f, fs);
    }
    out_row[j * 2] = re;
    out_row[j * 2 + 1] = im;
}

}  // namespace tdm
