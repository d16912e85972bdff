// Kernel bodies of the reference-parity path, written once and compiled twice:
//   * by hipcc for gfx950 (tdm_hip.hip wraps each body in a __global__ kernel; Comm = wavefront
//     shuffles / LDS reductions), and
//   * by g++ for the CPU test harness tests/emul (Comm = std::barrier lock-step emulation), so
//     the index/table logic can be checked against the oracle without a GPU.  The CPU build is
//     test infrastructure only; the product library has no CPU path.
//
// Reference functions restated here (tetraear/signal/processor.py):
//   zp_block_body / zp_carry_body / zp_fixup_body : scipy sosfiltfilt+[::q] (processor.py:254)
//                                                    and filtfilt (processor.py:79)
//   nco_rotate                                     : frequency_shift (processor.py:85-100)
//   finish_body                                    : extract_symbols (processor.py:168-219) +
//                                                    demodulate_dqpsk (processor.py:102-166)
#pragma once
#include <math.h>

#include "zp_common.hpp"

namespace tdm {

// ------------------------------------------------------------------------------------------
// exact-rounding helpers (no FMA contraction) for the few places where the reference's own
// rounding sequence defines the data (cu8 -> float conversion, slicer products)
// ------------------------------------------------------------------------------------------
// (HIP's __dmul_rn / __dsub_rn are plain operators and get contracted into one v_fma_f64 with a neighbouring
// operation under the default -ffp-contract=fast-honor-pragmas; the pragma is what keeps the two roundings)
TDM_HD double mul_rn(double a, double b)
{
#pragma clang fp contract(off)
    return a * b;
}
TDM_HD double add_rn(double a, double b)
{
#pragma clang fp contract(off)
    return a + b;
}
TDM_HD double sub_rn(double a, double b)
{
#pragma clang fp contract(off)
    return a - b;
}

#if defined(__HIPCC__)
#define TDM_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define TDM_NOINLINE __attribute__((noinline))
#endif

struct alignas(16) u32x4 {
    uint32_t x, y, z, w;
};
struct alignas(16) f64x2 {
    double x, y;
};

// frequency_shift (processor.py:98-99): t = n/fs; shift = exp(-1j*2*pi*f*t)
// Python evaluates ((-1j*2)*pi)*f -> (0, -(2*pi)*f) and multiplies by t[n].
// The phasor is computed out of line (by value, so callers' sample arrays stay in registers).
struct phasor {
    double c, s;
};
inline TDM_NOINLINE phasor nco_phasor(int64_t k, double f, double fs)
{
    const double ci = -(2.0 * M_PI) * f;
    const double t = (double)k / fs;
    const double th = ci * t;
    phasor p;
    sincos(th, &p.s, &p.c);
    return p;
}
TDM_HD void nco_rotate(double &re, double &im, int64_t k, double f, double fs)
{
    const phasor p = nco_phasor(k, f, fs);
    const double a = re, b = im;
    re = a * p.c - b * p.s;
    im = a * p.s + b * p.c;
}

// Running NCO for a lane that visits j_a, j_a+64, j_a+128, ...: the reference's phase is
// theta_j = fl(ci * fl(j / fs)) (processor.py:98-99).  One exact sincos anchors the lane; after that
// exp(i theta_j) = A * W^k * (1 + i eps_j), W = exp(i 64 Dd) advanced by complex multiplication and
// eps_j = (theta_j - theta_a) - (j - j_a) Dd, computed exactly (Dd has 40 significant bits so the
// product is exact; the difference of neighbouring thetas is exact), |eps| ~ 1e-11 so eps^2 drops.
// This reproduces the reference's own rounding of theta_j, not an idealised phase ramp.
template <int STRIDE>
struct NcoRunT {
    double ar = 1, ai = 0;   // anchor phasor (cos, sin)(theta_a)
    double wr = 1, wi = 0;   // W^k
    double sr = 1, si = 0;   // W = exp(i * STRIDE * Dd)
    double th_a = 0, dd = 0, rfs = 0;
    int64_t j_a = 0, j_cur = 0;
    bool on = false;
    // fl(a / b) from r = fl(1 / b) by two residual corrections (Markstein): 5 operations instead of the
    // ~15 of a division; equal to the IEEE quotient for every j < 2^26 at the sample rates the plans
    // produce (checked exhaustively on the host) and under the theorem's conditions in general
    TDM_HD static double quot(double a, double b, double r)
    {
        const double q0 = a * r;
        const double q1 = fma(fma(-q0, b, a), r, q0);
        return fma(fma(-q1, b, a), r, q1);
    }
    // step() split in two for callers that visit consecutive samples and want the (out-of-line) sincos calls
    // made while few registers are live: init() anchors the phasor at j, next() advances to j_cur + STRIDE.
    // Same arithmetic as step(), term for term.
    TDM_HD void init(int64_t j, double f, double fs)
    {
        const double ci = -(2.0 * M_PI) * f;
        rfs = 1.0 / fs;
        union { double d; uint64_t u; } v;
        v.d = ci / fs;
        v.u &= ~uint64_t(0x1FFF);
        dd = v.d;
        sincos((double)STRIDE * dd, &si, &sr);
        const double t = quot((double)(int32_t)j, fs, rfs);
        th_a = ci * t;
        const phasor p = nco_phasor(j, f, fs);
        ar = p.c; ai = p.s; wr = 1; wi = 0; j_a = j; j_cur = j;
        on = true;
    }
    // the step phasor W = exp(i STRIDE Dd) depends on (f, fs) only: a caller whose lanes share them forms it once
    // (step_phasor) and hands it to every lane's init_with
    TDM_HD static void step_phasor(double f, double fs, double &w_re, double &w_im)
    {
        const double ci = -(2.0 * M_PI) * f;
        union { double d; uint64_t u; } v;
        v.d = ci / fs;
        v.u &= ~uint64_t(0x1FFF);
        sincos((double)STRIDE * v.d, &w_im, &w_re);
    }
    TDM_HD void init_with(int64_t j, double f, double fs, double w_re, double w_im)
    {
        const double ci = -(2.0 * M_PI) * f;
        rfs = 1.0 / fs;
        union { double d; uint64_t u; } v;
        v.d = ci / fs;
        v.u &= ~uint64_t(0x1FFF);
        dd = v.d;
        sr = w_re;
        si = w_im;
        const double t = quot((double)(int32_t)j, fs, rfs);
        th_a = ci * t;
        const phasor p = nco_phasor(j, f, fs);
        ar = p.c; ai = p.s; wr = 1; wi = 0; j_a = j; j_cur = j;
        on = true;
    }
    TDM_HD void next(double f, double fs, double &c, double &s)
    {
        const double ci = -(2.0 * M_PI) * f;
        const int64_t j = j_cur + STRIDE;
        const double t = quot((double)(int32_t)j, fs, rfs);
        const double th = ci * t;
        j_cur = j;
        const double nwr = wr * sr - wi * si, nwi = wr * si + wi * sr;
        wr = nwr;
        wi = nwi;
        const double eps = (th - th_a) - (double)(int32_t)(j - j_a) * dd;
        const double pr = ar * wr - ai * wi, pi_ = ar * wi + ai * wr;
        c = pr - eps * pi_;
        s = pi_ + eps * pr;
    }
    TDM_HD void step(int64_t j, double f, double fs, double &c, double &s)
    {
        const double ci = -(2.0 * M_PI) * f;
        if (!on) {
            rfs = 1.0 / fs;
            // Dd: ci/fs with the low 13 mantissa bits cleared -> (j - j_a) * Dd is exact
            union { double d; uint64_t u; } v;
            v.d = ci / fs;
            v.u &= ~uint64_t(0x1FFF);
            dd = v.d;
            sincos((double)STRIDE * dd, &si, &sr);
        }
        const double t = quot((double)(uint32_t)j, fs, rfs);   // j < 2^32: tdm_plan_create bounds the chunk length
        const double th = ci * t;
        if (!on || j != j_cur + STRIDE || j - j_a > 4096) {
            const phasor p = nco_phasor(j, f, fs);
            ar = p.c; ai = p.s; wr = 1; wi = 0; th_a = th; j_a = j; j_cur = j;
            on = true;
            c = ar;
            s = ai;
            return;
        }
        j_cur = j;
        const double nwr = wr * sr - wi * si, nwi = wr * si + wi * sr;
        wr = nwr;
        wi = nwi;
        const double eps = (th - th_a) - (double)(int32_t)(j - j_a) * dd;
        const double pr = ar * wr - ai * wi, pi_ = ar * wi + ai * wr;
        c = pr - eps * pi_;
        s = pi_ + eps * pr;
    }
};
typedef NcoRunT<kWave> NcoRun;   // a lane of the staged loader visits j, j + 64, j + 128, ...

// ------------------------------------------------------------------------------------------
// Loaders: give a lane its L consecutive samples of the padded, odd-extended signal.
// ------------------------------------------------------------------------------------------
enum { FMT_CU8 = 0, FMT_CS8 = 1, FMT_CF32 = 2, FMT_CF64 = 3 };

template <int FMT>
TDM_HD void convert_one(const void *rowp, int64_t k, double &re, double &im)
{
    if (FMT == FMT_CU8) {
        // pyrtlsdr: iq = bytes.astype(float64).view(complex128); iq /= 127.5; iq -= (1+1j)
        // numpy's complex/real division multiplies by fl(1/127.5): two roundings, no FMA.
        const uint8_t *p = (const uint8_t *)rowp + 2 * k;
        const double c = 1.0 / 127.5;
        re = sub_rn(mul_rn((double)p[0], c), 1.0);
        im = sub_rn(mul_rn((double)p[1], c), 1.0);
    } else if (FMT == FMT_CS8) {
        const int8_t *p = (const int8_t *)rowp + 2 * k;
        re = (double)p[0] * (1.0 / 128.0);
        im = (double)p[1] * (1.0 / 128.0);
    } else if (FMT == FMT_CF32) {
        const float *p = (const float *)rowp + 2 * k;
        re = (double)p[0];
        im = (double)p[1];
    } else {
        const double *p = (const double *)rowp + 2 * k;
        re = p[0];
        im = p[1];
    }
}

template <int FMT, bool SHIFT>
struct RawLoader {
    const void *iq;            // first sample of row 0
    int64_t row_stride;        // samples between rows (0 = shared stream)
    const double *pre_shift;   // per row [Hz]; read only when SHIFT
    double fs;
    int32_t rows_per_chunk;    // > 1: that many consecutive plan rows read the same input row (RawLoaderRT::rows_per_chunk)

    static constexpr int kBytes = (FMT == FMT_CU8 || FMT == FMT_CS8) ? 2 : (FMT == FMT_CF32 ? 8 : 16);
    static constexpr bool kStaged = false;  // lanes load their own segment straight from memory

    TDM_HD const void *row_ptr(int row) const
    {
        return (const char *)iq + (int64_t)(rows_per_chunk > 1 ? row / rows_per_chunk : row) * row_stride * kBytes;
    }
    TDM_HD double row_shift(int row) const { return (SHIFT && pre_shift) ? pre_shift[row] : 0.0; }
    TDM_HD void sample(const void *rowp, int64_t k, double f, double &re, double &im) const
    {
        convert_one<FMT>(rowp, k, re, im);
        if (SHIFT && f != 0.0) nco_rotate(re, im, k, f, fs);
    }

    template <int L>
    TDM_HD void fast(const void *rowp, int64_t k, double f, double *xr, double *xi) const
    {
        const char *p = (const char *)rowp + k * kBytes;
        const bool aligned = (((uintptr_t)p) & 15) == 0;
        if ((FMT == FMT_CU8 || FMT == FMT_CS8) && aligned) {
            const u32x4 *v = (const u32x4 *)p;
#pragma unroll
            for (int c = 0; c < L / 8; ++c) {
                const u32x4 w = v[c];
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t s = ww[d] >> (16 * h);
                        const int idx = c * 8 + d * 2 + h;
                        if (FMT == FMT_CU8) {
                            const double cc = 1.0 / 127.5;
                            xr[idx] = sub_rn(mul_rn((double)(s & 255u), cc), 1.0);
                            xi[idx] = sub_rn(mul_rn((double)((s >> 8) & 255u), cc), 1.0);
                        } else {
                            xr[idx] = (double)(int8_t)(s & 255u) * (1.0 / 128.0);
                            xi[idx] = (double)(int8_t)((s >> 8) & 255u) * (1.0 / 128.0);
                        }
                    }
            }
        } else if (FMT == FMT_CF64 && aligned) {
            const f64x2 *v = (const f64x2 *)p;
#pragma unroll
            for (int i = 0; i < L; ++i) {
                const f64x2 w = v[i];
                xr[i] = w.x;
                xi[i] = w.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < L; ++i) convert_one<FMT>(rowp, k + i, xr[i], xi[i]);
        }
        if (SHIFT && f != 0.0) {
            // frequency_shift of the shared stream (processor.py:85-100) with a running phasor over the
            // lane's consecutive samples: one exact sincos per lane and block instead of one per sample
            NcoRunT<1> nco;
#pragma unroll
            for (int i = 0; i < L; ++i) {
                double c, sn;
                nco.step(k + i, f, fs, c, sn);
                const double a = xr[i], b = xi[i];
                xr[i] = a * c - b * sn;
                xi[i] = a * sn + b * c;
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_sched_barrier(0);   // keep the 32 steps from being interleaved (register pressure)
#endif
            }
        }
    }

    // Edge lanes (odd extension, zero pad): rolled loop into a small stack array, so the main
    // kernel's register budget is not shaped by this rarely taken path.
    TDM_NOINLINE void slow(const void *rowp, double f, int64_t e0, int64_t n, int edge, int L, double *out) const
    {
        double x0r = 0, x0i = 0, x1r = 0, x1i = 0;
        if (n > 0) {
            sample(rowp, 0, f, x0r, x0i);
            sample(rowp, n - 1, f, x1r, x1i);
        }
        for (int i = 0; i < L; ++i) {
            const int64_t e = e0 + i;
            double re = 0, im = 0;
            if (e >= 0 && e < n + 2 * (int64_t)edge) {
                if (e < edge) {  // 2*x[0] - x[edge - e]
                    sample(rowp, edge - e, f, re, im);
                    re = 2 * x0r - re;
                    im = 2 * x0i - im;
                } else if (e < edge + n) {
                    sample(rowp, e - edge, f, re, im);
                } else {  // 2*x[n-1] - x[n-2-(e-edge-n)]
                    sample(rowp, n - 2 - (e - edge - n), f, re, im);
                    re = 2 * x1r - re;
                    im = 2 * x1i - im;
                }
            }
            out[2 * i] = re;
            out[2 * i + 1] = im;
        }
    }

    // x[i] = padded-ext sample seg+i (zero outside [P0, Ne))
    template <int L, class Comm>
    TDM_HD void load(Comm &, int row, int blk, int lane, const ZpParams &P, double *xr, double *xi) const
    {
        const int64_t seg = (int64_t)blk * (kWave * L) + (int64_t)lane * L;
        const void *rowp = row_ptr(row);
        const double f = row_shift(row);
        const int64_t n = P.n;
        const int edge = P.edge;
        const int64_t e0 = seg - P.P0;  // ext index of x[0]
        if (e0 >= edge && e0 + L <= edge + n) {
            fast<L>(rowp, e0 - edge, f, xr, xi);
        } else {
            double tmp[2 * L];
            slow(rowp, f, e0, n, edge, L, tmp);
#pragma unroll
            for (int i = 0; i < L; ++i) { xr[i] = tmp[2 * i]; xi[i] = tmp[2 * i + 1]; }
        }
        const double g = P.in_gain;  // total gain of both passes, applied once (see ZpFilterDesc)
#pragma unroll
        for (int i = 0; i < L; ++i) { xr[i] *= g; xi[i] *= g; }
    }
};

// ------------------------------------------------------------------------------------------
// LDS staging for 16-byte complex samples.  A lane's L consecutive samples are 16*L bytes apart
// from its neighbour's, so direct per-lane loads/stores make 64 cache-line requests per
// instruction; instead the wavefront moves the block with fully coalesced accesses and transposes
// it through LDS.  Slot s (16 bytes) lives at s + s/32: the one-slot pad per 32 makes both the
// row-wise (coalesced side) and the column-wise (lane side) ds_*_b128 accesses conflict-free.
// ------------------------------------------------------------------------------------------
template <int L>
struct StageGeom {
    static constexpr int kSlots = kWave * L + (kWave * L) / 32 + 1;
    static constexpr int kDoubles = 2 * kSlots;
};
TDM_HD int stage_slot(int s) { return s + (s >> 5); }

// sample sources for the staged loader: get(state, row, j) -> sample j of the row, 0 <= j < n.
// `State` lets a source exploit that a lane asks for j, j+64, j+128, ... in turn.
struct PlainC128Src {
    const double *x;
    int64_t row_stride;  // samples
    struct State {};
    TDM_HD void get(State &, int row, int64_t j, double &re, double &im) const
    {
        const f64x2 v = *(const f64x2 *)(x + ((int64_t)row * row_stride + j) * 2);
        re = v.x;
        im = v.y;
    }
    // stage samples j0..j1-1 into LDS slots slot0 + (j - j0), coalesced
    template <class Comm>
    TDM_HD void stage_range(Comm &, int row, int64_t j0, int64_t j1, int lane, f64x2 *lds, int slot0) const
    {
        const f64x2 *p = (const f64x2 *)(x + (int64_t)row * row_stride * 2);
        for (int64_t j = j0 + lane; j < j1; j += kWave) lds[stage_slot(slot0 + (int)(j - j0))] = p[j];
    }
};

template <int D>
TDM_HD void zp_fixup_at(const ZpParams &P, int row, int b, int m, int64_t j, double &re, double &im);
template <int D>
TDM_HD void zp_fixup_rowed(const ZpParams &P, int row, int b, size_t r, int64_t j, double &re, double &im);
TDM_HD size_t zp_fixup_row(const ZpParams &P, int b, int m);
template <int D>
struct FixOperands {   // per-lane operands of one fix-up: table rows and the block-local output
    double t1[D], t2[D], yr, yi;
};
template <int D>
TDM_HD void zp_fixup_load(const ZpParams &P, int row, int b, size_t r, int64_t j, FixOperands<D> &o);
template <int D>
TDM_HD void zp_fixup_apply(const ZpParams &P, int row, int b, const FixOperands<D> &o, double &re, double &im);
template <int D>
TDM_HD void zp_fixup_load_tables(const ZpParams &P, int b, size_t r, FixOperands<D> &o);


// decimator output finished on the fly: block-local y0 + carry responses, then process()'s
// freq_offset NCO (processor.py:260-261) -- the former separate fix-up pass, fused into the load
template <int LDEC>
struct DecFixSrc {
    ZpParams dec;
    const double *freq_offset;  // per row or null
    double fs_out;
    struct State {
        NcoRun nco;
    };
    TDM_HD void get(State &st, int row, int64_t j, double &re, double &im) const
    {
        const int Bn = kWave * dec.L;   // (the decimator's lane length is a plan parameter: cascade 32, parallel form S*q)
        const int64_t pos = dec.k0L + j * dec.out_stride;
        const int b = (int)(pos / Bn);
        const int m = (int)(pos - (int64_t)b * Bn);
        zp_fixup_at<8>(dec, row, b, m, j, re, im);
        if (freq_offset) {
            const double f = freq_offset[row];
            if (f != 0.0) {
                double c, s;
                st.nco.step(j, f, fs_out, c, s);
                const double a = re, bb = im;
                re = a * c - bb * s;
                im = a * s + bb * c;
            }
        }
    }
    // Bulk form.  Phase 1 copies the block-local outputs y0[j0..j1) into the staging buffer with every
    // load of a lane in flight at once (one memory latency per block instead of one per output).
    // Phase 2 walks the decimator blocks that cover [j0, j1) and adds the carry responses in place:
    // inside one block the carries are wave-uniform (scalar loads), consecutive lanes read consecutive
    // table rows, and the rows of a lane's next output are requested before the current one is finished.
    template <class Comm>
    TDM_HD void stage_range(Comm &cm, int row, int64_t j0, int64_t j1, int lane, f64x2 *lds, int slot0) const
    {
        const int Bn = kWave * dec.L;
        const int q = dec.out_stride;
        const double f = freq_offset ? freq_offset[row] : 0.0;
        {
            const f64x2 *y0 = (const f64x2 *)(dec.y0 + (int64_t)row * dec.n_out * 2);
#pragma unroll 4
            for (int64_t j = j0 + lane; j < j1; j += kWave) lds[stage_slot(slot0 + (int)(j - j0))] = y0[j];
        }
        cm.wave_sync();
        int b = (int)((dec.k0L + j0 * q) / Bn);
        for (;; ++b) {
            // outputs of block b: positions pos = k0L + j*q in [b*Bn, (b+1)*Bn)
            const int64_t lo_pos = (int64_t)b * Bn - dec.k0L;
            const int64_t hi_pos = lo_pos + Bn;
            int64_t jl = lo_pos <= 0 ? 0 : (lo_pos + q - 1) / q;
            int64_t jh = (hi_pos + q - 1) / q;
            if (jl < j0) jl = j0;
            if (jh > j1) jh = j1;
            if (jl >= j1 || b >= dec.nb) break;
            NcoRun nco;
            // table row of the lane's first output; the next one (j + 64) is 64 rows further in the same
            // decimation phase, so the division by q happens once per block, not per sample
            size_t r = zp_fixup_row(dec, b, (int)(dec.k0L + (jl + lane) * q - (int64_t)b * Bn));
            // the table rows of the lane's next output are requested before the current one is finished
            FixOperands<8> cur;
            if (jl + lane < jh) zp_fixup_load_tables<8>(dec, b, r, cur);
#pragma unroll 1
            for (int64_t j = jl + lane; j < jh; j += kWave, r += kWave) {
                FixOperands<8> nxt = cur;
                if (j + kWave < jh) zp_fixup_load_tables<8>(dec, b, r + kWave, nxt);
                f64x2 &slot = lds[stage_slot(slot0 + (int)(j - j0))];
                cur.yr = slot.x;
                cur.yi = slot.y;
                double re, im;
                zp_fixup_apply<8>(dec, row, b, cur, re, im);
                cur = nxt;
                if (f != 0.0) {
                    double c, s;
                    nco.step(j, f, fs_out, c, s);
                    const double a = re, bb = im;
                    re = a * c - bb * s;
                    im = a * s + bb * c;
                }
                slot = f64x2{re, im};
            }
        }
    }
};

template <class Src>
struct StagedLoader {
    Src src;
    static constexpr bool kStaged = true;

    template <int L, class Comm>
    TDM_HD void load(Comm &cm, int row, int blk, int lane, const ZpParams &P, double *xr, double *xi) const
    {
        constexpr int Bn = kWave * L;
        f64x2 *lds = (f64x2 *)cm.stage();
        const int64_t n = P.n;
        const int edge = P.edge;
        const int64_t e_blk = (int64_t)blk * Bn - P.P0;  // ext index of the block's first position
        // interior samples (the signal itself), staged in bulk
        {
            int64_t j0 = e_blk - edge, j1 = j0 + Bn;
            const int64_t jb = j0;
            if (j0 < 0) j0 = 0;
            if (j1 > n) j1 = n;
            if (j0 < j1) src.stage_range(cm, row, j0, j1, lane, lds, (int)(j0 - jb));
        }
        // positions in the odd extension or in the zero pad (first / last block only)
        if (e_blk < edge || e_blk + Bn > edge + n) {
#pragma unroll 1
            for (int it = 0; it < L; ++it) {
                const int s = it * kWave + lane;
                const int64_t e = e_blk + s;
                if (e >= edge && e < edge + n) continue;
                double re = 0, im = 0;
                if (e >= 0 && e < n + 2 * (int64_t)edge) {
                    typename Src::State t0{}, t1{};
                    double ar, ai;
                    if (e < edge) {  // 2*x[0] - x[edge - e]
                        src.get(t0, row, 0, ar, ai);
                        src.get(t1, row, edge - e, re, im);
                    } else {  // 2*x[n-1] - x[n-2-(e-edge-n)]
                        src.get(t0, row, n - 1, ar, ai);
                        src.get(t1, row, n - 2 - (e - edge - n), re, im);
                    }
                    re = 2 * ar - re;
                    im = 2 * ai - im;
                }
                lds[stage_slot(s)] = f64x2{re, im};
            }
        }
        cm.wave_sync();
        const double g = P.in_gain;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const f64x2 v = lds[stage_slot(lane * L + i)];
            xr[i] = v.x * g;
            xi[i] = v.y * g;
        }
        cm.wave_sync();  // the output stage reuses the buffer
    }
};

// Zero-input response from the table of unit-state responses, four positions (8 table doubles, one
// s_load_dwordx16) at a time.  Each chunk's pointer passes through an empty asm so that the compiler
// cannot hoist all the loads to the top of the kernel (it then spills hundreds of SGPRs); the pointer of
// chunk c + 1 is made before chunk c is used, which lets its load overlap chunk c's arithmetic.
#if defined(__HIP_DEVICE_COMPILE__)
#define TDM_OPAQUE_SPTR(p) asm volatile("" : "+s"(p))
#else
#define TDM_OPAQUE_SPTR(p)
#endif
#define TDM_ZIR_CHUNK 2   // positions per scalar load (2 coefficients each)
template <int K, int L, bool REV>
TDM_HD void zir_from_table(const double *tab, const double *sr, const double *sq, double *xr, double *xi)
{
    constexpr int CP = TDM_ZIR_CHUNK, CD = CP * 2;
    static_assert(K == 2 && L % CP == 0, "chunks of whole positions of a biquad");
    constexpr int NCH = L / CP;
    const double *p0 = tab;
    TDM_OPAQUE_SPTR(p0);
    double h[CD];
#pragma unroll
    for (int k = 0; k < CD; ++k) h[k] = TDM_CPTR(p0)[k];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        double hn[CD];
        if (c + 1 < NCH) {
            const double *p1 = tab + (c + 1) * CD;
            TDM_OPAQUE_SPTR(p1);
#pragma unroll
            for (int k = 0; k < CD; ++k) hn[k] = TDM_CPTR(p1)[k];
        }
#pragma unroll
        for (int t = 0; t < CP; ++t) {
            const int pos = c * CP + t;
            const int i = REV ? L - 1 - pos : pos;
            xr[i] = fma(h[t * 2], sr[0], fma(h[t * 2 + 1], sr[1], xr[i]));
            xi[i] = fma(h[t * 2], sq[0], fma(h[t * 2 + 1], sq[1], xi[i]));
        }
        if (c + 1 < NCH) {
#pragma unroll
            for (int k = 0; k < CD; ++k) h[k] = hn[k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Block kernel body: one wavefront filters one block forward then backward.
//   Comm: stage() -> workgroup LDS of StageGeom<L>::kDoubles doubles (staged loaders only),
//   wave_sync(); shfl_up2<K>(a, b, outa, outb, d) / shfl_down2<K>: out[lane] = in[lane -/+ d] for two
//   K-vectors of doubles across the 64 lanes (own value where the source lane does not exist).
// ------------------------------------------------------------------------------------------
template <int K, int NSEC, int L, int EDGE, class Loader, class Comm>
TDM_HD void zp_block_body(const ZpParams &P, const Loader &ld, Comm &cm, int lane, int blk, int row)
{
    constexpr int D = K * NSEC;
    constexpr int Bn = kWave * L;
    double xr[L], xi[L];
    const int64_t seg = (int64_t)blk * Bn + (int64_t)lane * L;
    ld.template load<L>(cm, row, blk, lane, P, xr, xi);

    constexpr int P0 = (L - EDGE % L) % L;  // == P.P0
    const bool inject = (blk == 0 && lane == 0);
    const double e0r = xr[P0], e0i = xi[P0];

    const int64_t nbD = (int64_t)P.nb * D;
    double *Ef = P.Ef + ((int64_t)row * nbD + (int64_t)blk * D) * 2;
    double *Eb = P.Eb + ((int64_t)row * nbD + (int64_t)blk * D) * 2;

    // ---------------- forward: sections in cascade order ----------------
#pragma unroll
    for (int s = 0; s < NSEC; ++s) {
        static_assert(K == 2, "device sections are biquads with numerator [1,2,1]");
        double a[K + 1];
#pragma unroll
        for (int k = 0; k <= K; ++k) a[k] = P.a[s][k];
        double zr[K], zq[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { zr[k] = 0; zq[k] = 0; }
#pragma unroll
        for (int i = 0; i < L; ++i) {
            if (i == P0 && inject) {  // scipy: zi * ext[0] is the state before the first sample
#pragma unroll
                for (int k = 0; k < K; ++k) { zr[k] = P.zi[s][k] * e0r; zq[k] = P.zi[s][k] * e0i; }
            }
            xr[i] = lp121_step<double>(a, xr[i], zr);
            xi[i] = lp121_step<double>(a, xi[i], zq);
        }
        // inclusive scan of end states: I_p = sum_{j<=p} M^(p-j) e_j
#pragma unroll
        for (int j = 0; j < kScanSteps; ++j) {
            const int d = 1 << j;
            const auto M = TDM_CPTR(P.Mpow + ((size_t)s * kScanSteps + j) * K * K);
            double jr[K], jq[K];
            cm.template shfl_up2<K>(zr, zq, jr, jq, d);
            if (lane >= d) {
#pragma unroll
                for (int r = 0; r < K; ++r) {
                    double ar = zr[r], aq = zq[r];
#pragma unroll
                    for (int k = 0; k < K; ++k) { ar += M[r * K + k] * jr[k]; aq += M[r * K + k] * jq[k]; }
                    zr[r] = ar;
                    zq[r] = aq;
                }
            }
        }
        if (lane == kWave - 1) {
#pragma unroll
            for (int k = 0; k < K; ++k) { Ef[(s * K + k) * 2] = zr[k]; Ef[(s * K + k) * 2 + 1] = zq[k]; }
        }
        // start state of this lane = inclusive value of lane-1
        double sr[K], sq[K];
        cm.template shfl_up2<K>(zr, zq, sr, sq, 1);
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (lane == 0) { sr[k] = 0; sq[k] = 0; }
        // add the zero-input response of the start state
#pragma unroll
        for (int i = 0; i < L; ++i) {
            xr[i] += zir_step<K>(a, sr);
            xi[i] += zir_step<K>(a, sq);
        }
    }
    // positions past the end of the extended signal must not feed the backward pass
    if (blk == P.nb - 1) {
        const int64_t last = P.Ne - 1;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const int64_t g = seg + i;
            if (g == last) { P.flast[(int64_t)row * 2] = xr[i]; P.flast[(int64_t)row * 2 + 1] = xi[i]; }
            if (g > last) { xr[i] = 0; xi[i] = 0; }
        }
    }
    // ---------------- backward: same cascade, time reversed ----------------
#pragma unroll
    for (int s = 0; s < NSEC; ++s) {
        static_assert(K == 2, "device sections are biquads with numerator [1,2,1]");
        double a[K + 1];
#pragma unroll
        for (int k = 0; k <= K; ++k) a[k] = P.a[s][k];
        double zr[K], zq[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { zr[k] = 0; zq[k] = 0; }
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
            xr[i] = lp121_step<double>(a, xr[i], zr);
            xi[i] = lp121_step<double>(a, xi[i], zq);
        }
#pragma unroll
        for (int j = 0; j < kScanSteps; ++j) {
            const int d = 1 << j;
            const auto M = TDM_CPTR(P.Mpow + ((size_t)s * kScanSteps + j) * K * K);
            double jr[K], jq[K];
            cm.template shfl_down2<K>(zr, zq, jr, jq, d);
            if (lane + d < kWave) {
#pragma unroll
                for (int r = 0; r < K; ++r) {
                    double ar = zr[r], aq = zq[r];
#pragma unroll
                    for (int k = 0; k < K; ++k) { ar += M[r * K + k] * jr[k]; aq += M[r * K + k] * jq[k]; }
                    zr[r] = ar;
                    zq[r] = aq;
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) { Eb[(s * K + k) * 2] = zr[k]; Eb[(s * K + k) * 2 + 1] = zq[k]; }
        }
        double sr[K], sq[K];
        cm.template shfl_down2<K>(zr, zq, sr, sq, 1);
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (lane == kWave - 1) { sr[k] = 0; sq[k] = 0; }
#pragma unroll
        for (int i = L - 1; i >= 0; --i) {
            xr[i] += zir_step<K>(a, sr);
            xi[i] += zir_step<K>(a, sq);
        }
    }
    // ---------------- block-local outputs at padded-ext positions k0L + j*stride ----------------
    if (Loader::kStaged) {
        // stride-1 stage: transpose back through LDS and store 16 B per lane, coalesced
        f64x2 *lds = (f64x2 *)cm.stage();
#pragma unroll
        for (int i = 0; i < L; ++i) lds[stage_slot(lane * L + i)] = f64x2{xr[i], xi[i]};
        cm.wave_sync();
        f64x2 *y0 = (f64x2 *)(P.y0 + (int64_t)row * P.n_out * 2);
        const int64_t j_blk = (int64_t)blk * Bn - P.k0L;
#pragma unroll 1
        for (int it = 0; it < L; ++it) {
            const int s = it * kWave + lane;
            const int64_t j = j_blk + s;
            if (j >= 0 && j < P.n_out) y0[j] = lds[stage_slot(s)];
        }
    } else {
        // first output of the lane's segment, without a per-lane 64-bit division: the block's first output
        // (uniform; exact in double, the quotient is far below 2^53) and a small per-lane quotient (< Bn / q)
        const int q = P.out_stride;
        const int64_t blk_rel = (int64_t)blk * Bn - P.k0L;   // block start relative to output 0
        int64_t jb = 0;
        int64_t o0 = -blk_rel;                                // position of output jb inside the block
        if (blk_rel > 0) {
            jb = (int64_t)floor((double)(blk_rel + q - 1) / (double)q);
            o0 = jb * q - blk_rel;
        }
        const int64_t x = (int64_t)lane * L - o0;             // outputs of this lane: t >= ceil(x / q)
        int t0 = 0;
        if (x > 0) {
            t0 = (int)((float)(x + q - 1) / (float)q);
            if ((int64_t)t0 * q < x) ++t0;                    // (the float quotient is off by at most one)
            if ((int64_t)(t0 - 1) * q >= x) --t0;
        }
        int64_t j = jb + t0;
        int64_t next = o0 + (int64_t)t0 * q - (int64_t)lane * L;  // position inside this segment
        double *y0 = P.y0 + (int64_t)row * P.n_out * 2;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            if (i == next) {
                if (j < P.n_out) { y0[j * 2] = xr[i]; y0[j * 2 + 1] = xi[i]; }
                ++j;
                next += q;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Carry bodies.  The true state entering block b is the D-dimensional recurrence
//     Gf[b] = Mf Gf[b-1] + Ef[b-1],                      Gf[0] = 0 (the start state zi*ext[0]
//                                                         is injected inside block 0 itself)
//     Hb[b-1] = Mb(b) Hb[b] + Eb[b] + U(b) Gf[b],        Hb[nb-1] = zi * f[last]
// (scipy sosfiltfilt `zi * y_0`, _signaltools.py:4823-4824).  Mf = A^(64 L) is a contraction, so
// each carry is evaluated independently per block as the Horner form of its series, cut after
// P.carry_terms terms; the host picks carry_terms so that max|Mf^terms| < 1e-24 (or = nb, in
// which case the series is complete).  One thread per (row, block, component).
// ------------------------------------------------------------------------------------------
template <int D>
TDM_HD void pz_carry_last(const ZpParams &P, int row, int ch, const double *G);   // pz_kernels.hpp

template <int D, class MP>
TDM_HD void matvec_acc(MP M, const double *v, double *out)
{
#pragma unroll
    for (int r = 0; r < D; ++r) {
        double a = out[r];
#pragma unroll
        for (int k = 0; k < D; ++k) a += M[r * D + k] * v[k];
        out[r] = a;
    }
}

template <int K, int NSEC>
TDM_HD void zp_carry_fwd_body(const ZpParams &P, int row, int b, int ch)
{
    constexpr int D = K * NSEC;
    const int nb = P.nb;
    const int64_t base = (int64_t)row * nb * D * 2 + ch;
    const double *Ef = P.Ef + base;
    double *Gf = P.Gf + base;
    double G[D];
#pragma unroll
    for (int k = 0; k < D; ++k) G[k] = 0;
    int first = b - P.carry_terms;
    if (first < 0) first = 0;
    for (int bb = first; bb < b; ++bb) {  // G <- Mf G + Ef[bb]
        double Gn[D];
#pragma unroll
        for (int k = 0; k < D; ++k) Gn[k] = Ef[((int64_t)bb * D + k) * 2];
        matvec_acc<D>(TDM_CPTR(P.Mf), G, Gn);
#pragma unroll
        for (int k = 0; k < D; ++k) G[k] = Gn[k];
    }
#pragma unroll
    for (int k = 0; k < D; ++k) Gf[((int64_t)b * D + k) * 2] = G[k];
    if (b == nb - 1 && P.pform) {
        pz_carry_last<D>(P, row, ch, G);   // parallel form: anticausal bank's start state
    } else if (b == nb - 1) {
        // true forward output at the last extended sample -> start state of the backward pass
        double fl = P.flast[(int64_t)row * 2 + ch];
        const auto c = TDM_CPTR(P.cf_last);
#pragma unroll
        for (int k = 0; k < D; ++k) fl += c[k] * G[k];
        double *Hb = P.Hb + base;
#pragma unroll
        for (int s = 0; s < NSEC; ++s)
#pragma unroll
            for (int k = 0; k < K; ++k) Hb[((int64_t)b * D + s * K + k) * 2] = P.zi[s][k] * fl;
    }
}

// for b < nb-1 (Hb[nb-1] was written by zp_carry_fwd_body)
template <int K, int NSEC>
TDM_HD void zp_carry_bwd_body(const ZpParams &P, int row, int b, int ch)
{
    constexpr int D = K * NSEC;
    const int nb = P.nb;
    if (b >= nb - 1) return;
    const int64_t base = (int64_t)row * nb * D * 2 + ch;
    const double *Eb = P.Eb + base;
    const double *Gf = P.Gf + base;
    double *Hb = P.Hb + base;
    double H[D];
    int far = b + P.carry_terms;  // farthest block whose contribution is kept
    if (far >= nb - 1) {
        far = nb - 1;
#pragma unroll
        for (int k = 0; k < D; ++k) H[k] = Hb[((int64_t)(nb - 1) * D + k) * 2];
    } else {
#pragma unroll
        for (int k = 0; k < D; ++k) H[k] = 0;
    }
    for (int bb = far; bb > b; --bb) {  // H <- Mb(bb) H + Eb[bb] + U(bb) Gf[bb]
        const bool last = (bb == nb - 1);
        double Hn[D], Gb[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            Hn[k] = Eb[((int64_t)bb * D + k) * 2];
            Gb[k] = Gf[((int64_t)bb * D + k) * 2];
        }
        matvec_acc<D>(TDM_CPTR(last ? P.Mb_last : P.Mf), H, Hn);
        if (!P.pform) matvec_acc<D>(TDM_CPTR(last ? P.U_last : P.U_reg), Gb, Hn);   // (the parallel-form banks do not couple)
#pragma unroll
        for (int k = 0; k < D; ++k) H[k] = Hn[k];
    }
#pragma unroll
    for (int k = 0; k < D; ++k) Hb[((int64_t)b * D + k) * 2] = H[k];
}

// ------------------------------------------------------------------------------------------
// Fix-up body: out[j] = y0[j] + T1[m].Gf_b + T2[m].Hb_b, then (optionally) process()'s
// freq_offset NCO at the output rate (processor.py:260-261).  One thread per output sample.
// ------------------------------------------------------------------------------------------
// One output of block b at in-block offset m (row/b uniform over the workgroup, so the carries are
// read with scalar loads; the table rows of consecutive outputs are consecutive in memory).
// table row of block offset m (phase-major: outputs of one decimation phase are consecutive rows)
TDM_HD size_t zp_fixup_row(const ZpParams &P, int b, int m)
{
    const unsigned q = (unsigned)P.out_stride;
    if (q == 1) return (size_t)m;   // (uniform branch; spares the channel filter a runtime division)
    const unsigned R = (b == P.nb - 1) ? P.R_last : P.R_reg;
    return ((unsigned)m % q) * (size_t)R + (unsigned)m / q;
}

// y[j] = y0[j] + T1[r] . Gf[b] + T2[r] . Hb[b] with r the table row of the output (zp_fixup_row),
// split into the per-lane loads and the arithmetic so that a loop can request the next output's
// operands before it finishes the current one
template <int D>
TDM_HD void zp_fixup_load_tables(const ZpParams &P, int b, size_t r, FixOperands<D> &o)
{
    const bool last = (b == P.nb - 1);
    const double *T1 = (last ? P.T1_last : P.T1_reg) + r * D;
    const double *T2 = (last ? P.T2_last : P.T2_reg) + r * D;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        o.t1[k] = T1[k];
        o.t2[k] = T2[k];
    }
}
template <int D>
TDM_HD void zp_fixup_load(const ZpParams &P, int row, int b, size_t r, int64_t j, FixOperands<D> &o)
{
    zp_fixup_load_tables<D>(P, b, r, o);
    const double *y0 = P.y0 + ((int64_t)row * P.n_out + j) * 2;
#if defined(__HIP_DEVICE_COMPILE__)
    // streamed once: keep it from evicting the response tables out of the vector L1
    typedef double f64x2v __attribute__((ext_vector_type(2)));
    const f64x2v yv = __builtin_nontemporal_load((const f64x2v *)y0);
    o.yr = yv.x;
    o.yi = yv.y;
#else
    o.yr = y0[0];
    o.yi = y0[1];
#endif
}
template <int D>
TDM_HD void zp_fixup_apply(const ZpParams &P, int row, int b, const FixOperands<D> &o, double &re, double &im)
{
    const int64_t cb = ((int64_t)row * P.nb + b) * D * 2;
    const auto Gf = TDM_CPTR(P.Gf + cb);
    const auto Hb = TDM_CPTR(P.Hb + cb);
    // two FMA chains per component (forward and backward carries) instead of mul + fma + add per term
    double fr = o.yr, fi = o.yi, br = 0, bi = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        fr = fma(o.t1[k], Gf[k * 2], fr);
        fi = fma(o.t1[k], Gf[k * 2 + 1], fi);
        br = fma(o.t2[k], Hb[k * 2], br);
        bi = fma(o.t2[k], Hb[k * 2 + 1], bi);
    }
    re = fr + br;
    im = fi + bi;
}
template <int D>
TDM_HD void zp_fixup_rowed(const ZpParams &P, int row, int b, size_t r, int64_t j, double &re, double &im)
{
    FixOperands<D> o;
    zp_fixup_load<D>(P, row, b, r, j, o);
    zp_fixup_apply<D>(P, row, b, o, re, im);
}

template <int D>
TDM_HD void zp_fixup_at(const ZpParams &P, int row, int b, int m, int64_t j, double &re, double &im)
{
    zp_fixup_rowed<D>(P, row, b, zp_fixup_row(P, b, m), j, re, im);
}

// Fix-up body: one workgroup per (row, block); thread t handles the block's outputs t, t+nt, ...
template <int D, int L>
TDM_HD void zp_fixup_body(const ZpParams &P, int row, int b, int tid, int nt, double *out /* row base */,
                          const double *freq_offset /* per row or null */, double fs_out)
{
    const int Bn = kWave * P.L;
    const int q = P.out_stride;
    const int64_t base = (int64_t)b * Bn - P.k0L;  // pos - k0L of offset m == 0
    // first offset m0 >= 0 of this block that is an output: (base + m0) % q == 0 and base + m0 >= 0
    int64_t m0 = base >= 0 ? (q - base % q) % q : -base;
    const int len = (b == P.nb - 1) ? P.len_last : Bn;
    const double f = freq_offset ? freq_offset[row] : 0.0;
    for (int64_t m = m0 + (int64_t)tid * q; m < len; m += (int64_t)nt * q) {
        const int64_t j = (base + m) / q;
        if (j >= P.n_out) break;
        double re, im;
        zp_fixup_at<D>(P, row, b, (int)m, j, re, im);
        if (f != 0.0) nco_rotate(re, im, j, f, fs_out);
        out[j * 2] = re;
        out[j * 2 + 1] = im;
    }
}

// ------------------------------------------------------------------------------------------
// Finish body: timing-phase pick + symbol gather (extract_symbols) and the differential slicer
// (demodulate_dqpsk).  One workgroup per row.
//   Comm: tid(), nthreads() (a multiple of kMaxSps), sync(), lds(i) -> double& (nthreads() doubles of
//   workgroup scratch), reduce_sum/max/min(double) -> value on every thread.
// ------------------------------------------------------------------------------------------
struct FinishArgs {
    const double *z;      // [rows][n] c128 input at rate fs
    int64_t n;
    int64_t row_stride;   // samples between rows
    int32_t sps;          // int(fs / symbol_rate)
    int32_t do_extract;   // 0: the input already is the symbol stream
    int32_t do_demod;     // 0: stop after the gather
    int32_t max_soft;     // capacity per row of soft / hard
    double *soft;         // [rows][max_soft] c128
    uint8_t *hard;        // [rows][max_soft]
    int32_t *n_soft;      // [rows]
    int32_t *best_phase;  // [rows] or null
    double *min_margin;   // [rows] or null
    // optional per-block partial phase powers made by power_fixup_body (deterministic order):
    const double *partials;  // [rows][n_pblk][kMaxSps] or null -> powers are computed here
    int32_t n_pblk;
    // use_fix != 0: z is not materialised; sample j is the channel filter's block-local output plus its
    // carry responses, evaluated where it is gathered (zp_fixup_at<4> on `fix`, block length kFixBn)
    int32_t use_fix;
    ZpParams fix;
    // zt != null: the filter output is final and stored phase-major by the low-rate kernel (lp2_kernels.hpp):
    // sample p + sps*k of a row at zt[(row*sps + p)*zt_k + k]
    const double *zt;
    int64_t zt_k;
    // smear != 0: a zero-phase filter ran in front of this stage (scipy's decimate / filtfilt, processor.py:254, :79).  In the
    // reference such a filter carries ONE non-finite sample over the whole chunk -- forward pass to the end, backward pass,
    // started from the forward pass's last value, back to the start -- so every soft symbol is NaN, no timing phase beats
    // max_power = -1 (:196-210 -> phase 0) and every slicer comparison is false (:152-161 -> symbol 3).  The device's
    // filters are evaluated in blocks whose carries are cut below 1e-30, so a NaN stays inside its block and chunk; but it
    // does reach that chunk's phase powers, every one of them, and a non-finite phase power here is therefore the reference's
    // all-NaN chunk.  (smear == 0 -- the stand-alone methods, and the <= 15 samples no filter takes: nothing is carried
    // anywhere and the comparisons below behave as numpy's do.)
    int32_t smear;
};

constexpr int kMaxSps = 32;       // phases a partial-power record holds
constexpr int kPowThreads = 256;  // threads (= samples) per partial-power block
constexpr int kPowSub = 8;        // partial-power blocks per workgroup
#define TDM_LLPF 8
constexpr int kFixBn = kWave * TDM_LLPF;  // block length of the channel filter (ref_plan.hpp kLLpf)

TDM_HD uint8_t dqpsk_decide(double cr, double ci, double pr, double pi_, double &margin)
{
    // diff = sample * conj(prev), numpy scalar complex product: two roundings per component
    const double br = pr, bi = -pi_;
    const double dr = sub_rn(mul_rn(cr, br), mul_rn(ci, bi));
    const double di = add_rn(mul_rn(cr, bi), mul_rn(ci, br));
    const double ph = atan2(di, dr);
    const double t0 = -5 * M_PI / 8, t1 = -3 * M_PI / 8, t2 = 3 * M_PI / 8, t3 = 5 * M_PI / 8;
    uint8_t sym;
    if (ph < t0) sym = 3;
    else if (ph < t1) sym = 2;
    else if (ph < t2) sym = 0;
    else if (ph < t3) sym = 1;
    else sym = 3;
    double m = fabs(ph - t0);
    m = fmin(m, fabs(ph - t1));
    m = fmin(m, fabs(ph - t2));
    m = fmin(m, fabs(ph - t3));
    margin = m;
    return sym;
}

template <class Comm>
TDM_HD void finish_body(const FinishArgs &A, Comm &cm, int row)
{
    const int tid = cm.tid(), nt = cm.nthreads();
    const double *z = A.z + (int64_t)row * A.row_stride * 2;
    double *soft = A.soft + (int64_t)row * A.max_soft * 2;
    uint8_t *hard = A.hard ? A.hard + (int64_t)row * A.max_soft : nullptr;
    const int64_t n = A.n;
    int64_t best = 0;
    int64_t ns = n;
    const int64_t sps = A.sps;
    bool nonfinite = false;   // (FinishArgs::smear) the reference's zero-phase filters made this chunk all-NaN
    if (A.do_extract && n > 0 && sps > 1) {
        const int64_t step = sps / 8 > 1 ? sps / 8 : 1;
        if (A.partials) {
            // phase p = tid % kMaxSps, group g = tid / kMaxSps: each thread sums every (nt/kMaxSps)-th
            // block record of its phase, then the groups are added in group order (fixed order ->
            // reproducible result)
            const int p = tid % kMaxSps, g = tid / kMaxSps, ng = nt / kMaxSps;
            {
                double acc = 0;
                if (p < sps) {
                    const double *pp = A.partials + (int64_t)row * A.n_pblk * kMaxSps + p;
                    for (int b = g; b < A.n_pblk; b += ng) acc += pp[(int64_t)b * kMaxSps];
                }
                cm.lds(tid) = acc;
            }
            cm.sync();
            double power = -1.0;
            if (tid < sps && tid % step == 0) {
                const int64_t np_ = (n - tid) / sps;
                if (n - tid > 0 && np_ > 0) {
                    double acc = 0;
                    for (int gg = 0; gg < ng; ++gg) acc += cm.lds(gg * kMaxSps + tid);
                    power = acc / (double)np_;
                }
            }
            // "first strictly greater wins" == the lowest phase among those with the maximum power
            const double mxp = cm.reduce_max(power);   // (fmax: a NaN power never wins, as `power > max_power` never does)
            // a non-finite power of a tried phase rides the same reduction as candidate -1
            const double cand = (A.smear && !(fabs(power) <= 1.7976931348623157e308)) ? -1.0
                                : ((power >= 0 && power == mxp) ? (double)tid : 1e9);
            const double first = cm.reduce_min(cand);
            nonfinite = first < 0;
            best = (first < 1e8 && !nonfinite) ? (int64_t)first : 0;
        } else {
            double maxp = -1.0;
            for (int64_t ph = 0; ph < sps; ph += step) {
                const int64_t np_ = (n - ph) / sps;
                if (n - ph <= 0 || np_ <= 0) continue;
                double acc = 0;
                for (int64_t k = tid; k < np_; k += nt) {
                    const double *s = z + (ph + k * sps) * 2;
                    const double m = hypot(s[0], s[1]);
                    acc += m * m;
                }
                const double power = cm.reduce_sum(acc) / (double)np_;
                if (power > maxp) { maxp = power; best = ph; }  // identical on every thread
                if (A.smear && !(fabs(power) <= 1.7976931348623157e308)) nonfinite = true;
            }
            if (nonfinite) best = 0;
        }
        ns = (n - best) / sps;
    }
    if (ns > A.max_soft) ns = A.max_soft;
    if (ns < 0) ns = 0;
    const int64_t stride = (A.do_extract && sps > 1) ? sps : 1;
    double mx = 0;
    for (int64_t k = tid; k < ns; k += nt) {
        const int64_t j = best + k * stride;
        double re, im;
        if (nonfinite) {
            re = im = NAN;
        } else if (A.zt) {
            const double *sp = A.zt + (((int64_t)row * sps + best) * A.zt_k + k) * 2;
            re = sp[0];
            im = sp[1];
        } else if (A.use_fix) {
            const int64_t pos = j + A.fix.k0L;
            const int b = (int)(pos / kFixBn);
            zp_fixup_at<4>(A.fix, row, b, (int)(pos - (int64_t)b * kFixBn), j, re, im);
        } else {
            re = z[j * 2];
            im = z[j * 2 + 1];
        }
        soft[k * 2] = re;
        soft[k * 2 + 1] = im;
        const double h = hypot(re, im);
        mx = (mx != mx) ? mx : ((h != h || h > mx) ? h : mx);   // np.max: a NaN wins
    }
    if (tid == 0) {
        A.n_soft[row] = (int32_t)ns;
        if (A.best_phase) A.best_phase[row] = (int32_t)best;
    }
    if (!A.do_demod) return;
    mx = cm.reduce_max_nan(mx);   // (its barriers also make the soft symbols of other threads visible)
    double margin = INFINITY;
    if (ns >= 2) {
        // samples / max_power == samples * fl(1/max).  `if max_power > 0` (processor.py:126) is false for a NaN maximum: no
        // normalisation; an infinite one scales every finite sample to a signed zero, as the reference's division does
        const double scl = mx > 0 ? 1.0 / mx : 1.0;
        for (int64_t k = 1 + tid; k < ns; k += nt) {
            const double *c = soft + k * 2;
            const double *p = soft + (k - 1) * 2;
            double mg;
            hard[k - 1] = dqpsk_decide(mul_rn(c[0], scl), mul_rn(c[1], scl), mul_rn(p[0], scl),
                                       mul_rn(p[1], scl), mg);
            margin = fmin(margin, mg);
        }
    }
    margin = cm.reduce_min(margin);
    if (tid == 0 && A.min_margin) A.min_margin[row] = margin;
}

// ------------------------------------------------------------------------------------------
// Power/fix-up body: z[j] = fix-up value, stored, plus this block's partial sums of |z|^2 per
// timing phase (extract_symbols' mean powers, processor.py:196-206, summed in a fixed order so the
// result is reproducible run to run).  One workgroup of kPowThreads threads per kPowThreads samples.
//   Comm: tid(), sync(), lds(i) -> double& (workgroup-shared scratch of kPowThreads doubles)
// ------------------------------------------------------------------------------------------
template <int D, int L, class Comm>
TDM_HD void power_fixup_body(const ZpParams &P, Comm &cm, int row, int wg, double *z_row /* or null */, int64_t n,
                             int sps, double *partials_row /* [n_pblk][kMaxSps] */, int n_pblk)
{
    constexpr int Bn = kWave * L;
    constexpr int kSub = Bn / kPowThreads;  // partial-power blocks per filter block (out_stride == 1 here)
    static_assert(Bn % kPowThreads == 0, "block must be a whole number of power blocks");
    const int t = cm.tid();
    // |z|^2 of kPowSub x kPowThreads samples: independent chains per thread
#pragma unroll
    for (int u = 0; u < kPowSub; ++u) {
        const int blk = wg * kPowSub + u;
        double sq = 0;
        if (blk < n_pblk) {
            const int b = blk / kSub;
            const int m = (blk % kSub) * kPowThreads + t;
            const int64_t j = (int64_t)b * Bn + (blk % kSub) * kPowThreads - P.k0L + t;  // may be negative
            if (j >= 0 && j < n) {
                double re, im;
                zp_fixup_at<D>(P, row, b, m, j, re, im);
                if (z_row) {
                    z_row[j * 2] = re;
                    z_row[j * 2 + 1] = im;
                }
                const double mg = hypot(re, im);
                sq = mg * mg;
            }
        }
        cm.lds(u * kPowThreads + t) = sq;
    }
    cm.sync();
    // one thread per (power block, phase): the block's samples of that phase, summed in index order
    const int u = t / kMaxSps, ph = t % kMaxSps;
    const int blk = wg * kPowSub + u;
    if (u < kPowSub && blk < n_pblk) {
        double acc = 0;
        if (ph < sps) {
            const int b = blk / kSub;
            const int64_t j0 = (int64_t)b * Bn + (blk % kSub) * kPowThreads - P.k0L;
            const int64_t np_ = (n - ph) / sps;            // samples phase ph owns: j = ph + k*sps, k < np_
            const int64_t lim = ph + np_ * (int64_t)sps;   // first j NOT owned
            const int64_t first = (((int64_t)ph - j0) % sps + sps) % sps;  // (j0 + first) % sps == ph
            for (int64_t i = first; i < kPowThreads; i += sps)
                if (j0 + i >= 0 && j0 + i < lim) acc += cm.lds(u * kPowThreads + (int)i);
        }
        partials_row[(int64_t)blk * kMaxSps + ph] = acc;
    }
}

// frequency_shift as a stand-alone elementwise op (public method, processor.py:85-100)
TDM_HD void shift_body(const double *x, double *y, int64_t k, double f, double fs)
{
    double re = x[k * 2], im = x[k * 2 + 1];
    nco_rotate(re, im, k, f, fs);  // the reference multiplies even when f == 0 (exp(0) == 1)
    y[k * 2] = re;
    y[k * 2 + 1] = im;
}

// ------------------------------------------------------------------------------------------
// resample (processor.py:35-49 -> scipy.signal.resample, FFT method): X = fft(x); keep the
// min(num, N) lowest-|f| bins (Nyquist bin folded / split as scipy does); y = ifft(Y) * num/N.
// Evaluated as two direct DFTs over the kept bins only; every twiddle is an exact sincospi of the
// integer phase index (k*n mod N), so the error does not grow with N.  O(N * kept) work: this
// method is not on the process() path (no caller in the reference uses it) and is built for
// completeness of the class interface, not for speed.
//   Comm: tid(), nthreads(), reduce_sum(double)
// ------------------------------------------------------------------------------------------

TDM_HD void sincospi_d(double t, double *s, double *c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    sincospi(t, s, c);
#else
    // host (test harness): exact quadrant reduction, then libm on |r| <= 1/4
    const double k = rint(2.0 * t);
    const double r = t - 0.5 * k;
    double sr, cr;
    sincos(M_PI * r, &sr, &cr);
    switch (((long long)k % 4 + 4) % 4) {
    case 0: *s = sr; *c = cr; break;
    case 1: *s = cr; *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
    }
#endif
}

TDM_HD void unit_root(int64_t m, int64_t n, double sign, double &c, double &s)
{
    // exp(sign * 2*pi*i * m / n), 0 <= m < n
    sincospi_d(2.0 * (double)m / (double)n, &s, &c);
    s *= sign;
}

// out[o] = scale * sum_i w[i] * in[src[i]] * exp(sign*2*pi*i * f[i]*o' / n) with o' = (o_list ? o_list[o] : o)
// one workgroup per output o; threads stride over the terms i.
template <class Comm>
TDM_HD void dft_terms_body(Comm &cm, int64_t o, const int64_t *o_list, const double *in, int64_t n_terms,
                           const int64_t *src, const int64_t *freq, const double *weight, int64_t n, double sign,
                           double scale, double *out)
{
    const int64_t oo = o_list ? o_list[o] : o;
    const int tid = cm.tid(), nt = cm.nthreads();
    double accr = 0, acci = 0;
    for (int64_t i = tid; i < n_terms; i += nt) {
        const int64_t f = freq ? freq[i] : i;
        const int64_t m = (int64_t)(((uint64_t)f * (uint64_t)oo) % (uint64_t)n);  // f, oo < n < 2^31
        double c, s;
        unit_root(m, n, sign, c, s);
        const int64_t k = src ? src[i] : i;
        const double w = weight ? weight[i] : 1.0;
        const double xr = in[2 * k] * w, xi = in[2 * k + 1] * w;
        accr += xr * c - xi * s;
        acci += xr * s + xi * c;
    }
    accr = cm.reduce_sum(accr);
    acci = cm.reduce_sum(acci);
    if (tid == 0) {
        out[2 * o] = accr * scale;
        out[2 * o + 1] = acci * scale;
    }
}

}  // namespace tdm
