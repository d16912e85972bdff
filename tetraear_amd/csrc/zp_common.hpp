// Shared definitions for the zero-phase IIR block engine (host table builder + device kernels).
//
// The engine evaluates scipy's sosfiltfilt / filtfilt (as called by the reference at
// processor.py:254 via signal.decimate, and processor.py:79) WITHOUT running one long sequential
// recurrence: the padded, odd-extended signal is cut into blocks of 64 segments x L samples; a
// 64-lane wavefront filters one block with every lane starting from zero state, and linearity is
// used twice to restore the exact result:
//   in-block : per section, a wavefront prefix scan over the lanes' end states (K x K transition
//              matrices A^(L*2^j)) gives each lane its true start state; the zero-input response
//              of that state (the section run on zero input) is added to the lane's samples;
//   x-block  : each block exports its end states (forward Ef, backward Eb); a tiny carry kernel
//              runs the D-dimensional recurrences across blocks (tables Mf, Mb, U) and the
//              consumer adds the two carry responses (tables T1, T2) to the block-local outputs.
// In-block nothing is truncated.  Across blocks the carry recurrence is evaluated as a series in
// Mf = A^(64 L) cut where max|Mf^t| < 1e-30 (or complete, t = nb), far below fp64 rounding, so the
// result equals the sequential recurrence to rounding for any filter memory and any length.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TDM_HD __host__ __device__ __forceinline__
#else
#define TDM_HD inline
#endif

// Read-only tables are read through the constant address space on the device so that uniform
// reads become scalar loads (s_load) instead of per-lane vector loads.
#if defined(__HIP_DEVICE_COMPILE__)
#define TDM_CPTR(p) ((const __attribute__((address_space(4))) double *)(p))
#else
#define TDM_CPTR(p) (p)
#endif


namespace tdm {

constexpr int kWave = 64;        // lanes per block (one wavefront)
constexpr int kMaxSec = 4;       // sections per cascade
constexpr int kMaxOrd = 4;       // order of one section
constexpr int kMaxD = 8;         // total state dimension
constexpr int kScanSteps = 6;    // log2(kWave)

// One zero-phase stage over a batch of independent rows (carriers).
struct ZpParams {
    // ---- filter: cascade of nsec DF2T sections of order K (a[.][0] == 1)
    int32_t nsec, K;
    double b[kMaxSec][kMaxOrd + 1];
    double a[kMaxSec][kMaxOrd + 1];
    double zi[kMaxSec][kMaxOrd];  // steady-state start per unit input (sosfilt_zi / lfilter_zi)
    double in_gain;               // applied once to every input sample (see ZpFilterDesc)
    // ---- geometry (padded extended domain: index = P0 + ext index)
    int64_t n;         // signal samples per row
    int32_t edge;      // odd-extension length each side (27 cheby sos, 15 butter tf)
    int32_t L;         // samples per lane
    int32_t P0;        // lead pad: (P0 + edge) % L == 0, so signal sample 0 sits on a lane boundary
    int32_t k0L;       // P0 + edge
    int64_t Ne;        // P0 + n + 2*edge
    int32_t nb;        // blocks per row = ceil(Ne / (64 L))
    int32_t len_last;  // valid length of the last block
    int64_t n_out;     // outputs per row
    int32_t out_stride;  // q for the decimator, 1 otherwise
    int32_t carry_terms; // series length of the cross-block carries (see zp_kernels.hpp)
    // ---- tables (pointers valid where the kernels run)
    const double *Mpow;     // [nsec][6][K*K]   A_s^(L*2^j), row-major
    const double *zirh;     // [nsec][L][K]     zero-input response of section s from unit states
    // carry-response tables, one row of D doubles per in-block offset m, stored PHASE-MAJOR:
    // row(m) = (m % out_stride) * R + m / out_stride, so that consecutive outputs (m advancing by
    // out_stride) read consecutive rows.  "reg" = full 64L-sample block, "last" = the last block.
    const double *cf_last;  // [D]      forward zero-input response at offset len_last-1
    const double *T1_reg;   // [..][D]  fwd carry-in -> block-local fwd->bwd output at offset m
    const double *T2_reg;   // [..][D]  bwd carry-in -> output at offset m (backward zero-input resp.)
    const double *T1_last;
    const double *T2_last;
    int32_t R_reg, R_last;  // rows per phase: ceil(len / out_stride)
    const double *Mf;       // [D][D] A^(64L)
    const double *Mb_last;  // [D][D] A^(len_last)
    const double *U_reg;    // [D][D] fwd carry-in -> bwd end state of the block
    const double *U_last;   // [D][D]
    // ---- per-row work buffers
    double *y0;     // [rows][n_out] c128, block-local outputs
    double *Ef;     // [rows][nb][D] c128, block-local forward end states
    double *Eb;     // [rows][nb][D] c128, block-local backward end states
    double *flast;  // [rows] c128, block-local forward output at the last extended sample
    double *Gf;     // [rows][nb][D] c128, resolved forward carry into each block
    double *Hb;     // [rows][nb][D] c128, resolved backward carry into each block
    // ---- parallel-form stage (pz_kernels.hpp / pz_tables.hpp); pform == 0 for the cascade engine
    int32_t pform;
    const double *pz;   // constant block, layout PzLayout
    double *Elast;      // [rows][D] c128, causal lane state exported by the last block (see pz_tables.hpp)
};

// One DF2T section step, the operation order of scipy's sosfilt (K == 2,
// scipy/signal/_sosfilt.pyx) and lfilter (K == 4, scipy/signal/_lfilter.c.in).
template <int K, typename T>
TDM_HD T df2t_step(const T *b, const T *a, T x, T *z)
{
    if (K == 2) {
        T y = b[0] * x + z[0];
        z[0] = (b[1] * x - a[1] * y) + z[1];
        z[1] = b[2] * x - a[2] * y;
        return y;
    } else {
        T y = z[0] + b[0] * x;
#pragma unroll
        for (int k = 1; k < K; ++k) z[k - 1] = (z[k] + b[k] * x) - a[k] * y;
        z[K - 1] = b[K] * x - a[K] * y;
        return y;
    }
}

// Biquad with numerator exactly [1, 2, 1] (every section of a Butterworth / Chebyshev-I lowpass
// once its gain is pulled out): 4 fp64 operations per real sample instead of 6.
template <typename T>
TDM_HD T lp121_step(const T *a, T x, T *z)
{
    const T y = x + z[0];
    z[0] = (2 * x + z[1]) - a[1] * y;
    z[1] = x - a[2] * y;
    return y;
}

// Zero-input step of the same section: returns the output for x == 0 and advances the state.
template <int K, typename T>
TDM_HD T zir_step(const T *a, T *z)
{
    const T y = z[0];
#pragma unroll
    for (int k = 1; k < K; ++k) z[k - 1] = z[k] - a[k] * y;
    z[K - 1] = -(a[K] * y);
    return y;
}

}  // namespace tdm
