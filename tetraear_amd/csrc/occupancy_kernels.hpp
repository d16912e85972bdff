// Occupancy gate of the wideband chain (SURVEY 8(f) N2 in its many-carrier form): which channel rows of a polyphase
// channeliser's output carry a signal, decided on the device, so that the receiver is launched over those rows only.
//
// The rule is the reference's gate (tetraear/ui/modern.py:1921-2003: CaptureThread.run decides with it WHETHER process()
// runs), applied per channel row at the channel rate:
//     Hann-windowed FFT of the row's first 256 samples, power = 20 log10(|X| / N + 1e-20)           (:1926-1934)
//     signal_power = mean, peak_power = max of the power over the bins within 25 kHz around the centre (:1948-1957)
//     strong = snr > 15 dB  and  peak_power > -70 dBFS  and  peak_power - signal_power > 3 dB        (:1992-1995)
// with ONE change: the reference takes its noise floor from the bins outside the centre channel (:1969-1986), which in a
// bank of neighbouring carriers are other carriers; here the floor is the MEDIAN of signal_power over the stream's
// channels (fewer than half of them occupied).  Definition: oracle/pfb_np.py occupancy().  No reference counterpart for
// the many-carrier form ("parity unpinned"); the per-row formulas are the reference's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "small_dft.hpp"

namespace tdm {

constexpr int kOccFft = 256;          // samples of a row the gate looks at (3.2 ms at 80 kS/s; the reference: 2048 of 2.4 MS/s = 0.85 ms)
constexpr int kOccRowsPerWg = 16;     // 16 threads per row
constexpr int kOccMaxM = 1024;        // channels per stream the list kernel ranks in LDS

struct OccArgs {
    const float2 *chan;      // [rows][pitch] cf32 channel rows
    int64_t pitch;
    int32_t rows;
    int32_t bin_lo, bin_hi;  // fftshifted bins [bin_lo, bin_hi) = the 25 kHz around the centre
    float2 *stats;           // [rows] (signal_power, peak_power) in dBFS
    int32_t *n_rows;         // the list kernel's counter, put to zero here (the launch before it on the stream)
};

// power spectrum statistics of every row: 256-point FFT as 16 x 16 (register-resident 16-point transforms, one exchange
// through LDS), sixteen threads per row
__global__ __launch_bounds__(256) void k_occ_spectrum(const OccArgs A)
{
    __shared__ cf32v ex[kOccRowsPerWg][16][17];
    __shared__ cf32v tw[kOccFft];
    __shared__ float win[kOccFft];
    const int tid = threadIdx.x, rl = tid >> 4, t = tid & 15;
    if (blockIdx.x == 0 && tid == 0) *A.n_rows = 0;
    {
        float s, c;
        __sincosf(6.283185307179586f * (float)tid / (float)kOccFft, &s, &c);
        tw[tid] = cv(c, s);
        win[tid] = 0.5f - 0.5f * __cosf(6.283185307179586f * (float)tid / (float)(kOccFft - 1));   // np.hanning(N)
    }
    __syncthreads();
    const int row = (int)blockIdx.x * kOccRowsPerWg + rl;
    const bool valid = row < A.rows;
    const float2 *src = A.chan + (int64_t)(valid ? row : A.rows - 1) * A.pitch;
    cf32v x[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const int n = 16 * n1 + t;
        const float2 v = src[n];
        const float w = win[n];
        x[n1] = cv(v.x * w, -v.y * w);      // (conjugated: SmallDft sums with exp(+i...), numpy's fft with exp(-i...))
    }
    SmallDft<16>::run(x);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) ex[rl][k1][t] = cmulv(x[k1], tw[(k1 * t) & (kOccFft - 1)]);
    __syncthreads();
    cf32v a[16];
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) a[n2] = ex[rl][t][n2];
    SmallDft<16>::run(a);                   // a[k2] = X[t + 16 k2]
    float sum = 0.f, mx = -1e30f;
    int cnt = 0;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
        const int k = t + 16 * k2, i = (k + kOccFft / 2) & (kOccFft - 1);   // fftshift
        const float mag = sqrtf(a[k2].x * a[k2].x + a[k2].y * a[k2].y) * (1.0f / kOccFft) + 1e-20f;
        const float db = 20.0f * log10f(mag);
        if (i >= A.bin_lo && i < A.bin_hi) {
            sum += db;
            mx = fmaxf(mx, db);
            ++cnt;
        }
    }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        sum += __shfl_xor(sum, d, 64);
        mx = fmaxf(mx, __shfl_xor(mx, d, 64));
        cnt += __shfl_xor(cnt, d, 64);
    }
    if (valid && t == 0) A.stats[row] = make_float2(sum / (float)cnt, mx);
}

struct OccListArgs {
    const float2 *stats;     // [streams][M]
    int32_t M;
    float snr_db, min_dbfs, peak_db;
    uint8_t *flags;          // [rows] 1 = occupied
    int32_t *row_list;       // [rows] the occupied rows, in no particular order
    int32_t *n_rows;         // [1] their number (zero when the kernel starts)
    int32_t *n_soft;         // [rows] or null: set to 0 for rows that are NOT occupied (the receiver skips them)
};

// one workgroup per stream: noise floor = median of the channels' signal_power (bitonic sort of the M values, padded with
// +inf to 1024, in LDS: with one workgroup per stream every wavefront has its SIMD to itself, so it is the instruction
// COUNT that matters -- ranking every channel against every other took 30 us, the sort 3), the rule, the row list
__global__ __launch_bounds__(512) void k_occ_list(const OccListArgs A)
{
    __shared__ float sorted[kOccMaxM];
    const int M = A.M, s = blockIdx.x, tid = threadIdx.x;
    const float2 *st = A.stats + (int64_t)s * M;
    for (int k = tid; k < kOccMaxM; k += 512) sorted[k] = k < M ? st[k].x : __builtin_inff();
    __syncthreads();
    for (int size = 2; size <= kOccMaxM; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            // thread t handles the pair (i, i + stride), i = the t-th index with bit `stride` clear
            const int i = 2 * tid - (tid & (stride - 1));
            const bool up = (i & size) == 0;
            const float a = sorted[i], b = sorted[i + stride];
            if ((a > b) == up) {
                sorted[i] = b;
                sorted[i + stride] = a;
            }
            __syncthreads();
        }
    }
    const float floor_db = (M & 1) ? sorted[M / 2] : 0.5f * (sorted[M / 2 - 1] + sorted[M / 2]);
    for (int k = tid; k < M; k += 512) {
        const float2 v = st[k];
        const bool occ = (v.x - floor_db > A.snr_db) && (v.y > A.min_dbfs) && (v.y - v.x > A.peak_db);
        const int row = s * M + k;
        A.flags[row] = occ ? 1 : 0;
        if (occ) A.row_list[atomicAdd(A.n_rows, 1)] = row;
        else if (A.n_soft) A.n_soft[row] = 0;
    }
}

}  // namespace tdm
