// TETRA mode (north-star receiver): per-carrier pi/4-DQPSK demodulation of channelised baseband.
//
// There is no reference implementation of this mode: the algorithm is defined by
// oracle/tetra_np.py (fp64 numpy) and restated here in fp32 for gfx950:
//   k_tetra_rrc : root-raised-cosine matched filter, LDS-tiled sliding window, one pass
//                 HBM -> LDS -> registers -> LDS -> HBM (8 B in + 8 B out per sample, HBM-bound)
//   k_tetra_sym : feed-forward square-law timing estimate (wavefront reductions + prefix sums),
//                 cubic Farrow interpolation at the symbol instants, 4th-power carrier-offset
//                 estimate, differential quadrant decision.  One workgroup per carrier chunk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tdm {

constexpr int kRrcMaxTaps = 96;
constexpr int kRrcThreads = 256;
constexpr int kRrcPerThread = 8;                          // consecutive outputs per thread
constexpr int kRrcTile = kRrcThreads * kRrcPerThread;     // 2048 samples per workgroup
constexpr int kTimingBlock = 256;                         // samples per timing sub-block (TB)
constexpr int kTimingHalfWin = 2;                         // sub-blocks averaged each side (TW)
constexpr int kMaxTimingBlocks = 512;
constexpr int kSymThreads = 256;

struct TetraParams {
    int32_t n;          // samples per carrier chunk
    int32_t ntaps;      // odd
    int32_t max_soft;   // capacity of per-carrier symbol outputs
    int32_t ystride;    // row stride of the matched-filter output (n rounded up to even: 16-byte rows)
    double sps;         // samples per symbol (sample_rate / 18000)
    double inv_sps;
    float step_c, step_s;  // exp(-2 pi i / sps): symbol-clock phasor advance per sample
    float taps[kRrcMaxTaps];
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS index of tile sample s: one pad slot per 8 samples so that a thread's 8-sample-strided
// window reads (ds_read_b64, lane stride 9 slots = 18 dwords) hit 32 distinct bank pairs.
__device__ __forceinline__ int rrc_slot(int s) { return s + (s >> 3); }

template <int NT>
__global__ __launch_bounds__(kRrcThreads) void k_tetra_rrc(const float2 *__restrict__ x, int64_t in_stride,
                                                            float2 *__restrict__ y, float2 *__restrict__ tstat,
                                                            const TetraParams P)
{
    static_assert(kTimingBlock == 32 * kRrcPerThread, "one timing sub-block = 32 threads x 8 outputs");
    constexpr int HALO = NT - 1;
    constexpr int NS = kRrcTile + HALO;  // samples staged
    __shared__ float2 lds[NS + NS / 8 + 2];
    const int row = blockIdx.y;
    const int n = P.n;
    const int64_t base = (int64_t)blockIdx.x * kRrcTile;          // first output of the tile
    const float2 *xr = x + (int64_t)row * in_stride;   // rows of the channeliser may carry a pitch
    float2 *yr = y + (int64_t)row * P.ystride;
    const int t = threadIdx.x;
    // stage inputs base - HALO/2 .. base + tile + HALO/2 (zero outside the chunk), coalesced;
    // 16 bytes per lane (two samples) when the tile start is 16-byte aligned in the row
    if (((HALO / 2) & 1) == 0 && (in_stride & 1) == 0) {
        const f32x4 *x4 = (const f32x4 *)xr;
        for (int s2 = t; s2 < NS / 2; s2 += kRrcThreads) {
            const int s = 2 * s2;
            const int64_t g = base + s - HALO / 2;  // even
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (g >= 0 && g + 1 < n) {
                v = __builtin_nontemporal_load(x4 + (g >> 1));
            } else if (g >= 0 && g < n) {   // last sample of an odd-length chunk
                const float2 q = xr[g];
                v.x = q.x;
                v.y = q.y;
            }
            lds[rrc_slot(s)] = make_float2(v.x, v.y);
            lds[rrc_slot(s + 1)] = make_float2(v.z, v.w);
        }
        if ((NS & 1) && t == 0) {
            const int s = NS - 1;
            const int64_t g = base + s - HALO / 2;
            lds[rrc_slot(s)] = (g >= 0 && g < n) ? xr[g] : make_float2(0.f, 0.f);
        }
    } else {
        for (int s = t; s < NS; s += kRrcThreads) {
            const int64_t g = base + s - HALO / 2;
            float2 v = make_float2(0.f, 0.f);
            if (g >= 0 && g < n) v = xr[g];
            lds[rrc_slot(s)] = v;
        }
    }
    __syncthreads();
    // sliding window in registers: outputs base + 8t + v need staged samples 8t + v .. 8t + v + NT-1
    f32x2 w[kRrcPerThread + HALO];
#pragma unroll
    for (int j = 0; j < kRrcPerThread + HALO; ++j) {
        const float2 q = lds[rrc_slot(kRrcPerThread * t + j)];
        w[j] = f32x2{q.x, q.y};
    }
    f32x2 acc[kRrcPerThread];
#pragma unroll
    for (int v = 0; v < kRrcPerThread; ++v) acc[v] = f32x2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const f32x2 hh = {P.taps[k], P.taps[k]};
#pragma unroll
        for (int v = 0; v < kRrcPerThread; ++v) acc[v] = __builtin_elementwise_fma(hh, w[v + k], acc[v]);  // v_pk_fma_f32
    }
    // square-law timing statistic of this tile's 8 sub-blocks while the outputs are in registers:
    // C_b = sum |y[g]|^2 exp(-2 pi i g / sps)   (Oerder-Meyr); 32 threads x 8 samples per sub-block
    {
        const int64_t g0 = base + kRrcPerThread * t;
        const double ph = (double)g0 * P.inv_sps;
        const float fr = (float)(ph - floor(ph));
        float ps, pc;
        sincospif(-2.f * fr, &ps, &pc);
        float ar = 0.f, ai = 0.f;
#pragma unroll
        for (int v = 0; v < kRrcPerThread; ++v) {
            if (g0 + v < n) {
                const float p = acc[v].x * acc[v].x + acc[v].y * acc[v].y;
                ar = fmaf(p, pc, ar);
                ai = fmaf(p, ps, ai);
            }
            const float nc = pc * P.step_c - ps * P.step_s, ns = pc * P.step_s + ps * P.step_c;
            pc = nc;
            ps = ns;
        }
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) {
            ar += __shfl_xor(ar, d, 64);
            ai += __shfl_xor(ai, d, 64);
        }
        const int nb = (n + kTimingBlock - 1) / kTimingBlock;
        const int b = (int)(base / kTimingBlock) + (t >> 5);
        if ((t & 31) == 0 && b < nb) tstat[(int64_t)row * nb + b] = make_float2(ar, ai);
    }
    __syncthreads();
    // transpose through LDS so that the stores are coalesced
#pragma unroll
    for (int v = 0; v < kRrcPerThread; ++v) lds[rrc_slot(kRrcPerThread * t + v)] = make_float2(acc[v].x, acc[v].y);
    __syncthreads();
    f32x4 *y4 = (f32x4 *)yr;
    for (int s2 = t; s2 < kRrcTile / 2; s2 += kRrcThreads) {
        const int s = 2 * s2;
        const int64_t g = base + s;
        if (g + 1 < n) {
            const float2 a = lds[rrc_slot(s)], b = lds[rrc_slot(s + 1)];
            const f32x4 o = {a.x, a.y, b.x, b.y};
            __builtin_nontemporal_store(o, y4 + (g >> 1));
        } else if (g < n) {
            yr[g] = lds[rrc_slot(s)];
        }
    }
}

// ---- workgroup helpers -------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ float block_sum(float v, float *sm)
{
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sm[i];
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_min(float v, float *sm)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float r = sm[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = fminf(r, sm[i]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// cubic Lagrange (Farrow) interpolation at position t (1 <= t <= n-3), split into the four loads and
// the arithmetic so that a thread can have the loads of several symbols in flight
struct FarrowTaps {
    float2 ym1, y0, y1, y2;
    float mu;
};
__device__ __forceinline__ void farrow_load(const float2 *y, double t, FarrowTaps &f)
{
    const int m = (int)floor(t);   // (callers keep 1 <= t <= n-3; tau is sanitised in k_tetra_sym)
    f.mu = (float)(t - (double)m);
    f.ym1 = y[m - 1];
    f.y0 = y[m];
    f.y1 = y[m + 1];
    f.y2 = y[m + 2];
}
__device__ __forceinline__ float2 farrow_eval(const FarrowTaps &f)
{
    const float2 ym1 = f.ym1, y0 = f.y0, y1 = f.y1, y2 = f.y2;
    const float mu = f.mu;
    float2 r;
    {
        const float c1 = y1.x - ym1.x * (1.f / 3.f) - y0.x * 0.5f - y2.x * (1.f / 6.f);
        const float c2 = (ym1.x + y1.x) * 0.5f - y0.x;
        const float c3 = (y2.x - ym1.x) * (1.f / 6.f) + (y0.x - y1.x) * 0.5f;
        r.x = ((c3 * mu + c2) * mu + c1) * mu + y0.x;
    }
    {
        const float c1 = y1.y - ym1.y * (1.f / 3.f) - y0.y * 0.5f - y2.y * (1.f / 6.f);
        const float c2 = (ym1.y + y1.y) * 0.5f - y0.y;
        const float c3 = (y2.y - ym1.y) * (1.f / 6.f) + (y0.y - y1.y) * 0.5f;
        r.y = ((c3 * mu + c2) * mu + c1) * mu + y0.y;
    }
    return r;
}
__device__ __forceinline__ float2 farrow_at(const float2 *y, double t)
{
    FarrowTaps f;
    farrow_load(y, t, f);
    return farrow_eval(f);
}

// piecewise-linear timing estimate at sample position pos (sub-block centres at (b+0.5)*TB)
__device__ __forceinline__ float tau_at(const float *tau, int nb, double pos)
{
    if (nb == 1) return tau[0];
    const double u = pos / (double)kTimingBlock - 0.5;
    int b0 = (int)floor(u);
    if (b0 < 0) b0 = 0;
    if (b0 > nb - 2) b0 = nb - 2;
    double f = u - (double)b0;
    if (f < 0.0) f = 0.0;
    if (f > 1.0) f = 1.0;
    return (float)((double)tau[b0] * (1.0 - f) + (double)tau[b0 + 1] * f);
}

constexpr int kSymUnroll = 8;   // symbols per thread whose loads are in flight together

__global__ __launch_bounds__(kSymThreads) void k_tetra_sym(const float2 *__restrict__ y,
                                                            const float2 *__restrict__ tstat, const TetraParams P,
                                                            float2 *__restrict__ soft, uint8_t *__restrict__ hard,
                                                            int32_t *n_soft, int32_t *timing_milli, double *min_margin)
{
    __shared__ float Cr[kMaxTimingBlocks + 1], Ci[kMaxTimingBlocks + 1];  // later: prefix sums
    __shared__ float tau[kMaxTimingBlocks];
    __shared__ float sm[kSymThreads / 64];
    __shared__ int k_lo_s, n_sym_s;
    __shared__ float delta_s;
    const int row = blockIdx.x;
    const int n = P.n;
    const double sps = P.sps;
    const float2 *yr = y + (int64_t)row * P.ystride;
    float2 *sr = soft + (int64_t)row * P.max_soft;  // soft symbols (cf32) double as the scratch of step 4
    const int tid = threadIdx.x;
    const int nb = (n + kTimingBlock - 1) / kTimingBlock;
    // 1. square-law timing statistic per sub-block: produced by k_tetra_rrc
    for (int b = tid; b < nb; b += kSymThreads) {
        const float2 c = tstat[(int64_t)row * nb + b];
        Cr[b + 1] = c.x;
        Ci[b + 1] = c.y;
    }
    __syncthreads();
    // 2. prefix sums (one thread), vector average over +-TW sub-blocks and its argument (one thread per
    //    sub-block), unwrap (one thread: a short chain of roundings)
    if (tid == 0) {
        Cr[0] = 0.f; Ci[0] = 0.f;
        float ar = 0.f, ai = 0.f;
        for (int b = 1; b <= nb; ++b) {
            ar += Cr[b];
            ai += Ci[b];
            Cr[b] = ar;
            Ci[b] = ai;
        }
    }
    __syncthreads();
    for (int b = tid; b < nb; b += kSymThreads) {
        const int hi = min(nb, b + kTimingHalfWin + 1), lo = max(0, b - kTimingHalfWin);
        const float cr = Cr[hi] - Cr[lo], ci = Ci[hi] - Ci[lo];
        float tb = -atan2f(ci, cr) * 0.15915494309189535f;  // / (2 pi)
        // a non-finite input sample makes the statistic NaN and the prefix sums carry it to every later sub-block:
        // such a carrier demodulates garbage, but it must not index outside its row (the range checks below are
        // false for NaN)
        if (!(fabsf(tb) <= 1.0f)) tb = 0.f;
        tau[b] = tb;
    }
    __syncthreads();
    if (tid == 0) {
        float prev = 0.f;
        for (int b = 0; b < nb; ++b) {
            float tb = tau[b];
            if (b > 0) tb += rintf(prev - tb);
            tau[b] = tb;
            prev = tb;
        }
        // symbol index range: t_k = (k + tau(k*sps)) * sps must lie in [1, n-3]
        int k_lo = 0;
        while (k_lo < 8 && ((double)k_lo + (double)tau_at(tau, nb, k_lo * sps)) * sps < 1.0) ++k_lo;
        int k_hi = (int)floor((double)n / sps) + 1;
        while (k_hi >= k_lo && ((double)k_hi + (double)tau_at(tau, nb, k_hi * sps)) * sps > (double)n - 3.0) --k_hi;
        int ns = k_hi - k_lo + 1;
        if (ns < 0) ns = 0;
        if (ns > P.max_soft) ns = P.max_soft;
        k_lo_s = k_lo;
        n_sym_s = ns;
    }
    __syncthreads();
    const int k_lo = k_lo_s, ns = n_sym_s;
    // 3. interpolate the matched-filter output at the symbol instants (kSymUnroll symbols per thread with
    //    all their loads in flight: the loops of this kernel are bound by memory latency, not bandwidth)
    constexpr int U = kSymUnroll;
    for (int i0 = tid; i0 < ns; i0 += kSymThreads * U) {
        FarrowTaps f[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kSymThreads;
            if (i < ns) {
                const int k = k_lo + i;
                const double t = ((double)k + (double)tau_at(tau, nb, (double)k * sps)) * sps;
                farrow_load(yr, t, f[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kSymThreads;
            if (i < ns) sr[i] = farrow_eval(f[u]);
        }
    }
    __syncthreads();
    // 4. differential products and the 4th-power carrier-offset estimate (per-thread sums in index order)
    float a4r = 0.f, a4i = 0.f;
    for (int i0 = 1 + tid; i0 < ns; i0 += kSymThreads * U) {
        float2 c[U], p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kSymThreads;
            if (i < ns) { c[u] = sr[i]; p[u] = sr[i - 1]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kSymThreads;
            if (i < ns) {
                const float2 d = make_float2(c[u].x * p[u].x + c[u].y * p[u].y, c[u].y * p[u].x - c[u].x * p[u].y);
                const float2 d2 = cmulf(d, d);
                const float2 d4 = cmulf(d2, d2);
                a4r += d4.x;
                a4i += d4.y;
            }
        }
    }
    a4r = block_sum(a4r, sm);
    a4i = block_sum(a4i, sm);
    if (tid == 0) delta_s = (a4r == 0.f && a4i == 0.f) ? 0.f : atan2f(-a4i, -a4r) * 0.25f;
    __syncthreads();
    float rs, rc;
    sincosf(-delta_s, &rs, &rc);
    // 5. quadrant decision of d_k exp(-i delta): +pi/4 -> 0, +3pi/4 -> 1, -pi/4 -> 2, -3pi/4 -> 3
    float margin = 3.4e38f;
    for (int i0 = 1 + tid; i0 < ns; i0 += kSymThreads * U) {
        float2 c[U], p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kSymThreads;
            if (i < ns) { c[u] = sr[i]; p[u] = sr[i - 1]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kSymThreads;
            if (i < ns) {
                const float2 d = make_float2(c[u].x * p[u].x + c[u].y * p[u].y, c[u].y * p[u].x - c[u].x * p[u].y);
                const float2 dd = make_float2(d.x * rc - d.y * rs, d.x * rs + d.y * rc);
                const uint8_t h = dd.y >= 0.f ? (dd.x >= 0.f ? 0 : 1) : (dd.x >= 0.f ? 2 : 3);
                hard[(int64_t)row * P.max_soft + i - 1] = h;
                // angular distance to the nearest decision boundary (an axis)
                const float ang = atan2f(fabsf(dd.y), fabsf(dd.x));      // 0 .. pi/2
                margin = fminf(margin, fminf(ang, 1.5707963267948966f - ang));
            }
        }
    }
    margin = block_min(margin, sm);
    if (tid == 0) {
        n_soft[row] = ns;
        if (timing_milli) timing_milli[row] = (int32_t)rintf(tau[nb / 2] * 1000.f);
        if (min_margin) min_margin[row] = (double)margin;
    }
}

}  // namespace tdm
