// TETRA mode, Gardner variant (TDM_MODE_TETRA_GARDNER): the textbook receiver BASELINE.json's north_star names --
// RRC matched filter -> Gardner timing-error detector -> proportional-integral loop filter -> period-controlled cubic
// Farrow interpolator -> differential quadrant decisions -- on the device, defined by oracle/tetra_np.py demod_gardner.
//
// The loop is a nonlinear recurrence over a carrier's symbols (every symbol instant depends on the errors of all symbols
// before it), so it cannot be tiled over time like the feed-forward receiver of tetra_kernels.hpp: the parallel axis is
// the CARRIER.  Three kernels:
//   k_tetra_mf       matched filter, LDS-tiled sliding window, fp32, output to HBM (one workgroup per 2048 outputs)
//   k_tetra_gardner  ONE LANE PER CARRIER walks its carrier's symbols; a wavefront's 64 carriers share an LDS ring of
//                    matched-filter samples (four 64-sample chunks per carrier, refilled cooperatively with coalesced
//                    loads one chunk ahead), so that the loop's dependent chain sees LDS latency, not HBM latency
//   k_tetra_decide   differential products, 4th-power carrier-offset estimate, quadrant decisions, margin (one workgroup
//                    per carrier)
// It is the slower receiver by construction and exists because the north-star names it: tests compare it with the fp64
// definition, bench.py times it beside the feed-forward receiver.  Measured (MI355X, 4096 x 32 768): matched filter 0.37 ms
// (5.8 TB/s), loop 2.92 ms, decisions 0.10 ms.  The loop's time does not depend on the number of carriers up to 16 384 (64
// per wavefront, one wavefront per SIMD): 8190 turns x ~860 cycles, a turn being its ~100 vector instructions at one issue
// per four cycles plus the LDS round trip and the loop filter's dependent chain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tetra_params.hpp"


namespace tdm {

#ifndef TDM_MF_TPW
#define TDM_MF_TPW 1   // tiles a workgroup walks, the next one's window in flight (measured: 1 -> 0.371 ms, 2 -> 0.377, 4 -> 0.390, 8 -> 0.424)
#endif
constexpr int kMfThreads = 256, kMfPer = 8, kMfTile = kMfThreads * kMfPer, kMfTilesPerWg = TDM_MF_TPW;

// The RRC stage on its own: y[n] = sum_t h[t] x[n + t - (NT-1)/2], zero outside the chunk (oracle/tetra_np.py
// matched_filter); y rows have pitch y_pitch.  HBM-bound by design (SURVEY 8(d) "unfused": 16 B per sample, 8 in + 8 out,
// for 2 NT multiply-adds).  Measured on one MI355X, 4096 x 32 768 (A/B builds on the same box):
//   * a workgroup's window (2048 + NT - 1 samples) travels HBM -> registers -> LDS with all of a thread's loads in flight
//     at once -- under a branch each is waited for on its own: 2.25 ms instead of 0.60 -- as 16-byte pairs where the whole
//     window lies inside the chunk (8-byte loads everywhere: +2 %);
//   * a thread forms EIGHT CONSECUTIVE outputs from the 8 + NT - 1 samples under them, read from LDS once (40 bytes of LDS
//     traffic per output; with one output per tap-read, 8 NT bytes per output, the kernel was LDS-bound: 0.60 ms,
//     SQ_WAIT_INST_LDS 60 % of the wave cycles), taps in scalar registers, multiply-adds packed over (re, im);
//   * the outputs leave through LDS, transposed, so that consecutive lanes store consecutive 16-byte pairs;
//   * the LDS layout has two pad slots per eight samples: a lane's window starts 80 bytes after its neighbour's, so the
//     lanes of a read fall on different banks and every pair stays 16-byte aligned;
//   * ONE tile per workgroup: 0.371 ms = 5.79 TB/s = 0.72 of 8 TB/s.  (A workgroup walking 2 / 4 / 8 tiles with the next
//     window in flight: 0.377 / 0.390 / 0.424 ms -- with seven workgroups per compute unit the dispatcher's interleaving
//     hides the load phase better than a static walk does.)
template <int NT>
__global__ __launch_bounds__(kMfThreads) void k_tetra_mf(const float2 *__restrict__ x, int64_t in_stride, const TetraParams P,
                                                         float2 *__restrict__ y, int64_t y_pitch)
{
    constexpr int H = (NT - 1) / 2, W = kMfTile + NT - 1, WIN = kMfPer + NT - 1;   // WIN: samples under a thread's outputs
    constexpr int NLD = (W + kMfThreads - 1) / kMfThreads;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float2 xs[W + 2 * (W / 8) + 4];
    auto slot = [](int s) { return s + 2 * (s >> 3); };
    const int row = blockIdx.y, tid = threadIdx.x, n = P.n;
    const float2 *xr = x + (int64_t)row * in_stride;
    float2 *yr = y + (int64_t)row * y_pitch;
    // a workgroup walks kMfTilesPerWg consecutive tiles with the NEXT tile's window already on its way from HBM while it
    // works on the current one: the memory pipes never wait for a workgroup's arithmetic phase
    // interior tiles (the whole window inside the chunk) travel as 16-byte pairs, NLP per thread, lanes past the window
    // masked (they would fetch the next tile's first samples: +11 % of read traffic when they were not); the first and the
    // last tile of a row take 8-byte loads with clamped addresses and zeros outside the chunk
    constexpr int NLP = (W / 2 + kMfThreads - 1) / kMfThreads;
    static_assert(W % 2 == 0 && NLP * 4 >= NLD * 2, "pairs; the registers of the pair path hold the single-sample path too");
    typedef f32x4 __attribute__((aligned(8))) f32x4_a8;   // (rows are 8-byte aligned: pitched channeliser rows)
    f32x4 vp[NLP];
    auto interior = [&](int base) { return base - H >= 0 && base - H + W <= n; };
    auto fetch = [&](int base) {
        if (interior(base)) {
            const f32x4_a8 *pb = (const f32x4_a8 *)(xr + (base - H));
#pragma unroll
            for (int k = 0; k < NLP; ++k) {
                const int pi = tid + k * kMfThreads;
                if (k < NLP - 1 || pi < W / 2) vp[k] = __builtin_nontemporal_load(pb + pi);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int g = base - H + tid + k * kMfThreads;
                const f32x2 q = __builtin_nontemporal_load((const f32x2 *)xr + min(max(g, 0), n - 1));
                if (k & 1) { vp[k >> 1].z = q.x; vp[k >> 1].w = q.y; }
                else { vp[k >> 1].x = q.x; vp[k >> 1].y = q.y; }
            }
        }
    };
    auto stage = [&](int base) {
        if (interior(base)) {
#pragma unroll
            for (int k = 0; k < NLP; ++k) {
                const int pi = tid + k * kMfThreads;
                if (k < NLP - 1 || pi < W / 2) *(f32x4 *)(xs + slot(2 * pi)) = vp[k];   // (an even sample and its successor: adjacent slots)
            }
        } else {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int i = tid + k * kMfThreads, g = base - H + i;
                const float2 q = (k & 1) ? make_float2(vp[k >> 1].z, vp[k >> 1].w) : make_float2(vp[k >> 1].x, vp[k >> 1].y);
                if (i < W) xs[slot(i)] = (g >= 0 && g < n) ? q : make_float2(0.f, 0.f);
            }
        }
    };
    const int tile0 = blockIdx.x * kMfTilesPerWg, ntiles = (n + kMfTile - 1) / kMfTile;
    // Order of a tile's memory operations: its stores are issued right AFTER the next tile's window has been taken out of
    // the load registers, never before -- loads and stores count on one counter (vmcnt), so a wait for loads with younger
    // stores in flight would be a wait for those stores' acknowledgements as well.
    fetch(tile0 * kMfTile);
    stage(tile0 * kMfTile);
    for (int ti = 0; ti < kMfTilesPerWg; ++ti) {
        const int tile = tile0 + ti;
        if (tile >= ntiles) break;
        const int base = tile * kMfTile;
        const bool more = ti + 1 < kMfTilesPerWg && tile + 1 < ntiles;
        __syncthreads();                                  // the tile's window is in LDS
        if (more) fetch(base + kMfTile);
        // the thread's window: samples 8 tid .. 8 tid + WIN - 1 of the staged tile
        f32x2 w[WIN + 1];
        {
            const float2 *p = xs + slot(kMfPer * tid);       // (8 tid is a multiple of 8: the window starts a padded group)
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                f32x2 q = *(const f32x2 *)(p + i + 2 * (i >> 3));
                asm volatile("" : "+v"(q));   // (opaque: read once, kept in registers; 16-byte reads or no pin measure the same)
                w[i] = q;
            }
        }
        f32x2 acc[kMfPer];
#pragma unroll
        for (int j = 0; j < kMfPer; ++j) acc[j] = f32x2{0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float h = P.taps[t];
#pragma unroll
            for (int j = 0; j < kMfPer; ++j) acc[j] += w[j + t] * h;
        }
        // out through LDS: a thread's eight outputs are 64 contiguous bytes, so stored from registers every store
        // instruction would touch 64 lines with 16 bytes each; transposed, consecutive lanes store consecutive 16-byte pairs
        __syncthreads();
        {
            float2 *p = xs + slot(kMfPer * tid);
#pragma unroll
            for (int j = 0; j < kMfPer; j += 2) *(f32x4 *)(p + j) = f32x4{acc[j].x, acc[j].y, acc[j + 1].x, acc[j + 1].y};
        }
        __syncthreads();
        f32x4 q[kMfPer / 2];
#pragma unroll
        for (int k = 0; k < kMfPer / 2; ++k) q[k] = *(const f32x4 *)(xs + slot(2 * (tid + k * kMfThreads)));
        __syncthreads();                                  // the LDS is free for the next window
        if (more) stage(base + kMfTile);
#pragma unroll
        for (int k = 0; k < kMfPer / 2; ++k) {
            const int g = base + 2 * (tid + k * kMfThreads);   // chunk position of the pair
            if (g + 1 < n) __builtin_nontemporal_store(q[k], (f32x4 *)(yr + g));
            else if (g < n) yr[g] = make_float2(q[k].x, q[k].y);
        }
    }
}

// ---- the loop ---------------------------------------------------------------------------------------------------------
constexpr int kGChunk = 64, kGChunks = 4, kGRing = kGChunks * kGChunk, kGPitch = kGRing + 1;   // LDS slots per carrier (+1: rows on different banks)
static_assert((kGRing & (kGRing - 1)) == 0, "ring index by mask");

struct GardnerConsts {
    float k1, k2;      // loop filter gains (oracle/tetra_np.py demod_gardner: Rice eq. C.61, detector gain 2.7, bn_t 0.01, zeta 0.7071)
};

// cubic Lagrange interpolation (the definition's _farrow1: samples at -1, 0, 1, 2 around the whole part m of t).  Sample g of
// a carrier lives in slot g mod 256 of its ring row (a chunk is 64 slots, four chunks are resident), so any index --
// also that of a carrier whose loop has run away -- stays inside the row.
struct GardnerTaps { float2 ym1, y0, y1, y2; };
__device__ __forceinline__ GardnerTaps gardner_taps(const float2 *ring_row, int m)
{
    GardnerTaps g;
    g.ym1 = ring_row[(m - 1) & (kGRing - 1)];
    g.y0 = ring_row[m & (kGRing - 1)];
    g.y1 = ring_row[(m + 1) & (kGRing - 1)];
    g.y2 = ring_row[(m + 2) & (kGRing - 1)];
    return g;
}
__device__ __forceinline__ float2 gardner_eval(const GardnerTaps &g, float mu)
{
    // the cubic through the four samples at -1, 0, 1, 2 as Lagrange weights of mu (the definition's _farrow1 in Horner form is
    // the same polynomial): 11 scalar operations, then four multiply-adds per component -- half the instructions of forming
    // the three Horner coefficients per component (3.11 -> 2.92 ms)
    const float a = mu + 1.f, b = mu - 1.f, c = mu - 2.f;
    const float s1 = (mu * b) * (1.f / 6.f), s2 = (a * c) * 0.5f;
    const float w2 = s1 * a, wm1 = -(s1 * c), w0 = s2 * b, w1 = -(s2 * mu);
    return make_float2(fmaf(g.ym1.x, wm1, fmaf(g.y0.x, w0, fmaf(g.y1.x, w1, g.y2.x * w2))),
                       fmaf(g.ym1.y, wm1, fmaf(g.y0.y, w0, fmaf(g.y1.y, w1, g.y2.y * w2))));
}

__global__ __launch_bounds__(64) void k_tetra_gardner(const float2 *__restrict__ y, int64_t y_pitch, const TetraParams P,
                                                      const GardnerConsts G, int rows, float2 *__restrict__ soft,
                                                      int32_t *__restrict__ n_soft, int32_t *__restrict__ timing_milli)
{
    extern __shared__ float2 ring[];               // [64 carriers][kGPitch]
    const int lane = threadIdx.x;
    const int row = min((int)blockIdx.x * 64 + lane, rows - 1);
    const bool mine = (int)blockIdx.x * 64 + lane < rows;
    const int n = P.n;
    const double sps = P.sps;
    const int back = (int)sps + 4;                 // samples behind floor(t) a strobe may need (mid-symbol strobe + interpolator)
    float2 *my = ring + lane * kGPitch;
    float2 *sr = soft + (int64_t)row * P.max_soft;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef f32x4 __attribute__((aligned(8))) f32x4_a8;   // (a clamped pair at the end of an odd-length row starts on an odd sample)
    // ---- cooperative chunk moves: chunk c = samples [64 c, 64 c + 64) of every carrier of the wavefront.  One 16-byte load
    // fetches two samples; lanes 0..31 serve carrier 2 q, lanes 32..63 carrier 2 q + 1 (rows are 16-byte aligned: even pitch)
    f32x4 pf[32];
    const int half = lane >> 5, l32 = lane & 31;
    auto request = [&](int c) {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int r = min((int)blockIdx.x * 64 + 2 * q + half, rows - 1);
            const int g = kGChunk * c + 2 * l32;
            const int gg = min(g, max(n - 2, 0));                      // (clamped address; masked when it lands)
            pf[q] = *(const f32x4_a8 *)(y + (int64_t)r * y_pitch + gg);
        }
    };
    auto land = [&](int c) {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int g = kGChunk * c + 2 * l32;
            float2 a = make_float2(pf[q].x, pf[q].y), b = make_float2(pf[q].z, pf[q].w);
            if (g >= n) a = make_float2(0.f, 0.f);
            if (g + 1 >= n) b = make_float2(0.f, 0.f);
            if (g == n - 1 && n >= 2) a = make_float2(pf[q].z, pf[q].w);   // (the clamped pair ends at n - 1: its second half is sample n - 1)
            float2 *dst = ring + (2 * q + half) * kGPitch + (g & (kGRing - 1));
            dst[0] = a;
            dst[1] = b;
        }
    };
#pragma unroll 1
    for (int c = 0; c < kGChunks; ++c) { request(c); land(c); }
    int c0 = 0;                                     // oldest resident chunk: chunks c0 .. c0 + kGChunks - 1 are in the ring
    request(kGChunks);                              // in flight while the first symbols are formed
    __syncthreads();
    // ---- per-carrier loop state (oracle/tetra_np.py demod_gardner)
    // the symbol instant t = m + mu, whole samples and a fraction in [0, 1): integer and fp32 arithmetic only in the loop's
    // dependent chain (an fp32 t would be good to 4e-3 samples at the end of a 32 768-sample chunk, an fp64 t puts a dozen
    // long-latency instructions into every turn)
    const float sps_f = (float)sps;
    int m = 1 + (int)floor(sps);
    float mu = (float)(sps - floor(sps));
    float integ = 0.f, pw = 1.f;
    float2 prev = make_float2(0.f, 0.f);
    bool have_prev = false;
    int k = 0;
    double t_mid_sym = 0.0;
    const int m_end = n - 3;                        // t <= n - 3  <=>  m < n - 3 or (m == n - 3 and mu == 0)
    const int k_mid = (int)(0.5 * (double)n / sps);
    auto in_chunk = [&](int mm, float uu) { return mm < m_end || (mm == m_end && uu == 0.f); };
    bool active = mine && in_chunk(m, mu);
    // Turns come in blocks of kGBlock without any wavefront-wide decision inside (a carrier outside the resident chunks
    // simply waits out the turn); whether the ring can move on, and whether anybody is still active, is voted on between
    // blocks.  (With the votes in every turn: 3.62 ms instead of 2.92 for 4096 x 32 768.)
    constexpr int kGBlock = 8;
    const int max_blocks = (4 * P.max_soft + 64) / kGBlock;   // (bounded whatever the input: every turn advances the slowest active carrier)
    for (int blk = 0; blk < max_blocks; ++blk) {
        if (!__any(active)) break;
#pragma unroll
        for (int turn = 0; turn < kGBlock; ++turn) {
            // a carrier takes its strobes when the samples both of them can touch lie in the resident chunks; one that has
            // run ahead of the wavefront's slowest carrier by more than three chunks waits for the ring to move on
            const int mlo = max(m - back, 0), mhi = m + 2;
            const bool ok = active && mlo >= kGChunk * c0 && mhi < kGChunk * (c0 + kGChunks);
            if (ok) {
                // mid-symbol strobe at t - 0.5 sps (1 - integ), not before sample 1 (formed for the first symbol too and not
                // used).  All eight ring reads are issued before either interpolation starts: two independent chains the
                // scheduler interleaves -- with one wavefront per SIMD every dependent instruction's latency is exposed
                const float um = mu - 0.5f * sps_f * (1.f - integ), fm = floorf(um);
                const int m2 = m + (int)fm;
                const GardnerTaps ta = gardner_taps(my, m), tb = gardner_taps(my, max(m2, 1));
                __builtin_amdgcn_sched_barrier(0);
                const float2 sk = gardner_eval(ta, mu);
                const float2 mid = gardner_eval(tb, m2 >= 1 ? um - fm : 0.f);
                float v = 0.f;
                {
                    const float pw_n = 0.99f * pw + 0.01f * (sk.x * sk.x + sk.y * sk.y);
                    const float dx = sk.x - prev.x, dy = sk.y - prev.y;
                    const float e = (dx * mid.x + dy * mid.y) * __builtin_amdgcn_rcpf(fmaxf(pw_n, 1e-12f));
                    const float integ_n = integ + G.k2 * e;
                    pw = have_prev ? pw_n : pw;
                    integ = have_prev ? integ_n : integ;
                    v = have_prev ? G.k1 * e + integ_n : 0.f;
                }
                sr[k] = sk;
                if (k == k_mid) t_mid_sym = (double)m + (double)mu;
                prev = sk;
                have_prev = true;
                ++k;
                const float un = mu + sps_f * (1.f - v), fn = floorf(un);   // (a late strobe makes e positive: shorten the period)
                m += (int)fn;
                mu = un - fn;
                active = in_chunk(m, mu) && k < P.max_soft;
            }
        }
        // the ring moves on while no active carrier needs its oldest chunk any more (wavefront-uniform votes)
#pragma unroll 1
        for (int hop = 0; hop < 2; ++hop) {
            const bool can_drop = !active || (m - back) >= kGChunk * (c0 + 1);
            if (!(__all(can_drop) && __any(active))) break;
            __syncthreads();                         // (one wavefront: orders the ring reads above against the writes below)
            land(c0 + kGChunks);                     // into the slots chunk c0 held
            ++c0;
            request(c0 + kGChunks);
            __syncthreads();
        }
    }
    if (mine) {
        n_soft[row] = k;
        if (timing_milli) {
            const double u = t_mid_sym / sps;
            timing_milli[row] = (int32_t)rint((u - rint(u)) * 1000.0);
        }
    }
}

// ---- decisions (the same detection as demod(): d_k = s_k conj(s_{k-1}), delta = arg(-sum d^4) / 4, quadrant of d e^{-i delta})
__global__ __launch_bounds__(256) void k_tetra_decide(const float2 *__restrict__ soft, int max_soft, const int32_t *__restrict__ n_soft,
                                                      uint8_t *__restrict__ hard, double *__restrict__ min_margin)
{
    __shared__ float sm[3][4];
    __shared__ float delta_s;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float2 *sr = soft + (int64_t)row * max_soft;
    uint8_t *hr = hard + (int64_t)row * max_soft;
    const int ns = n_soft[row];
    // scale: a power of two that brings the carrier's middle symbol to [0.5, 1) (exact; see tetra_kernels.hpp)
    float sc = 1.f;
    if (ns > 0) {
        const float2 smid = sr[ns >> 1];
        const float a = fmaxf(fabsf(smid.x), fabsf(smid.y));
        int ex = 0;
        if (a > 0.f && a < 3.0e38f) (void)frexpf(a, &ex);
        sc = ldexpf(1.f, -ex);
    }
    auto product = [&](int i) {
        const float2 p = sr[i - 1], c = sr[i];
        const float px = p.x * sc, py = p.y * sc, cx = c.x * sc, cy = c.y * sc;
        return make_float2(cx * px + cy * py, cy * px - cx * py);
    };
    float a_pp = 0.f, a_qq = 0.f, a_pq = 0.f;
    for (int i = 1 + tid; i < ns; i += 256) {
        const float2 d = product(i);
        const float p4 = fmaf(d.x, d.x, -(d.y * d.y)), q4 = d.x * d.y;
        a_pp = fmaf(p4, p4, a_pp);
        a_qq = fmaf(q4, q4, a_qq);
        a_pq = fmaf(p4, q4, a_pq);
    }
    float r = fmaf(-4.f, a_qq, a_pp), q = 4.f * a_pq;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { r += __shfl_xor(r, d, 64); q += __shfl_xor(q, d, 64); }
    if (lane == 0) { sm[0][wv] = r; sm[1][wv] = q; }
    __syncthreads();
    if (tid == 0) {
        float rr = 0.f, qq = 0.f;
        for (int w = 0; w < 4; ++w) { rr += sm[0][w]; qq += sm[1][w]; }
        delta_s = (rr == 0.f && qq == 0.f) ? 0.f : atan2f(-qq, -rr) * 0.25f;
    }
    __syncthreads();
    float rs, rc;
    __sincosf(-delta_s, &rs, &rc);
    float mlo = 3.0e38f, mhi = 1.f, hmin = 3.0e38f;
    for (int i = 1 + tid; i < ns; i += 256) {
        const float2 d = product(i);
        const float ddx = d.x * rc - d.y * rs, ddy = d.x * rs + d.y * rc;
        hr[i - 1] = (uint8_t)(((ddy < 0.f) ? 2u : 0u) | ((ddx < 0.f) ? 1u : 0u));
        const float lo = fminf(fabsf(ddx), fabsf(ddy)), hi = fmaxf(fabsf(ddx), fabsf(ddy));
        hmin = fminf(hmin, hi);
        const bool take = lo * mhi < mlo * hi;
        mlo = take ? lo : mlo;
        mhi = take ? hi : mhi;
    }
    const float mratio = hmin == 0.f ? 0.f : mlo / mhi;
    float margin = mratio <= 1.f ? atanf(mratio) : 3.4e38f;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) margin = fminf(margin, __shfl_xor(margin, d, 64));
    if (lane == 0) sm[2][wv] = margin;
    __syncthreads();
    if (tid == 0 && min_margin) min_margin[row] = (double)fminf(fminf(sm[2][0], sm[2][1]), fminf(sm[2][2], sm[2][3]));
}

}  // namespace tdm
