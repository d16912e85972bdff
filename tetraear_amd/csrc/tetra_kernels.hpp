// TETRA mode (north-star receiver): per-carrier pi/4-DQPSK demodulation of channelised baseband.
//
// There is no reference implementation of this mode: the algorithm is defined by
// oracle/tetra_np.py (fp64 numpy) and restated here in fp32 for gfx950 as ONE kernel, k_tetra_fused:
// a workgroup owns a carrier and walks its chunk tile by tile,
//   HBM -> registers (next tile in flight) -> split into bf16 halves -> LDS -> RRC matched filter on the matrix cores
//          (v_mfma_f32_16x16x32_bf16, fp32 accumulation: a Toeplitz tile of the taps times 16 runs of the staged input;
//          samples and taps as sums of two bf16, three products)
//       -> square-law (Oerder-Meyr) timing statistic of the tile's sub-blocks from the accumulators
//       -> matched-filter output into an LDS ring (never to HBM)
//       -> timing estimates of the sub-blocks whose averaging window is complete
//       -> cubic Farrow interpolation at the symbol instants out of the ring -> soft symbols to HBM,
// then the 4th-power carrier-offset estimate over the carrier's symbols and the differential quadrant
// decision.  HBM traffic = the input once + 9 bytes per symbol (SURVEY 8(d) "fused": R*8 + 8 + 1 B/symbol).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "tetra_params.hpp"

namespace tdm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The ring of matched-filter outputs is plain: its writers store 16 consecutive samples per 16 lanes, and the four-way
// bank conflicts of the symbol stage's strided reads (eight 8-byte reads per wavefront and tile) cost less than the
// LDS a padded ring takes from the second workgroup of a compute unit.
__device__ __forceinline__ constexpr int rrc_slot(int s) { return s; }

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

// cubic Lagrange (Farrow) interpolation between y0 and y1 at fraction mu
struct FarrowTaps {
    float2 ym1, y0, y1, y2;
    float mu;
};
__device__ __forceinline__ float2 farrow_eval(const FarrowTaps &f)
{
    // the cubic through the four samples at -1, 0, 1, 2 as Lagrange weights of mu (11 scalar operations), then four
    // packed multiply-adds on the (re, im) pairs as they come out of LDS
    const float mu = f.mu, a = mu + 1.f, b = mu - 1.f, c = mu - 2.f;
    const float s1 = (mu * b) * (1.f / 6.f), s2 = (a * c) * 0.5f;
    const float w2 = s1 * a, wm1 = -(s1 * c), w0 = s2 * b, w1 = -(s2 * mu);
    f32x2 r = f32x2{f.ym1.x, f.ym1.y} * wm1;
    r += f32x2{f.y0.x, f.y0.y} * w0;
    r += f32x2{f.y1.x, f.y1.y} * w1;
    r += f32x2{f.y2.x, f.y2.y} * w2;
    return make_float2(r.x, r.y);
}
// piecewise-linear timing estimate at sample position pos (sub-block centres at (b+0.5)*TB)
__device__ __forceinline__ float tau_at(const float2 *tau, int nb, double pos)
{
    if (nb == 1) return tau[0].x;   // (tau: ring of kTauRing estimates, each with its successor: see the kernel)
    const double u = pos / (double)kTimingBlock - 0.5;
    int b0 = (int)floor(u);
    if (b0 < 0) b0 = 0;
    if (b0 > nb - 2) b0 = nb - 2;
    double f = u - (double)b0;
    if (f < 0.0) f = 0.0;
    if (f > 1.0) f = 1.0;
    return (float)((double)tau[b0 & (kTauRing - 1)].x * (1.0 - f) + (double)tau[(b0 + 1) & (kTauRing - 1)].x * f);
}


// clamp(x, LO, hi) for a small constant LO and a wave-uniform hi >= LO in ONE instruction (the compiler emits v_min + v_max:
// it cannot know hi >= LO)
template <int LO>
__device__ __forceinline__ int clamp_med3(int x, int hi_uniform)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(LO), "s"(hi_uniform));
    return r;
}

// atan2(y, x) / (2 pi), absolute error below 2e-6 turns (odd minimax polynomial on [0, 1] + octant folding): the timing
// estimate it feeds is good to 1e-3 symbols at best
__device__ __forceinline__ float atan2_turns(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float hi = fmaxf(ax, ay), lo = fminf(ax, ay);
    const float a = hi > 0.f ? lo * __builtin_amdgcn_rcpf(hi) : 0.f;   // (v_rcp_f32: 1 ulp; __fdividef is a full division here)
    const float s = a * a;
    float r = a * (0.99997726f + s * (-0.33262347f + s * (0.19354346f + s * (-0.11643287f + s * (0.05265332f + s * -0.01172120f)))));
    r *= 0.15915494309189535f;
    if (ay > ax) r = 0.25f - r;
    if (x < 0.f) r = 0.5f - r;
    return y < 0.f ? -r : r;
}

// waves per SIMD the register allocation aims for (LDS holds four workgroups of 256 threads or two of 512)
#define TDM_TETRA_WAVES(NT) 4


// Split-bf16 arithmetic of the matched filter: a float is the sum of two bf16 (16 significant bits), a product of two
// such sums keeps its three leading terms; the matrix cores multiply bf16 exactly and accumulate in fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t pk_bf16(float a, float b)   // (bf16(a), bf16(b)) in one dword, round to nearest even
{
    const bf16x2 v = __builtin_convertvector((f32x2{a, b}), bf16x2);
    return __builtin_bit_cast(uint32_t, v);
}
// leading and trailing bf16 halves of two floats: hi = (bf16(a), bf16(b)), lo = (bf16(a - hi_a), bf16(b - hi_b))
__device__ __forceinline__ void split_bf16(float a, float b, uint32_t &hi, uint32_t &lo)
{
    hi = pk_bf16(a, b);
    const float a1 = __builtin_bit_cast(float, hi << 16), b1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pk_bf16(a - a1, b - b1);
}

// FMT8 (round 6; north_star: "coalesced complex-int8/float loads"): 0 = cf32 input; 1 = cu8, 2 = cs8 -- 2 bytes per sample
// through HBM instead of 8 (algorithmic bytes per symbol at 4 samples/symbol: 17 instead of 41).  An 8-bit sample is EXACT in
// one bf16 -- cu8 as the odd integer 2u - 255 (|.| <= 255: eight significant bits), cs8 as it is -- so the staged window has
// ONE plane per component instead of a leading and a trailing one and the matched filter two matrix-core products per step
// instead of four; the format's scale (1/255: x = u / 127.5 - 1 = (2u - 255) / 255; 1/128) multiplies the soft symbols where
// they are stored -- everything between (timing statistic, estimates, interpolation) does not depend on the scale.
template <int FMT8>
struct TetraIn8 {
    static constexpr float scale = FMT8 == 1 ? 1.f / 255.f : (FMT8 == 2 ? 1.f / 128.f : 1.f);
    __device__ __forceinline__ static float2 conv(uint32_t h)   // low 16 bits: I, Q -> the integers the filter runs on
    {
        if (FMT8 == 1) return make_float2(fmaf((float)(h & 255u), 2.f, -255.f), fmaf((float)((h >> 8) & 255u), 2.f, -255.f));   // (v_cvt_f32_ubyteN + one fma: exact)
        return make_float2((float)(int8_t)(h & 255u), (float)(int8_t)((h >> 8) & 255u));
    }
};

template <int NT, int FMT8 = 0>
__global__ __launch_bounds__(kRrcThreads, TDM_TETRA_WAVES(NT)) void k_tetra_fused(const void *__restrict__ x_, int64_t in_stride,
                                                              const TetraParams P, float2 *__restrict__ soft,
                                                              uint8_t *__restrict__ hard, int32_t *n_soft,
                                                              int32_t *timing_milli, double *min_margin,
                                                              const int32_t *__restrict__ row_list, const int32_t *__restrict__ n_rows)
{
    static_assert(kRrcPerThread == 8 && kRrcThreads % 64 == 0 && kTimingBlock == 256, "a wavefront owns two timing sub-blocks of a tile");
    static_assert(kRing % kTimingBlock == 0 && kRing - kRrcTile - kTimingBlock * (2 * kTimingHalfWin + 1) / 2 >= kTimingBlock / 2, "ring too short");
    constexpr int PER = kRrcPerThread;
    constexpr int HALO = NT - 1, H2 = HALO / 2;
    constexpr int KS = (kRrcRun + HALO + 31) / 32;        // matrix-core steps (32 window positions each) per run of 16 outputs
    constexpr int NS = kRrcTile - kRrcRun + 32 * KS;      // samples staged per tile: base - H2 .. base - H2 + NS
    constexpr int NP = (NS / 2 + kRrcThreads - 1) / kRrcThreads;   // 16-byte sample pairs per thread
    constexpr int PLANE = (NS / 2 + 3) / 4 * 4;           // dwords per plane of bf16 pairs (16-byte operand loads: a multiple of 4)
    static_assert(NS % 2 == 0 && kRrcThreads * (NP - 1) < NS / 2, "only the last pair of a thread can fall outside the staged window");
    // staged input, four planes of bf16: leading / trailing halves of the real parts, then of the imaginary parts; sample
    // q of a plane is half q of the plane's dwords.  (No padding: a 16-byte operand load of lane l starts at sample
    // 16 (l & 15) + 8 (l >> 4) of its block, and the lane groups the LDS serves together cover 256 distinct bytes.)
    __shared__ __attribute__((aligned(16))) uint32_t xsb[4 * PLANE];
    __shared__ float2 yring[kRing + 4];   // + the first three samples again past the end: a symbol's four never wrap
    constexpr int kCstRing = 4 * kTileBlocks;   // the statistic of the previous and the current tile's sub-blocks, and zeros in the next tile's slots
    __shared__ float2 Cst[kCstRing];
    // timing estimates: slot b holds (tau_b, tau_{b+1}) -- the pair a symbol's interpolation needs comes out of one 8-byte
    // read; the second half of a slot is written when the next estimate is formed (same round, or the first lane of the
    // next round's duty wavefront), the last sub-block's successor is itself
    __shared__ float2 tau[kTauRing];
    __shared__ float tau_mid_s;
    __shared__ float sm[kRrcThreads / 64], sm2[kRrcThreads / 64];
    __shared__ float delta_s;
    // row_list (the wideband chain's occupancy gate, occupancy_kernels.hpp): workgroup i takes row row_list[i] of the batch
    // -- input row, output rows and all -- and the workgroups past the list's length leave at once
    int row = blockIdx.x;
    if (row_list) {
        if (row >= *n_rows) return;
        row = row_list[row];
    }
    const int tid = threadIdx.x;
    const int n = P.n;
    const double sps = P.sps;
    const float2 *xr = (const float2 *)x_ + (int64_t)row * in_stride;     // rows of the channeliser may carry a pitch
    const uint16_t *xr8 = (const uint16_t *)x_ + (int64_t)row * in_stride;   // (FMT8: one 2-byte sample per element)
    float2 *sr = soft + (int64_t)row * P.max_soft;
    const int nb = (n + kTimingBlock - 1) / kTimingBlock;
    const int ntiles = (n + kRrcTile - 1) / kRrcTile;
    // ---- input: a tile's samples base - H2 .. base - H2 + NS (zero outside the chunk) travel HBM -> registers one
    // tile ahead of their use, as 16-byte pairs.  A row starts on an 8-byte boundary (pitched channeliser rows, odd
    // offsets, odd tap half-lengths), so a pair may lie across two 16-byte segments: the loads are declared 8-byte
    // aligned, a wavefront's 1 KB then touches nine cache lines instead of eight, and nothing downstream depends on where
    // the caller's buffer lies.  The one sample a clamped pair can miss at either end of the chunk is read once up
    // front.  fetch() only issues loads (clamped addresses, nothing that consumes a loaded value: a use would wait for
    // the load on the spot); stage() masks what lies outside the chunk, splits into bf16 and writes LDS.
    typedef f32x4 __attribute__((aligned(8))) f32x4_a8;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the staged window is NP - 1 pairs for every thread and a few more (24 at 33 taps) that fall to the first lanes of
    // wavefront 0: the other wavefronts skip the last pair altogether (load, split and all)
    static_assert(NS / 2 - (NP - 1) * kRrcThreads <= 64, "the pairs of the last turn lie in wavefront 0");
    const bool last_turn = wv == 0;
    const int gmaxp = n - 2;                              // last pair start inside the chunk
    const float2 x_first = FMT8 ? make_float2(0.f, 0.f) : xr[0], x_last = FMT8 ? make_float2(0.f, 0.f) : xr[n - 1];
    f32x4 pf[NP];
    uint32_t pf8[NP];   // (FMT8) a pair of consecutive samples as it arrives: low half the first
    auto fetch = [&](int tile) {
        const int g0 = tile * kRrcTile - H2;
        if (FMT8) {
            if (g0 >= 0 && g0 + 2 * (NP * kRrcThreads - 1) + 1 <= n - 1 && (((uintptr_t)(xr8 + g0)) & 3) == 0) {
                // an inner tile whose pairs are 4-byte aligned: one load per pair
                const uint32_t *pb = (const uint32_t *)(xr8 + g0);
#pragma unroll
                for (int j = 0; j < NP - 1; ++j) pf8[j] = __builtin_nontemporal_load(pb + tid + j * kRrcThreads);
                if (last_turn) pf8[NP - 1] = __builtin_nontemporal_load(pb + tid + (NP - 1) * kRrcThreads);
                return;
            }
            // else two 2-byte loads per pair from clamped positions (a row of bytes has no alignment to speak of); stage() masks
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                if (j == NP - 1 && !last_turn) break;
                const int g = g0 + 2 * (tid + j * kRrcThreads);
                const uint32_t a = xr8[min(max(g, 0), n - 1)], b = xr8[min(max(g + 1, 0), n - 1)];
                pf8[j] = a | (b << 16);
            }
            return;
        }
        if (g0 >= 0 && g0 + 2 * (NP * kRrcThreads - 1) <= gmaxp) {
            // an inner tile: nothing to clamp, one scalar base and the thread's own offset (no vector address arithmetic)
            const f32x4_a8 *pb = (const f32x4_a8 *)(xr + g0);
#pragma unroll
            for (int j = 0; j < NP - 1; ++j) pf[j] = __builtin_nontemporal_load(pb + tid + j * kRrcThreads);
            if (last_turn) pf[NP - 1] = __builtin_nontemporal_load(pb + tid + (NP - 1) * kRrcThreads);
            return;
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if (j == NP - 1 && !last_turn) break;
            const int g = g0 + 2 * (tid + j * kRrcThreads);
            pf[j] = __builtin_nontemporal_load((const f32x4_a8 *)(xr + min(max(g, 0), gmaxp)));
        }
    };
    // a loaded pair of consecutive samples (pair idx = staged samples 2 idx, 2 idx + 1) into the four planes
    auto put = [&](int idx, bool last_pair, float re0, float im0, float re1, float im1) __attribute__((always_inline)) {
        uint32_t w[4];
        split_bf16(re0, re1, w[0], w[1]);
        split_bf16(im0, im1, w[2], w[3]);
        if (!last_pair || idx < NS / 2) {
#pragma unroll
            for (int pl = 0; pl < 4; ++pl) xsb[pl * PLANE + idx] = w[pl];
        }
    };
    auto stage = [&](int tile) __attribute__((always_inline)) {
        const int g0 = tile * kRrcTile - H2;       // chunk position of the first pair
        if (FMT8) {
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                if (j == NP - 1 && !last_turn) break;
                const int idx = tid + j * kRrcThreads, g = g0 + 2 * idx;
                const float2 e0 = (g >= 0 && g < n) ? TetraIn8<FMT8>::conv(pf8[j]) : make_float2(0.f, 0.f);
                const float2 e1 = (g + 1 >= 0 && g + 1 < n) ? TetraIn8<FMT8>::conv(pf8[j] >> 16) : make_float2(0.f, 0.f);
                if (j < NP - 1 || idx < NS / 2) {   // one plane per component: the integers are exact in bf16
                    xsb[idx] = pk_bf16(e0.x, e1.x);
                    xsb[2 * PLANE + idx] = pk_bf16(e0.y, e1.y);
                }
            }
            return;
        }
        if (g0 >= 0 && g0 + NS <= gmaxp) {         // every pair of the tile lies inside the chunk: no masks
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                if (j == NP - 1 && !last_turn) break;
                const f32x4 v = pf[j];
                put(tid + j * kRrcThreads, j == NP - 1, v.x, v.y, v.z, v.w);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if (j == NP - 1 && !last_turn) break;
            const int idx = tid + j * kRrcThreads;
            const int g = g0 + 2 * idx;               // chunk position of the pair's first sample
            if (g0 + 2 * (64 * wv + j * kRrcThreads) >= n) {
                // the wavefront's 64 pairs of this turn all lie past the end of the chunk: zeros, nothing to split (the fifth
                // tile of an 8389-sample chunk: 13 of its 17 wavefront-turns)
                if (j < NP - 1 || idx < NS / 2) {
#pragma unroll
                    for (int pl = 0; pl < 4; ++pl) xsb[pl * PLANE + idx] = 0u;
                }
                continue;
            }
            const bool in = g >= 0 && g <= gmaxp;
            const f32x4 v = pf[j];
            const float2 e0 = in ? make_float2(v.x, v.y) : (g == n - 1 ? x_last : make_float2(0.f, 0.f));
            const float2 e1 = in ? make_float2(v.z, v.w) : (g == -1 ? x_first : make_float2(0.f, 0.f));
            put(idx, j == NP - 1, e0.x, e0.y, e1.x, e1.y);
        }
    };

    // ---- matched filter on the matrix cores.  A wavefront owns 512 consecutive outputs of a tile = two timing sub-blocks
    // of 16 runs of 16:  Y[J][i] = y[16 J + i] = sum_m X[J][m] T[m][i],  X[J][m] = staged sample 16 J + m,
    // T[m][i] = h[m - i] (Toeplitz, zero outside the taps).
    // Data and taps are split into two bf16 each, x = x1 + x2, h = h1 + h2 (the taps ARE such sums: 16-bit coefficients,
    // oracle/tetra_np.py coeff16), and all four products x2 h2 + x2 h1 + x1 h2 + x1 h1 go through
    // v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 4 x KS instructions per 16 x 16 outputs at 16x the fp32 rate.
    // What is left is the third chunk of a sample (2^-18 of |x|): soft symbols within 6.2e-6 of the largest symbol of the
    // fp64 definition over 22 065 random carriers (median 3.3e-6; the fp32 chain around the filter is ~1e-6 of that).
    // (The fp32-input MFMA computes the same sums exactly but runs on the vector ALUs' own multipliers: measured, its
    // 48 instructions per wavefront and tile add their full 0.20 ms to the 0.37 ms of the rest of this kernel.)
    // Lane l supplies X[l & 15][32 s + 8 (l >> 4) .. + 7] (16 bytes per plane and step from LDS) and the constants
    // T[32 s + 8 (l >> 4) .. + 7][l & 15]; it receives Y[4 (l >> 4) + r][l & 15], r < 4.
    u32x4 hB1[KS], hB2[KS];
    if (P.tap_ops) {   // (the plan's table: two 16-byte loads per step instead of the split below)
        const u32x4 *tp = (const u32x4 *)P.tap_ops + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            hB1[s] = tp[(2 * s) * 64];
            hB2[s] = tp[(2 * s + 1) * 64];
        }
    } else
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        uint32_t w1[4], w2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float hv[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int t = 32 * s + 8 * (lane >> 4) + 2 * j + c - (lane & 15);
                const float h = P.taps[min(max(t, 0), NT - 1)];
                hv[c] = (t >= 0 && t < NT) ? h : 0.f;
            }
            split_bf16(hv[0], hv[1], w1[j], w2[j]);
        }
        hB1[s] = u32x4{w1[0], w1[1], w1[2], w1[3]};
        hB2[s] = u32x4{w2[0], w2[1], w2[2], w2[3]};
    }
    const u32x4 *ab = (const u32x4 *)(xsb + 256 * wv + 8 * (lane & 15) + 4 * (lane >> 4));
    const int out_off = 64 * (lane >> 4) + (lane & 15);   // the lane's outputs inside a sub-block: out_off + 16 r
    const int ring_lane = rrc_slot(out_off);

    // symbol-clock phasor exp(-2 pi i g / sps) at the lane's first output of the current tile
    float pc, ps;
    {
        const double ph = (double)(512 * wv + out_off) * P.inv_sps;
        sincospif(-2.f * (float)(ph - floor(ph)), &ps, &pc);
    }
    uint64_t sps40 = (uint64_t)(sps * 1099511627776.0);          // samples per symbol, 40 fraction bits
    {
        // into scalar registers (the builtin is folded away on a value the compiler already knows to be uniform, and the
        // uniform products of the symbol stage would stay on the vector ALU: 64-bit multiplies at a quarter of its rate)
        uint32_t vlo = (uint32_t)sps40, vhi = (uint32_t)(sps40 >> 32), slo, shi;
        asm volatile("v_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3" : "=s"(slo), "=s"(shi) : "v"(vlo), "v"(vhi));
        sps40 = ((uint64_t)shi << 32) | slo;
    }
// (Round 4 also built this kernel with differential products, 4th-power sums and provisional one-byte decisions formed in the
// symbol stage, so that the final pass read one byte per symbol back instead of the 8-byte soft symbols: traffic 1.24x -> 1.05x
// algorithmic, 0.354 ms against 0.314 -- the ~30 vector instructions per symbol it added to the tile loop cost more than the
// read-back it removed.  Removed from the source in round 6; docs/HISTORY.md A.2.)
    constexpr int NW = kRrcThreads / 64;
    constexpr int WSYM = 64;                                     // symbols per wavefront and turn of the symbol stage
    constexpr int TSYM = WSYM * NW;                              // symbols per turn of the workgroup
    const int off = WSYM * wv + lane;                            // the thread's symbol is k = kb + off + TSYM u
    uint64_t ptid = (uint64_t)(uint32_t)off * sps40;             // the thread's share of its symbols' nominal positions
    {
        // (opaque: otherwise the compiler folds it back into (k_uniform + tid) * sps40, a 64-bit vector multiply per symbol)
        uint32_t plo = (uint32_t)ptid, phi = (uint32_t)(ptid >> 32);
        asm volatile("" : "+v"(plo), "+v"(phi));
        ptid = ((uint64_t)phi << 32) | plo;
    }
    float tau_prev = 0.f;   // (wave 0) last unwrapped estimate
    int b_done = 0;         // sub-blocks whose estimate is final
    int k_lo = 0, k_begin = 0, ns = 0;
    float a_pp = 0.f, a_qq = 0.f, a_pq = 0.f, sc = 1.f;   // running sums of d^4 and the products' power-of-two scale (final passes)
    uint8_t *const hr = hard + (int64_t)row * P.max_soft;

    static_assert(kTimingHalfWin <= kTileBlocks, "the slots below sub-block 0 must be free during the first tile");
    if (tid < kCstRing) Cst[tid] = make_float2(0.f, 0.f);   // (made visible by the first barrier of the loop)
    fetch(0);
    stage(0);
    if (ntiles > 1) fetch(1);
    for (int i = 0; i < ntiles; ++i) {
        const bool last = i == ntiles - 1;
        const int base = i * kRrcTile;
        __syncthreads();   // tile i staged; the previous round's ring reads are done
        // (a wavefront whose 512 outputs of the last tile all lie past the end of the chunk has no filter output, no
        // statistic and nothing for the ring: a chunk of 8389 samples -- the 10 MS/s channeliser's -- ends 197 samples
        // into its fifth tile, and three of the four wavefronts skip it)
        const bool wave_data = !last || base + 512 * wv < n;
        if (wave_data) {
        f32x4 cre[2], cim[2];
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            cre[bb] = f32x4{0.f, 0.f, 0.f, 0.f};
            cim[bb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#define TDM_TETRA_MFMA_REPS 1   // (experiment hook: 0 / 2 / 3 price the matched filter's share of the kernel)
#pragma unroll
        for (int s_ = 0; s_ < KS * TDM_TETRA_MFMA_REPS; ++s_) {
            const int s = s_ % KS;
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int o = 32 * bb + 4 * s;   // (16-byte units: a sub-block is 128 dwords of a plane, a step 16)
                const bf16x8 r1 = __builtin_bit_cast(bf16x8, ab[o]), r2 = __builtin_bit_cast(bf16x8, ab[PLANE / 4 + o]);
                const bf16x8 i1 = __builtin_bit_cast(bf16x8, ab[2 * (PLANE / 4) + o]), i2 = __builtin_bit_cast(bf16x8, ab[3 * (PLANE / 4) + o]);
                const bf16x8 h1 = __builtin_bit_cast(bf16x8, hB1[s]), h2 = __builtin_bit_cast(bf16x8, hB2[s]);
                if (!FMT8) {   // (8-bit input: no trailing halves)
                    cre[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r2, h2, cre[bb], 0, 0, 0);   // (smallest terms first)
                    cim[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(i2, h2, cim[bb], 0, 0, 0);
                    cre[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r2, h1, cre[bb], 0, 0, 0);
                    cim[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(i2, h1, cim[bb], 0, 0, 0);
                }
                cre[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r1, h2, cre[bb], 0, 0, 0);
                cim[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(i1, h2, cim[bb], 0, 0, 0);
                cre[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(r1, h1, cre[bb], 0, 0, 0);
                cim[bb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(i1, h1, cim[bb], 0, 0, 0);
            }
        }
        // ---- square-law timing statistic of the wavefront's two sub-blocks, C_b = sum |y[g]|^2 exp(-2 pi i g / sps): the
        // phasor of a lane's four outputs of a sub-block is a constant table times the lane's own
        {
            float pw[8], acc4[4];
#pragma unroll
            for (int v = 0; v < 8; ++v) pw[v] = cre[v >> 2][v & 3] * cre[v >> 2][v & 3] + cim[v >> 2][v & 3] * cim[v >> 2][v & 3];
            if (last) {
                asm volatile("" : "+v"(pw[0]));   // (a real branch: only the last tile pays for the masks)
#pragma unroll
                for (int v = 0; v < 8; ++v)
                    if (base + 512 * wv + 256 * (v >> 2) + out_off + 16 * (v & 3) >= n) pw[v] = 0.f;
            }
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                float qr = 0.f, qi = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qr = fmaf(pw[4 * bb + r], P.ev_c[4 * bb + r], qr);
                    qi = fmaf(pw[4 * bb + r], P.ev_s[4 * bb + r], qi);
                }
                acc4[2 * bb] = qr * pc - qi * ps;
                acc4[2 * bb + 1] = qr * ps + qi * pc;
            }
            const float nc = pc * P.tile_c - ps * P.tile_s, nsn = pc * P.tile_s + ps * P.tile_c;
            pc = nc;
            ps = nsn;
            // sum of the four values over the wavefront without LDS round trips, halving the number of live values in the
            // first two steps: after them a lane holds the quad's sum of value (lane & 3) (0: re of the first sub-block, 1: re
            // of the second, 2 / 3: the imaginary parts); then the quads of a row (row rotations keep lane & 3) and the four
            // rows (permlane swaps).  17 vector instructions instead of the 44 of four separate butterflies.
#define TDM_DPP(X, CTRL) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, X), CTRL, 0xf, 0xf, true))
            const bool odd = lane & 1, hi = lane & 2;
            const float s0 = odd ? acc4[2] : acc4[0], t0 = odd ? acc4[0] : acc4[2];
            const float s1 = odd ? acc4[3] : acc4[1], t1 = odd ? acc4[1] : acc4[3];
            const float y0 = s0 + TDM_DPP(t0, 0xB1), y1 = s1 + TDM_DPP(t1, 0xB1);   // quad_perm [1,0,3,2]
            const float sz = hi ? y1 : y0, tz = hi ? y0 : y1;
            float z = sz + TDM_DPP(tz, 0x4E);                                      // quad_perm [2,3,0,1]
            z += TDM_DPP(z, 0x124);                                                // row_ror:4
            z += TDM_DPP(z, 0x128);                                                // row_ror:8
#undef TDM_DPP
            {
                auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, z), __builtin_bit_cast(unsigned, z), false, false);
                const unsigned r0 = r[0], r1 = r[1];   // (scalar copies: see the note at the final passes)
                z = __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
            }
            {
                auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, z), __builtin_bit_cast(unsigned, z), false, false);
                const unsigned r0 = r[0], r1 = r[1];
                z = __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
            }
            // (sub-blocks past the end of the chunk, in this tile and in the next one's slots, are stored as zero: the
            // averaging window below reads its +-TW neighbours without masks)
            const int b = i * kTileBlocks + 2 * wv + (lane & 1);
            if (lane < 4) ((float *)Cst)[2 * (b & (kCstRing - 1)) + (lane >> 1)] = b < nb ? z : 0.f;
            if (last && lane < 4) ((float *)Cst)[2 * ((b + kTileBlocks) & (kCstRing - 1)) + (lane >> 1)] = 0.f;
        }
        // ---- matched-filter output into the ring (the ring is a whole number of sub-blocks: a sub-block's 256 outputs
        // never straddle its end)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            const int pw = (base + 512 * wv + 256 * bb) % kRing;
            float2 *yw = yring + rrc_slot(pw) + ring_lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) yw[rrc_slot(16 * r)] = make_float2(cre[bb][r], cim[bb][r]);
            if (pw == 0 && lane < 3) yring[rrc_slot(kRing) + lane] = make_float2(cre[bb][0], cim[bb][0]);
        }
        } else if (lane < 4) {
            const int b = i * kTileBlocks + 2 * wv + (lane & 1);
            ((float *)Cst)[2 * (b & (kCstRing - 1)) + (lane >> 1)] = 0.f;
            ((float *)Cst)[2 * ((b + kTileBlocks) & (kCstRing - 1)) + (lane >> 1)] = 0.f;
        }
        __syncthreads();   // ring, statistic visible; staging buffer free
        // ---- timing estimates that are final now: vector average over +-TW sub-blocks, argument, unwrap.  ONE wavefront
        // computes them (the duty rotates with the tile, so that over a carrier every SIMD of the compute unit does a
        // quarter of this work) while the other three stage the next tile; a third barrier hands the estimates over.
        // (Round 2 had every wavefront compute them, identically: no barrier, but ~75 vector instructions per wavefront and
        // tile, an eighth of the loop, spent three times over.)
        const bool est_duty = wv == (i & (kRrcThreads / 64 - 1));
        const int b_known = last ? nb - 1 : (i + 1) * kTileBlocks - 1 - kTimingHalfWin;
        if (!est_duty && !last) {
            stage(i + 1);
            if (i + 2 < ntiles) fetch(i + 2);
        }
        if (est_duty) {
            if (b_done > 0) tau_prev = tau[(b_done - 1) & (kTauRing - 1)].x;   // (the last estimate of the previous duty wavefront)
            const int lane = tid & 63;
            const int cnt = b_known - b_done + 1;   // <= kTileBlocks + kTimingHalfWin
            const int b = b_done + lane;
            float cr = 0.f, ci = 0.f;
#pragma unroll
            for (int e = 0; e < 2 * kTimingHalfWin + 1; ++e) {   // (slots outside [0, nb) hold zeros)
                const float2 c = Cst[(b - kTimingHalfWin + e) & (kCstRing - 1)];
                cr += c.x;
                ci += c.y;
            }
            float tb = -atan2_turns(ci, cr);
            // a non-finite input sample makes the statistic NaN: such a carrier demodulates garbage, but it must
            // not index outside its row
            if (!(fabsf(tb) <= 1.0f) || lane >= cnt) tb = 0.f;
            // unwrap: tau_b = raw_b + N_b, N_b = N_{b-1} + rint(raw_{b-1} - raw_b) (whole symbols), as an inclusive scan
            // over the (<= 10) new sub-blocks in the first row of the wavefront (DPP row shifts)
            const float rawp = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tb), 0x138, 0xf, 0xf, false));   // wave_shr:1 (the new sub-blocks may reach into the second row of lanes)
            float stp = lane == 0 ? (b_done > 0 ? rintf(tau_prev - tb) : 0.f) : rintf(rawp - tb);
#define TDM_ROW_SHR_ADD(D) stp += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, stp), 0x110 + D, 0xf, 0xf, true));   // row_shr:D, zero shifted in
            TDM_ROW_SHR_ADD(1) TDM_ROW_SHR_ADD(2) TDM_ROW_SHR_ADD(4) TDM_ROW_SHR_ADD(8)
#undef TDM_ROW_SHR_ADD
            if (kTileBlocks + kTimingHalfWin > 16)   // more new sub-blocks than a row of 16 lanes: the first row's total enters the second
                stp += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, stp), 0x142, 0x2, 0xf, false));   // row_bcast:15 into row 1
            tb += stp;
            tau_prev = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tb), cnt - 1));
            {
                // the successor's estimate from the lane to the right (wave_shl:1); the last new sub-block has none yet
                const float nxt = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, tb), 0x130, 0xf, 0xf, false));
                if (lane < cnt) {
                    tau[b & (kTauRing - 1)] = make_float2(tb, (lane == cnt - 1) ? tb : nxt);
                    if (b == nb / 2) tau_mid_s = tb;
                }
                if (lane == 0 && b_done > 0) tau[(b_done - 1) & (kTauRing - 1)].y = tb;
            }
        }
        b_done = b_known + 1;
        __syncthreads();   // estimates visible
        if (est_duty && !last) {
            stage(i + 1);
            if (i + 2 < ntiles) fetch(i + 2);
        }
        // ---- symbols whose two timing estimates are final: t_k = (k + tau(k sps)) sps, in [1, n-3]
        if (i == 0) {
            while (k_lo < 8 && ((double)k_lo + (double)tau_at(tau, nb, k_lo * sps)) * sps < 1.0) ++k_lo;
            k_begin = k_lo;
        }
        int k_end;
        if (last) {
            int k_hi = (int)floor((double)n / sps) + 1;
            while (k_hi >= k_lo && ((double)k_hi + (double)tau_at(tau, nb, k_hi * sps)) * sps > (double)n - 3.0) --k_hi;
            ns = k_hi - k_lo + 1;
            if (ns < 0) ns = 0;
            if (ns > P.max_soft) ns = P.max_soft;
            k_end = k_lo + ns;
        } else {
            // the symbols whose nominal position k sps lies below the centre of the first sub-block without a final
            // estimate (the interpolation below never reads past b_known, so a rounding of this bound only moves a
            // symbol into the next round)
            k_end = min((int)(((double)b_known + 0.5) * (double)kTimingBlock * P.inv_sps), k_lo + P.max_soft);
        }
        k_end = __builtin_amdgcn_readfirstlane(k_end);       // (uniform by construction: scalar loop control and store bases)
        k_begin = __builtin_amdgcn_readfirstlane(k_begin);
        const int ring_lo = max(0, base + kRrcTile - kRing), ring_hi = base + kRrcTile;
        const int ring_off = ring_lo % kRing;
        const int span4 = ring_hi - ring_lo - 4;
        const int b0_max = __builtin_amdgcn_readfirstlane(max(min(nb - 2, b_known - 1), 0));   // (>= 0: clamp_med3's contract)
        const int m_max = __builtin_amdgcn_readfirstlane(max(n - 3, 1));
        const float sps_f = (float)sps;
        constexpr int SU = PER / 4;   // symbols per thread in flight together
        // whole sample m and fraction mu of the instant t_k = (k + tau(k sps)) sps for the symbols kb0 + off + u * TSYM; SU
        // symbols side by side so that their LDS round trips (two timing estimates, then four ring samples) overlap.  The
        // nominal position k sps is a 64-bit fixed-point number with 40 fraction bits (3e-8 samples at the end of the
        // longest chunk): the thread's own share is a constant, the rest of the sum is uniform -- one 64-bit add per symbol
        // where the first version converted to and from fp64.
        auto instants = [&](uint64_t ub, int (&mm)[SU], float (&mu)[SU]) {   // ub = kb * sps40 (uniform, scalar registers)
            float tk[SU], ta[SU], tb2[SU], ff[SU];
            int mk[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const uint64_t pos = ptid + (ub + (uint64_t)u * (sps40 * TSYM));
                const uint32_t hi = (uint32_t)(pos >> 32), lo = (uint32_t)pos;
                mk[u] = (int)(hi >> 8);
                tk[u] = (float)__builtin_amdgcn_alignbit(hi, lo, 8) * 2.3283064365386963e-10f;   // top 32 fraction bits * 2^-32
                // piecewise-linear timing estimate between sub-block centres (both estimates final: b0 + 1 <= b_known):
                // u = (position - TB/2) / TB, b0 = floor(u) clamped, f = u - b0 clamped
                const int t = mk[u] - kTimingBlock / 2;
                const int b0 = clamp_med3<0>(t >> 8, b0_max);
                static_assert(kTimingBlock == 256, "shift");
                ff[u] = __builtin_amdgcn_fmed3f(((float)(t - (b0 << 8)) + tk[u]) * (1.f / (float)kTimingBlock), 0.f, 1.f);
                const float2 tp = tau[b0 & (kTauRing - 1)];   // (tau_b0, tau_{b0+1}: b0 + 1 <= b_known <= nb - 1, or nb == 1: its own)
                ta[u] = tp.x;
                tb2[u] = tp.y;
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const float tauk = ta[u] + ff[u] * (tb2[u] - ta[u]);
                const float ts = tk[u] + tauk * sps_f;      // t_k relative to the whole sample mk
                const float fl = floorf(ts);
                mu[u] = ts - fl;
                mm[u] = clamp_med3<1>(mk[u] + (int)fl, m_max);
            }
        };
        bool any_direct = false;
        uint64_t ub = (uint64_t)(uint32_t)k_begin * sps40;
        for (int kb = k_begin; kb < k_end; kb += TSYM * SU, ub += sps40 * (TSYM * SU)) {
            FarrowTaps f[SU];
            int mm[SU];
            float mu[SU];
            instants(ub, mm, mu);
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                f[u].mu = mu[u];
                // position of the symbol's first filter output in the ring's window; outside it -> direct path below
                const int q = mm[u] - 1 - ring_lo;
                const int k = kb + off + u * TSYM;
                any_direct |= (unsigned)q > (unsigned)span4 && k < k_end;
                const int p = clamp_med3<0>(q, span4) + ring_off;                  // < 2 kRing
                const int p0 = (int)min((unsigned)p, (unsigned)(p - kRing));       // p >= kRing ? p - kRing : p
                const float2 *yp = yring + rrc_slot(p0);
                f[u].ym1 = yp[0];
                f[u].y0 = yp[1];
                f[u].y1 = yp[2];
                f[u].y2 = yp[3];
            }
            float2 *so = sr + (kb - k_lo);   // (uniform base, the thread's own offset)
#pragma unroll
            for (int u = 0; u < SU; ++u)
                if (kb + off + u * TSYM < k_end) {
                    float2 sv = farrow_eval(f[u]);
                    if (FMT8) { sv.x *= TetraIn8<FMT8>::scale; sv.y *= TetraIn8<FMT8>::scale; }
                    so[off + u * TSYM] = sv;
                }
        }
        if (any_direct) {
            // the timing estimate has carried some instants out of the ring (more than 48 symbols from their nominal
            // positions): their four filter outputs again from the input.  Rare, rolled, and kept apart from the loop
            // above so that its loads never order that loop's registers; the explicit wait leaves nothing pending.
            for (int kb = k_begin; kb < k_end; kb += TSYM * SU) {
                int mm[SU];
                float mu[SU];
                instants((uint64_t)(uint32_t)kb * sps40, mm, mu);
#pragma unroll 1
                for (int u = 0; u < SU; ++u) {
                    // (selects, not mm[u] / mu[u]: a dynamically indexed private array is promoted to LDS, 2 KB per workgroup)
                    static_assert(SU == 2, "select");
                    const int m = u ? mm[1] : mm[0], k = kb + off + u * TSYM;
                    if ((m - 1 >= ring_lo && m + 2 < ring_hi) || k >= k_end) continue;
                    float a0x = 0.f, a0y = 0.f, a1x = 0.f, a1y = 0.f, a2x = 0.f, a2y = 0.f, a3x = 0.f, a3y = 0.f;
                    float2 q0 = make_float2(0.f, 0.f), q1 = q0, q2 = q0;   // sliding window x[idx - 3 .. idx - 1]
                    const int first = m - 1 - H2;
#pragma unroll 1
                    for (int k2 = -3; k2 < NT; ++k2) {
                        const int idx = first + k2 + 3;
                        float2 q3 = make_float2(0.f, 0.f);
                        if (idx >= 0 && idx < n) q3 = FMT8 ? TetraIn8<FMT8>::conv(xr8[idx]) : xr[idx];
                        if (k2 >= 0) {   // tap k2 multiplies x[first + k2 + e] for output e
                            const float h = P.taps[k2];
                            a0x = fmaf(h, q0.x, a0x); a0y = fmaf(h, q0.y, a0y);
                            a1x = fmaf(h, q1.x, a1x); a1y = fmaf(h, q1.y, a1y);
                            a2x = fmaf(h, q2.x, a2x); a2y = fmaf(h, q2.y, a2y);
                            a3x = fmaf(h, q3.x, a3x); a3y = fmaf(h, q3.y, a3y);
                        }
                        q0 = q1; q1 = q2; q2 = q3;
                    }
                    FarrowTaps f;
                    f.mu = u ? mu[1] : mu[0];
                    f.ym1 = make_float2(a0x, a0y);
                    f.y0 = make_float2(a1x, a1y);
                    f.y1 = make_float2(a2x, a2y);
                    f.y2 = make_float2(a3x, a3y);
                    float2 sv = farrow_eval(f);
                    if (FMT8) { sv.x *= TetraIn8<FMT8>::scale; sv.y *= TetraIn8<FMT8>::scale; }
                    sr[k - k_lo] = sv;
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
        }
        k_begin = max(k_begin, k_end);
    }
    __syncthreads();   // the carrier's soft symbols are visible to the whole workgroup
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef u32x2 __attribute__((aligned(1))) u32x2_a1;
    // Margin: smallest min(|re|,|im|) / max(|re|,|im|) (the angular distance to the nearest boundary is its atan), kept as the
    // pair (lo, hi) and compared by cross-multiplication -- one reciprocal per thread at the end instead of one per symbol.
    // The pair starts at (3e38, 1) ("no symbol yet": any symbol with hi > 0 replaces it); a symbol that is exactly zero
    // (margin 0 by the definition's atan2(0, 0)) never wins a strict comparison and is caught by the smallest hi instead;
    // a NaN symbol fails every comparison.
    float mlo = 3.0e38f, mhi = 1.f, hmin = 3.0e38f;
    {
        // ---- differential products d_i = s_i conj(s_{i-1}), the 4th-power carrier-offset estimate over them, then the quadrant
        // decisions.  A thread owns CH consecutive symbols of a chunk of CH * 256: four 16-byte loads (the carrier's soft
        // symbols come back from L2), the predecessor of its first symbol from the lane to its left (one wavefront shift), one
        // 8-byte store of its decisions.  The products of the first KEEP chunks (8192 symbols) stay in registers between the
        // two passes; longer chunks form theirs again.
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        typedef u32x2 __attribute__((aligned(1))) u32x2_a1;
        constexpr int CH = 8, CSYM = CH * kRrcThreads, KEEP = 8192 / CSYM;
        const int ms2 = P.max_soft - 2;   // (ns <= max_soft - 2: a pair that holds a symbol below ns is never clamped)
        // The 4th power of a differential product is the 8th power of the input's scale: symbols enter the products times a
        // power of two that brings the carrier's middle symbol to [0.5, 1) -- exact, so estimate, decisions and margin are what
        // they would be without it, and inputs anywhere in fp32's range (int16-scaled IQ, 1e-6-scaled IQ) neither overflow
        // nor flush to zero.
        if (ns > 0) {
            const float2 smid = sr[ns >> 1];
            const float a = fmaxf(fabsf(smid.x), fabsf(smid.y));
            int ex = 0;
            if (a > 0.f && a < 3.0e38f) (void)frexpf(a, &ex);
            sc = ldexpf(1.f, -ex);
        }
        // (TAIL: the carrier's last chunk of CSYM symbols -- clamped addresses, nothing beyond symbol ns - 1; every other chunk
        // lies inside [0, ns) and needs neither)
        auto products = [&](int c0, float2 (&d)[CH], auto tail_c) __attribute__((always_inline)) {
            constexpr bool TAIL = decltype(tail_c)::value;
            const int i0 = c0 + CH * tid;
            f32x4 v[CH / 2];
    #pragma unroll
            for (int j = 0; j < CH / 2; ++j) v[j] = *(const f32x4_a8 *)(sr + (TAIL ? min(i0 + 2 * j, ms2) : i0 + 2 * j));
            float2 pm = sr[max((TAIL ? min(i0, ms2) : i0) - 1, 0)];   // (used by lane 0 of a wavefront)
            pm.x *= sc;
            pm.y *= sc;
    #pragma unroll
            for (int j = 0; j < CH / 2; ++j) v[j] *= sc;
            // (copies first: __builtin_bit_cast of a vector-element lvalue reads element 0 with this compiler)
            const float lx = v[CH / 2 - 1].z, ly = v[CH / 2 - 1].w;
            float px = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, lx), 0x138, 0xf, 0xf, false));   // wave_shr:1
            float py = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ly), 0x138, 0xf, 0xf, false));
            if (lane == 0) {   // symbol 0 has no predecessor: its product is zero
                px = i0 == 0 ? 0.f : pm.x;
                py = i0 == 0 ? 0.f : pm.y;
            }
    #pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float cx = (u & 1) ? v[u >> 1].z : v[u >> 1].x, cy = (u & 1) ? v[u >> 1].w : v[u >> 1].y;
                d[u] = make_float2(cx * px + cy * py, cy * px - cx * py);
                px = cx;
                py = cy;
            }
            if (TAIL) {
    #pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (i0 + u >= ns) d[u] = make_float2(0.f, 0.f);
            }
        };
        auto products_chunk = [&](int c0, float2 (&d)[CH]) __attribute__((always_inline)) {
            if (c0 + CSYM > ns) products(c0, d, std::true_type{});
            else products(c0, d, std::false_type{});
        };
        // sum of d^4 = (p + 2 i q)^2 with p = Re d^2 = x^2 - y^2, q = Im d^2 / 2 = x y:  p^2 - 4 q^2 + 4 i p q, the three sums
        // kept apart (one multiply-add each per symbol) and combined once per thread
        auto power4 = [&](const float2 (&d)[CH]) __attribute__((always_inline)) {
    #pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float x = d[u].x, y = d[u].y;
                const float p4 = fmaf(x, x, -(y * y)), q4 = x * y;
                a_pp = fmaf(p4, p4, a_pp);
                a_qq = fmaf(q4, q4, a_qq);
                a_pq = fmaf(p4, q4, a_pq);
            }
        };
        float2 dk[KEEP][CH];
    #pragma unroll
        for (int c = 0; c < KEEP; ++c)
            if (c * CSYM < ns) {
                products_chunk(c * CSYM, dk[c]);
                power4(dk[c]);
            }
        for (int c0 = KEEP * CSYM; c0 < ns; c0 += CSYM) {
            float2 d[CH];
            products_chunk(c0, d);
            power4(d);
        }
        {
            // both sums through one exchange
            float a4r = wave_sum(fmaf(-4.f, a_qq, a_pp));
            float a4i = wave_sum(4.f * a_pq);
            constexpr int NW = kRrcThreads / 64;
            if (lane == 0) { sm[wv] = a4r; sm2[wv] = a4i; }
            __syncthreads();
            if (tid == 0) {
                float r = 0.f, q = 0.f;
                for (int i = 0; i < NW; ++i) { r += sm[i]; q += sm2[i]; }
                delta_s = (r == 0.f && q == 0.f) ? 0.f : atan2f(-q, -r) * 0.25f;
            }
            __syncthreads();
        }
        float rs, rc;
        __sincosf(-delta_s, &rs, &rc);   // |delta| <= pi/4
        // ---- quadrant decision of dd = d_k exp(-i delta): +pi/4 -> 0, +3pi/4 -> 1, -pi/4 -> 2, -3pi/4 -> 3, i.e. the dibit
        // (Im dd < 0, Re dd < 0) = the two sign bits (three integer instructions per symbol).
        // Margin: smallest min(|re|,|im|) / max(|re|,|im|) (the angular distance to the nearest boundary is its atan), kept as the
        // pair (lo, hi) and compared by cross-multiplication -- one reciprocal per thread at the end instead of one per symbol.
        // The pair starts at (3e38, 1) ("no symbol yet": any symbol with hi > 0 replaces it); a symbol that is exactly zero
        // (margin 0 by the definition's atan2(0, 0)) never wins a strict comparison and is caught by the smallest hi instead;
        // a NaN symbol fails every comparison.
        auto decide = [&](int c0, float2 (&d)[CH], auto tail_c) __attribute__((always_inline)) {
            constexpr bool TAIL = decltype(tail_c)::value;
            const int i0 = c0 + CH * tid;
            // (symbol 0: its slot repeats symbol 1, which leaves the minimum margin alone; its decision is not stored)
            if (c0 == 0 && i0 == 0) d[0] = d[1];
            uint32_t w[2] = {0u, 0u};
    #pragma unroll
            for (int u = 0; u < CH; ++u) {
                const float ddx = d[u].x * rc - d[u].y * rs, ddy = d[u].x * rs + d[u].y * rc;
                uint32_t h = __builtin_bit_cast(uint32_t, ddy) >> 31;
                h = __builtin_amdgcn_alignbit(h, __builtin_bit_cast(uint32_t, ddx), 31);   // (h << 1) | sign of Re
                w[u >> 2] |= h << (8 * (u & 3));
                float lo = fminf(fabsf(ddx), fabsf(ddy)), hi = fmaxf(fabsf(ddx), fabsf(ddy));
                if (TAIL) {   // symbols outside [1, ns) repeat the running pair: they never win the strict comparison, so a
                              // carrier without a single decision (ns <= 1) keeps the "no symbol" sentinel
                    const bool valid = i0 + u < ns && i0 + u >= 1;
                    lo = valid ? lo : mlo;
                    hi = valid ? hi : mhi;
                }
                hmin = fminf(hmin, hi);
                // lo / hi < mlo / mhi  <=>  lo * mhi < mlo * hi (all non-negative)
                const bool take = lo * mhi < mlo * hi;
                mlo = take ? lo : mlo;
                mhi = take ? hi : mhi;
            }
            if (!TAIL && i0 > 0) {
                *(u32x2_a1 *)(hr + i0 - 1) = u32x2{w[0], w[1]};
            } else {
    #pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (i0 + u >= 1 && i0 + u < ns) hr[i0 + u - 1] = (uint8_t)(w[u >> 2] >> (8 * (u & 3)));
            }
        };
        auto decide_chunk = [&](int c0, float2 (&d)[CH]) __attribute__((always_inline)) {
            if (c0 + CSYM > ns) decide(c0, d, std::true_type{});
            else decide(c0, d, std::false_type{});
        };
    #pragma unroll
        for (int c = 0; c < KEEP; ++c)
            if (c * CSYM < ns) decide_chunk(c * CSYM, dk[c]);
        for (int c0 = KEEP * CSYM; c0 < ns; c0 += CSYM) {
            float2 d[CH];
            products_chunk(c0, d);
            decide_chunk(c0, d);
        }
    }
    const float mratio = hmin == 0.f ? 0.f : mlo * __builtin_amdgcn_rcpf(mhi);   // (no decision, ns <= 1 or all NaN: 3e38)
    float margin = mratio <= 1.f ? atanf(mratio) : 3.4e38f;   // (a NaN ratio never replaces the running minimum)
    margin = block_min(margin, sm);
    if (tid == 0) {
        n_soft[row] = ns;
        if (timing_milli) timing_milli[row] = (int32_t)rintf(tau_mid_s * 1000.f);
        if (min_margin) min_margin[row] = (double)margin;
    }
}

}  // namespace tdm
