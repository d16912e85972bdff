// TETRA mode, Gardner variant (TDM_MODE_TETRA_GARDNER): the textbook receiver BASELINE.json's north_star names --
// RRC matched filter -> Gardner timing-error detector -> proportional-integral loop filter -> period-controlled cubic
// Farrow interpolator -> differential quadrant decisions -- on the device, defined by oracle/tetra_np.py demod_gardner.
//
// The loop is a nonlinear recurrence over a carrier's symbols (every symbol instant depends on the errors of all symbols
// before it), so it cannot be tiled over time like the feed-forward receiver of tetra_kernels.hpp: the parallel axis is
// the CARRIER.  Kernels:
//   k_tetra_gardner<NT>  (the default) one workgroup = THREE wavefronts for sixteen carriers: the LOOP wavefront, FOUR LANES
//                    PER CARRIER (lane = strobe x component), walks the symbols out of an LDS ring of matched-filter
//                    samples; two PRODUCER wavefronts run the NT-tap RRC matched filter for eight carriers each, a
//                    64-sample chunk at a time, from the raw input straight into that ring (the filter output never goes to
//                    HBM); hand-over once per block of 16 symbols through one workgroup barrier
//   k_tetra_mf       the matched filter alone, LDS-tiled sliding window, fp32, output to HBM (one workgroup per 2048
//                    outputs): the RRC stage as its own kernel (tdm_plan_rrc_filter) and the first of the three launches
//   k_tetra_gardner<0>   the loop alone, fed from k_tetra_mf's output in HBM (TDM_GARDNER_FUSED=0: three launches)
//   k_tetra_decide   differential products, 4th-power carrier-offset estimate, quadrant decisions, margin (one workgroup
//                    per carrier)
// It is the slower receiver by construction and exists because the north-star names it: tests compare it with the fp64
// definition, bench.py times it beside the feed-forward receiver.  Measured (MI355X, 4096 x 32 768 at 4 samples per symbol):
// fused filter + loop 1.28 ms, decisions 0.08 ms = 1.37 ms per batch (three launches: matched filter 0.37 ms at 5.8 TB/s, loop
// 1.37, decisions 0.10 = 1.85).  Fused filter + loop at 3 / 4 / 5 / 6 / 8 samples per symbol (25 ... 65 taps): 1.65 / 1.28 /
// 1.14 / 1.13 / 1.17 ms against 2.07 / 1.74 / 1.57 / 1.44 / 1.31 for filter and loop as two launches.  What the producers
// needed for that (a chunk costs them 8 NT multiply-adds per lane, and at 8 samples per symbol the loop uses up a chunk in
// 1.4 us): the next window requested during this chunk's arithmetic; taps from LDS as pairs (as scalar kernel arguments
// they spilled: two v_readlane and a wait state per multiply-add); the multiply-adds written out tap by tap (the compiler
// walks output by output: dependent, each behind a wait state, every tap first copied into a (h, h) pair); all window
// reads in flight together (k_tetra_mf's one-wait-per-read is hidden by six other workgroups there, by nobody here).
// The loop's time does not depend on the number of carriers up to 16 384
// (one loop wavefront per compute unit): it is 8190 symbols x the ~165 ns ONE symbol's chain of ~40 vector instructions takes in a
// wavefront that has its SIMD to itself (tools/harness/ubench_chain.hip: a lone wavefront issues one vector instruction
// per 3.4 ns, a dependent multiply-add takes 4.6 ns, an LDS round trip 23 ns) -- which is also why the producers are free:
// the loop wavefront uses a third of ONE of its compute unit's four SIMDs.  History of that number: one lane per carrier
// 2.92 ms (135 instructions a turn); four lanes per carrier 2.31; first symbol peeled, one-compare window test 2.18; the rare
// work (capacity, middle symbol, clamp) in a second copy of the block, cross-lane operands folded into the arithmetic (DPP),
// all four lanes store 1.83; straight-line turns behind ONE wavefront-uniform branch 1.76; ring moves without clamps and
// address arithmetic, 16-turn blocks 1.47; weights and taps as packed instructions 1.46 (they sat in the shadow of the ring
// reads); the next symbol's ring reads issued at the end of the turn (software pipeline) 1.37; fed by the producers (no ring
// moves of its own) 1.28; the strobe and the instant updated side by side in packed instructions 1.25.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "tetra_params.hpp"


namespace tdm {

#define TDM_MF_TPW 1   // tiles a workgroup walks, the next one's window in flight (measured: 1 -> 0.371 ms, 2 -> 0.377, 4 -> 0.390, 8 -> 0.424)
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
// FMT8: 0 = cf32 input (above); 1 = cu8, 2 = cs8 (north_star: "coalesced complex-int8/float loads"): the window arrives as
// 2-byte samples -- 10 instead of 16 bytes per sample through HBM -- and is converted where it is staged,
// cu8 as pyrtlsdr and the channeliser do (u * fl(1/127.5) - 1 in fp32), cs8 as s / 128; everything behind the staging is the
// cf32 kernel.  (2-byte loads, one per sample and thread: a row of bytes has no alignment to speak of, and at 2 of the
// kernel's 10 bytes per sample the load instructions are not what bounds it.)
template <int FMT8>
__device__ __forceinline__ float2 mf_convert8(uint32_t h)   // low 16 bits: I, Q
{
    if (FMT8 == 1) return make_float2((float)(h & 255u) * (1.f / 127.5f) - 1.f, (float)((h >> 8) & 255u) * (1.f / 127.5f) - 1.f);
    return make_float2((float)(int8_t)(h & 255u) * (1.f / 128.f), (float)(int8_t)((h >> 8) & 255u) * (1.f / 128.f));
}

template <int NT, int FMT8 = 0>
__global__ __launch_bounds__(kMfThreads) void k_tetra_mf(const void *__restrict__ x_, int64_t in_stride, const TetraParams P,
                                                         float2 *__restrict__ y, int64_t y_pitch)
{
    const float2 *x = (const float2 *)x_;
    constexpr int H = (NT - 1) / 2, W = kMfTile + NT - 1, WIN = kMfPer + NT - 1;   // WIN: samples under a thread's outputs
    constexpr int NLD = (W + kMfThreads - 1) / kMfThreads;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float2 xs[W + 2 * (W / 8) + 4];
    auto slot = [](int s) { return s + 2 * (s >> 3); };
    const int row = blockIdx.y, tid = threadIdx.x, n = P.n;
    const float2 *xr = x + (int64_t)row * in_stride;
    const uint16_t *xr8 = (const uint16_t *)x_ + (int64_t)row * in_stride;   // (FMT8: one 2-byte sample per element)
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
    uint32_t v8[NLD];   // (FMT8) the thread's samples of the window as they arrive
    auto fetch = [&](int base) {
        if (FMT8) {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int g = base - H + tid + k * kMfThreads;
                v8[k] = xr8[min(max(g, 0), n - 1)];          // (unconditional, clamped: all of a thread's loads in flight at once)
            }
            return;
        }
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
        if (FMT8) {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int i = tid + k * kMfThreads, g = base - H + i;
                if (i < W) xs[slot(i)] = (g >= 0 && g < n) ? mf_convert8<FMT8>(v8[k]) : make_float2(0.f, 0.f);
            }
            return;
        }
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
// FOUR LANES PER CARRIER: lane (s, c) of a carrier's quad forms component c (re / im) of strobe s (on time / mid-symbol) --
// its own four ring reads, its own Lagrange weights (packed, two at a time), its four taps -- and the detector's two sums over the
// quad travel by DPP (no LDS, no cross-lane latency beyond the instruction itself).  The loop's time is the length of ONE
// symbol's dependent chain times the symbols of a carrier, whatever the number of carriers (every wavefront has a SIMD to
// itself up to 16 384 carriers), so the lanes are spent on shortening that chain: one lane per carrier (both strobes, both
// components and eight ring reads in every lane: ~135 instructions a turn, 64 carriers and 131 KB of ring per wavefront)
// took 2.92 ms for 4096 x 32 768; this one 1.37 ms (the steps in between: head of this file).
constexpr int kGChunk = 64, kGChunks = 4;    // the loop works inside four resident chunks of 64 samples
constexpr int kGQuads = 16;                  // carriers per wavefront
// LDS slots per carrier: the ring (a power of two: index by mask) and its first three slots once more behind it, so that a
// lane's four taps are four CONSECUTIVE slots wherever the window starts (odd pitch: rows on different banks).  The loop fed
// from HBM keeps exactly the four chunks; the loop fed by the matched-filter wavefronts of its own workgroup (below) eight:
// four for the loop, four the producers may fill ahead.
template <bool FUSED> struct GardnerRing {
    static constexpr int chunks = FUSED ? 8 : kGChunks, slots = chunks * kGChunk, pitch = slots + 3;
};
constexpr int kGProducers = 2;               // matched-filter wavefronts of the fused kernel: eight carriers each
// TDM_GARDNER_PLACE=1: the fused kernel's workgroup has FOUR wavefronts -- one per SIMD -- of which one leaves at once; which
// SIMD's wavefront runs the loop depends on the parity of the workgroup's slot on its compute unit (HW_ID.TG_ID), so that the
// two workgroups of a compute unit put their loops on DIFFERENT SIMDs (0 and 2) and their producers together on the other
// two: a loop wavefront then has its SIMD to itself also when two workgroups share the unit
#define TDM_GARDNER_PLACE 1   // (measured: 4096 carriers 1.229 -> 1.204 ms, 8192 carriers 1.607 -> 1.418 ms)
constexpr int kGWaves = TDM_GARDNER_PLACE ? 4 : 1 + kGProducers;   // wavefronts a fused workgroup is launched with
constexpr int kGQuota = 2;                   // chunks a producer makes between two hand-overs (a block of 16 symbols uses sps / 4)
template <int NT> struct GardnerWindow {     // a producer's input window per carrier and chunk
    static constexpr int H = (NT - 1) / 2, W = kGChunk + NT - 1, pairs = W / 2;
    // (no pad slots, unlike k_tetra_mf's window: padding the rows did not change the producers' time -- they wait on
    //  arithmetic, not on LDS banks -- and without it TWO workgroups fit a compute unit's 160 KB up to 41 taps: 8192 carriers
    //  in the time of 4096)
    static constexpr int pitch = W + 4;
    static constexpr int slot(int s) { return s; }
    static constexpr int loads = (8 * pairs + 63) / 64;
};

struct GardnerConsts {
    float k1, k2;      // loop filter gains (oracle/tetra_np.py demod_gardner: Rice eq. C.61, detector gain 2.7, bn_t 0.01, zeta 0.7071)
                       // x 100: the kernel tracks 100 x the symbol power (one multiplication less per symbol)
};

template <int CTRL>
__device__ __forceinline__ float gardner_quad(float v)   // DPP quad_perm: lane i of a quad reads lane (CTRL >> 2 i) & 3
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int kQuadOtherComponent = 0xB1;    // [1, 0, 3, 2]
constexpr int kQuadMidStrobe = 0xEE;         // [2, 3, 2, 3]
constexpr int kQuadFirst = 0x00;             // [0, 0, 0, 0]
constexpr int kQuadSymbolPair = 0x44;        // [0, 1, 0, 1]
constexpr int kQuadOtherPair = 0x4E;         // [2, 3, 0, 1]

// NT = 0: the loop alone, one wavefront per workgroup, fed from the matched filter's output y in HBM (k_tetra_mf before it).
// NT > 0: FUSED -- the workgroup has two more wavefronts that run the NT-tap matched filter for the loop's sixteen carriers
// (eight each, a 64-sample chunk at a time, from the raw input x) straight into the LDS ring: the filter output never goes
// to HBM.  A lone wavefront leaves two thirds of its SIMD's issue slots and the other three SIMDs of its compute unit idle;
// the producers' ~350 instructions per chunk (33 taps) fit there many times over.  Hand-over at the loop's block boundaries through
// ONE workgroup barrier per block (no polling): the producers publish how many chunks are complete, the loop how far it has
// moved on; every wavefront passes the same barriers and leaves after the one at which `done` was set.
// FMT8 (fused form only; round 6): 0 = cf32 input, 1 = cu8, 2 = cs8 -- the producers fetch 2-byte samples and convert them where
// they write their window (mf_convert8); everything behind the window is the cf32 kernel.
template <int NT, int FMT8 = 0>
__global__ __launch_bounds__(NT > 0 ? 64 * kGWaves : 64) void k_tetra_gardner(const float2 *__restrict__ y, int64_t y_pitch, const TetraParams P,
                                                      const GardnerConsts G, int rows, float2 *__restrict__ soft,
                                                      int32_t *__restrict__ n_soft, int32_t *__restrict__ timing_milli, const GardnerSeg S)
{
    constexpr bool FUSED = NT > 0;
    static_assert(FUSED || FMT8 == 0, "the loop alone reads the matched filter's cf32 output");
    // input row of (virtual) carrier v: second halves start seg_off samples into their carrier's row
    auto row_in = [&](int v) {
        v = min(v, rows - 1);
        const int h = S.pieces > 0 ? v / S.rows_phys : 0;     // the piece
        return y + (int64_t)(v - h * S.rows_phys) * y_pitch + (int64_t)h * S.seg_step;
    };
    auto row_in8 = [&](int v) {                               // (FMT8: the same row as 2-byte samples)
        v = min(v, rows - 1);
        const int h = S.pieces > 0 ? v / S.rows_phys : 0;
        return (const uint16_t *)y + (int64_t)(v - h * S.rows_phys) * y_pitch + (int64_t)h * S.seg_step;
    };
    constexpr int kGRing = GardnerRing<FUSED>::slots, kGPitch = GardnerRing<FUSED>::pitch;
    __shared__ float2 ring[kGQuads * kGPitch];      // sample g of a carrier in slot g mod kGRing of its row (33 KB; fused: 66 KB)
    __shared__ __attribute__((aligned(16))) float2 xwin[FUSED ? kGProducers * 8 * GardnerWindow<FUSED ? NT : 1>::pitch : 1];
    __shared__ int sh_prod[kGProducers], sh_c0, sh_done;
    // the producers' taps, in LDS: as kernel arguments they are scalar registers, and a packed multiply-add wants its scalar
    // operand as an aligned PAIR -- 2 x 65 scalar registers do not exist, and the compiler's spills (two v_readlane and a
    // wait state per multiply-add) made a 65-tap chunk take 5.5 us instead of 2.  A broadcast 8-byte LDS read per two taps.
    __shared__ __attribute__((aligned(8))) float taps_s[FUSED ? kRrcMaxTaps + 2 : 2];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef f32x4 __attribute__((aligned(8))) f32x4_a8;   // (a clamped pair at the end of an odd-length row starts on an odd sample)
    const int lane = threadIdx.x & 63;
    if constexpr (FUSED) {
        for (int t = threadIdx.x; t < NT + 1; t += 64 * kGWaves) taps_s[t] = t < NT ? P.taps[t] : 0.f;
        // roles by SIMD (see kGWaves): 0 the loop, 1 and 2 the producers, 3 leaves
        __shared__ int sh_simd[4];
        const int hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID, all 32 bits
        const int simd = (hw >> 4) & 3, slot = (hw >> 16) & 1;
        if (lane == 0) sh_simd[threadIdx.x >> 6] = simd;
        __syncthreads();
        const bool spread = ((1 << sh_simd[0]) | (1 << sh_simd[1]) | (1 << sh_simd[2]) | (1 << sh_simd[3])) == 15;
        const int widx = (int)threadIdx.x >> 6;
        const int rel = spread ? ((simd - 2 * slot) & 3) : widx;            // 0: the loop's SIMD, 2: the one that leaves
        const int wave = __builtin_amdgcn_readfirstlane(rel == 0 ? 0 : (rel == 1 ? 1 : (rel == 3 ? 2 : 3)));
        if (wave == 3) return;
        if (wave > 0) {
            // ---- a producer: carriers 8 (wave - 1) .. + 7 of the workgroup; lane = (carrier j, group of eight outputs gI)
            typedef GardnerWindow<FUSED ? NT : 1> GW;
            const int n = P.n, j = lane >> 3, gI = lane & 7, car0 = 8 * (wave - 1);
            float2 *xw = xwin + (wave - 1) * 8 * GW::pitch;
            const float2 *xrow[GW::loads];              // this lane's pair of every load, in chunk 0's window
            const uint16_t *xrow8[GW::loads];           // (FMT8) the carrier's sample 0
            int xcar[GW::loads], xpr[GW::loads];
#pragma unroll
            for (int k = 0; k < GW::loads; ++k) {
                const int f = min(lane + 64 * k, 8 * GW::pairs - 1);
                xcar[k] = f / GW::pairs;
                xpr[k] = f - xcar[k] * GW::pairs;
                xrow[k] = row_in((int)blockIdx.x * kGQuads + car0 + xcar[k]) + 2 * xpr[k] - GW::H;
                xrow8[k] = row_in8((int)blockIdx.x * kGQuads + car0 + xcar[k]);
            }
            float2 *myring = ring + (car0 + j) * kGPitch;
            // the window of the NEXT chunk is requested while this chunk's arithmetic runs (a chunk's 2 us of load latency
            // would otherwise be paid 512 times in a row: the producers, not the loop, set the pace above 4 samples per symbol)
            f32x4 pf[GW::loads];
            uint32_t pf8[GW::loads];   // (FMT8) a pair of samples as it arrives
            auto inside = [&](int cn) { return kGChunk * cn - GW::H >= 0 && kGChunk * cn - GW::H + GW::W <= n; };
            auto issue = [&](int cn) {
                if (FMT8) {
                    // two 2-byte loads per pair from clamped positions, every chunk alike (masked where they are written)
#pragma unroll
                    for (int k = 0; k < GW::loads; ++k) {
                        const int ga = kGChunk * cn - GW::H + 2 * xpr[k];
                        const uint32_t a = xrow8[k][min(max(ga, 0), n - 1)], b = xrow8[k][min(max(ga + 1, 0), n - 1)];
                        pf8[k] = a | (b << 16);
                    }
                    return;
                }
                if (inside(cn)) {
#pragma unroll
                    for (int k = 0; k < GW::loads; ++k) pf[k] = __builtin_nontemporal_load((const f32x4_a8 *)(xrow[k] + kGChunk * cn));
                }
            };
            issue(0);
            auto produce = [&](int cn) {
                const int g0 = kGChunk * cn - GW::H;            // the window's first sample
                if (kGChunk * cn >= n) {                        // past the row: zeros
#pragma unroll
                    for (int o = 0; o < 8; ++o) myring[((kGChunk * cn + 8 * gI) & (kGRing - 1)) + o] = make_float2(0.f, 0.f);
                } else {
                    // window: HBM -> registers -> this wavefront's LDS rows (nobody else reads them: no barrier, only the wait)
                    if (FMT8) {
#pragma unroll
                        for (int k = 0; k < GW::loads; ++k) {
                            const int ga = g0 + 2 * xpr[k];
                            if (k < GW::loads - 1 || lane + 64 * k < 8 * GW::pairs) {
                                float2 *d = xw + xcar[k] * GW::pitch + GW::slot(2 * xpr[k]);
                                d[0] = (ga >= 0 && ga < n) ? mf_convert8<FMT8 ? FMT8 : 1>(pf8[k]) : make_float2(0.f, 0.f);
                                d[1] = (ga + 1 >= 0 && ga + 1 < n) ? mf_convert8<FMT8 ? FMT8 : 1>(pf8[k] >> 16) : make_float2(0.f, 0.f);
                            }
                        }
                    } else if (inside(cn)) {
#pragma unroll
                        for (int k = 0; k < GW::loads; ++k)
                            if (k < GW::loads - 1 || lane + 64 * k < 8 * GW::pairs) *(f32x4 *)(xw + xcar[k] * GW::pitch + GW::slot(2 * xpr[k])) = pf[k];
                    } else {
#pragma unroll
                        for (int k = 0; k < GW::loads; ++k) {
                            const int ga = g0 + 2 * xpr[k];
                            const float2 *rowp = xrow[k] - (2 * xpr[k] - GW::H);     // the carrier's sample 0
                            const float2 a = rowp[min(max(ga, 0), n - 1)], b = rowp[min(max(ga + 1, 0), n - 1)];
                            if (k < GW::loads - 1 || lane + 64 * k < 8 * GW::pairs) {
                                float2 *d = xw + xcar[k] * GW::pitch + GW::slot(2 * xpr[k]);
                                d[0] = (ga >= 0 && ga < n) ? a : make_float2(0.f, 0.f);
                                d[1] = (ga + 1 >= 0 && ga + 1 < n) ? b : make_float2(0.f, 0.f);
                            }
                        }
                    }
                    issue(cn + 1);
                    // (the same wavefront reads what it wrote: LDS operations of a wavefront are served in order)
                    // eight consecutive outputs from the 8 + NT - 1 samples under them (as k_tetra_mf), the taps as pairs from LDS
                    f32x2 w[8 + NT - 1];
                    const float2 *pw = xw + j * GW::pitch + GW::slot(8 * gI);
#pragma unroll
                    for (int i = 0; i < 8 + NT - 1; ++i) w[i] = *(const f32x2 *)(pw + GW::slot(i));   // (all reads in flight together:
                    f32x2 acc[8];                                                                         //  a lone wavefront has nobody to hide a wait per read)
#pragma unroll
                    for (int o = 0; o < 8; ++o) acc[o] = f32x2{0.f, 0.f};
                    // Tap by tap, eight INDEPENDENT packed multiply-adds per tap, the tap taken from its half of the pair by the
                    // instruction's operand select -- written out: left to itself the compiler walks output by output (a chain of
                    // NT dependent packed multiply-adds, each behind a wait state) and copies every tap into a (h, h) register
                    // pair first: 5.9 ns per multiply-add instead of under 3.
                    // (the asm statements keep their place, so the read of the NEXT pair of taps stands in front of this pair's
                    //  sixteen multiply-adds: its LDS round trip passes behind them)
                    f32x2 hp = *(const f32x2 *)taps_s;                     // taps t, t + 1 (a zero behind the last)
#pragma unroll
                    for (int t = 0; t < NT; t += 2) {
                        f32x2 hn = hp;
                        if (t + 2 < NT) hn = *(const f32x2 *)(taps_s + t + 2);
#pragma unroll
                        for (int o = 0; o < 8; ++o)
                            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[o]) : "v"(w[o + t]), "v"(hp));
                        if (t + 1 < NT) {
#pragma unroll
                            for (int o = 0; o < 8; ++o)
                                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[o]) : "v"(w[o + t + 1]), "v"(hp));
                        }
                        hp = hn;
                    }
                    const int gs = kGChunk * cn + 8 * gI;
                    float2 *d = myring + (gs & (kGRing - 1));
#pragma unroll
                    for (int o = 0; o < 8; ++o) {
                        const float2 v = gs + o < n ? make_float2(acc[o].x, acc[o].y) : make_float2(0.f, 0.f);
                        d[o] = v;
                        if (o < 3 && (gs & (kGRing - 1)) == 0) d[kGRing + o] = v;    // slots 0..2 once more behind the ring
                    }
                }
            };
            int next = 0;
            for (; next < GardnerRing<true>::chunks; ++next) produce(next);
            if (lane == 0) sh_prod[wave - 1] = next;
            __syncthreads();                                    // the first hand-over: eight chunks
            for (;;) {
                if (*(volatile int *)&sh_done) break;
                const int c0_seen = *(volatile int *)&sh_c0;    // (possibly the value of the hand-over before: it only grows)
                for (int made = 0; made < kGQuota && next <= c0_seen + GardnerRing<true>::chunks - 1; ++made, ++next) produce(next);
                if (lane == 0) sh_prod[wave - 1] = next;
                __syncthreads();
            }
            return;
        }
    }
    const int quad = lane >> 2, s = (lane >> 1) & 1, c = lane & 1;
    const int row = min((int)blockIdx.x * kGQuads + quad, rows - 1);
    const bool mine = (int)blockIdx.x * kGQuads + quad < rows;
    const int n = P.n;
    const double sps = P.sps;
    const int back = (int)sps + 4;                 // samples behind floor(t) a strobe may need (mid-symbol strobe + interpolator)
    const float *my = (const float *)(ring + quad * kGPitch) + c;      // component c of slot i: my[2 i]
    // soft symbols: the workgroup's rows from a scalar base, a lane's position in them as a 32-bit byte offset that also
    // counts the carrier's symbols (k = (off - off0) / 8)
    // (pieces of a chunk, GardnerSeg::soft_a: a wavefront of first pieces writes into the caller's rows, one of later
    //  halves into the temporary, which then holds the second halves only)
    const bool wg_a = S.pieces > 0 && S.soft_a && (int)blockIdx.x * kGQuads < S.rows_phys;
    const int wg_row0 = (int)blockIdx.x * kGQuads - ((S.pieces > 0 && S.soft_a && !wg_a) ? S.rows_phys : 0);
    const uint32_t soft_pitch = wg_a ? (uint32_t)S.pitch_a : (uint32_t)P.max_soft;
    char *const wg_soft = (char *)((wg_a ? S.soft_a : soft) + (int64_t)wg_row0 * soft_pitch);
    const uint32_t off0 = (uint32_t)quad * soft_pitch * 8u + 4u * (uint32_t)c;
    uint32_t off = off0;
    // ---- cooperative chunk moves: chunk cn = samples [64 cn, 64 cn + 64) of every carrier of the wavefront.  One 16-byte load
    // fetches two samples; lanes 0..31 serve carrier 2 q, lanes 32..63 carrier 2 q + 1 (rows are 16-byte aligned: even pitch)
    f32x4 pf[kGQuads / 2];
    const int half = lane >> 5, l32 = lane & 31;
    const float2 *src[kGQuads / 2];                // this lane's pair in chunk 0 of the carriers it serves
    float2 *dst0[kGQuads / 2];                     // ... and its slots in their ring rows
#pragma unroll
    for (int q = 0; q < kGQuads / 2; ++q) {
        src[q] = row_in((int)blockIdx.x * kGQuads + 2 * q + half) + 2 * l32;
        dst0[q] = ring + (2 * q + half) * kGPitch + 2 * l32;
    }
    // (a chunk that ends inside the row -- all but the last one or two -- moves without clamps and masks: a ring move is
    //  paid by the sixteen symbols a chunk holds)
    auto request = [&](int cn) {
        if (kGChunk * cn + kGChunk <= n) {
#pragma unroll
            for (int q = 0; q < kGQuads / 2; ++q) pf[q] = *(const f32x4_a8 *)(src[q] + kGChunk * cn);
        } else {
            const int g = kGChunk * cn + 2 * l32;
            const int gg = min(g, max(n - 2, 0)) - 2 * l32;            // (clamped address; masked when it lands)
#pragma unroll
            for (int q = 0; q < kGQuads / 2; ++q) pf[q] = *(const f32x4_a8 *)(src[q] + gg);
        }
    };
    auto land = [&](int cn) {
        const int g = kGChunk * cn + 2 * l32, ring0 = (kGChunk * cn) & (kGRing - 1);
        const bool inside = kGChunk * cn + kGChunk <= n;
#pragma unroll
        for (int q = 0; q < kGQuads / 2; ++q) {
            float2 a = make_float2(pf[q].x, pf[q].y), b = make_float2(pf[q].z, pf[q].w);
            if (!inside) {
                if (g >= n) a = make_float2(0.f, 0.f);
                if (g + 1 >= n) b = make_float2(0.f, 0.f);
                if (g == n - 1 && n >= 2) a = make_float2(pf[q].z, pf[q].w);   // (the clamped pair ends at n - 1: its second half is sample n - 1)
            }
            float2 *dst = dst0[q] + ring0;
            dst[0] = a;
            dst[1] = b;
            if (ring0 == 0) {                                          // slots 0..2 once more behind the ring
                if (l32 == 0) { dst[kGRing] = a; dst[kGRing + 1] = b; }
                if (l32 == 1) dst[kGRing] = a;
            }
        }
    };
    int c0 = 0;                                     // oldest resident chunk: chunks c0 .. c0 + kGChunks - 1 are in the ring
    if constexpr (!FUSED) {
#pragma unroll 1
        for (int cn = 0; cn < kGChunks; ++cn) { request(cn); land(cn); }
        request(kGChunks);                          // in flight while the first symbols are formed
        __syncthreads();
    }
    // ---- per-carrier loop state (oracle/tetra_np.py demod_gardner), the same in the four lanes of a quad
    // the symbol instant t = m + mu, whole samples and a fraction in [0, 1): integer and fp32 arithmetic only in the loop's
    // dependent chain (an fp32 t would be good to 4e-3 samples at the end of a 32 768-sample chunk, an fp64 t puts a dozen
    // long-latency instructions into every turn)
    const float sps_f = (float)sps;
    const float half_lane = s ? 0.5f * sps_f : 0.f;  // lanes s = 0: the "mid-symbol" formulas below yield the symbol instant itself
    const float gain_t = sps_f * G.k1;              // period correction in samples per unit of error
    const float k2_v = G.k2;
    int m = 1 + (int)floor(sps);
    float mu = (float)(sps - floor(sps));
    float integ = 0.f;
    float q = 100.f;                                // running symbol power x 100 (the factor rides on the gains: GardnerConsts)
    float prev = 0.f;                               // lanes s = 0: component c of the previous symbol
    int m_mid = 0;
    float mu_mid = 0.f;                             // the instant of the chunk's middle symbol
    const int m_end = n - 3;                        // t <= n - 3  <=>  m < n - 3 or (m == n - 3 and mu == 0)
    // (pieces of a chunk, GardnerSeg: the first symbol at or behind the piece's incoming and its outgoing seam -- index and instant)
    const int piece = S.pieces > 0 ? row / S.rows_phys : 0;
    const int k_mid = S.pieces > 0 ? (piece == S.piece_mid ? S.k_mid : 0x7fffffff) : (int)(0.5 * (double)n / sps);
    const int seam_in = (S.pieces > 0 && piece > 0) ? S.seam_in : -0x7fffffff;
    const int seam_out = (S.pieces > 0 && piece < S.pieces - 1) ? S.seam_out : 0x7fffffff;
    int k_in_rec = seam_in < 0 ? 0 : -1, k_out_rec = -1;
    float t_in_rec = 0.f, t_out_rec = 0.f;
    // a carrier takes its strobes in a turn when m < hi_v: see the block loop
    int hi_v = -0x7fffffff;
    // Software pipeline over the symbols: the strobe positions of the NEXT symbol are known as soon as this symbol's error is
    // (both follow from the instant of this symbol, the loop filter's state and e), so a turn ends by issuing the next
    // turn's ring reads -- their round trip passes behind the store, the state update and the turn's branch, and behind
    // the next turn's weights.  aim: whole part and fraction of a lane's strobe at m0 + x samples, and the reads.
    int pm = 1;                                     // the lane's strobe of the coming symbol: taps in slots pm - 1 .. pm + 2,
    float pu = 0.f;                                 // fraction pu
    f32x2 py01 = {0.f, 0.f}, py23 = {0.f, 0.f};     // the taps, on their way
    auto fetch = [&]() {
        const float *p = my + 2 * ((pm - 1) & (kGRing - 1));
        py01 = f32x2{p[0], p[2]};
        py23 = f32x2{p[4], p[6]};
    };
    auto aim = [&](auto slow, int m0, float x) {
        constexpr bool SLOW = decltype(slow)::value;
        const float fx = floorf(x);
        const int m2 = m0 + (int)fx;
        // (the mid-symbol strobe is not taken before sample 1: the first chunk's business)
        pm = SLOW ? max(m2, 1) : m2;
        pu = (!SLOW || m2 >= 1) ? x - fx : 0.f;
        fetch();
    };
    const float c_int = half_lane - sps_f;          // d(strobe position) / d(integrator)
    const float c_err = c_int * G.k2 - gain_t;      // d(strobe position) / d(error)
    const float c_err_t = -sps_f * G.k2 - gain_t;   // the same for the symbol instant itself (lanes s = 0: c_err)
    // {the lane's strobe, the symbol instant} as pairs: position relative to m = mu + k_base + k_int integ + k_err e
    const f32x2 k_base = {sps_f - half_lane, sps_f}, k_int = {c_int, -sps_f}, k_err = {c_err, c_err_t};
    // One symbol of one carrier (the quad's four lanes together).  FIRST: no predecessor yet -- no error, nominal period.
    // SLOW: the work only a few turns of a carrier need -- the mid strobe held at sample 1 (first chunk), the middle symbol's
    // instant, the capacity of the output row -- compiled into a second copy of the block that runs when some carrier of
    // the wavefront is near one of them.
    // The symbol goes to position off + 8 T of the row (T: the turn of an unrolled run; `off` moves on behind it).
    auto symbol = [&](auto first, auto slow, const int T) {
        constexpr bool FIRST = decltype(first)::value, SLOW = decltype(slow)::value;
        // cubic Lagrange interpolation (the definition's _farrow1: samples at -1, 0, 1, 2 around the whole part of the
        // instant; in Horner form it is the same polynomial) as Lagrange weights:
        // w-1 = -u(u-1)(u-2)/6, w0 = (u+1)(u-1)(u-2)/2, w1 = -(u+1)u(u-2)/2, w2 = (u+1)u(u-1)/6, two at a time -- a packed
        // instruction costs a lone wavefront the same issue slot as a plain one (ubench_chain) --: 6 instructions, and the four
        // taps as a packed product, a packed multiply-add and one addition
        const f32x2 uu = {pu, pu};
        const f32x2 A = uu + f32x2{0.f, 1.f};                  // {u, u + 1}
        const f32x2 Bn = f32x2{1.f, 2.f} - uu;                 // {1 - u, 2 - u}   (signs arranged so that no half needs a negation of its own)
        const f32x2 S = (A * Bn) * f32x2{-1.f / 6.f, 0.5f};    // {u (u-1) / 6, -(u+1)(u-2) / 2}
        const f32x2 W21 = S * f32x2{A.y, A.x};                 // {w2, w1}
        const f32x2 Wm10 = S * f32x2{Bn.y, Bn.x};              // {w-1, w0}
        const f32x2 acc = __builtin_elementwise_fma(py23, f32x2{W21.y, W21.x}, py01 * Wm10);
        const float val = acc.x + acc.y;                       // s = 0: the symbol's component c; s = 1: the mid strobe's
        // where the next symbol's strobes lie, up to the error's share: this symbol's instant + sps (1 - integ) for the
        // symbol, half a period (0.5 sps (1 - integ)) earlier for the mid strobe (half_lane: 0 in lanes s = 0)
        // (the lane's strobe and the instant side by side in packed instructions: {x, un})
        f32x2 xu = __builtin_elementwise_fma(k_int, f32x2{integ, integ}, f32x2{mu, mu} + k_base);
        if (!FIRST) {
            // detector e = Re{(s_k - s_k-1) conj(s_k-1/2)} / (running power), right in lane (0, re) of the quad and broadcast from
            // there; loop filter and the next instant in every lane (the cross-lane operands ride on the instructions: DPP)
#pragma clang fp contract(off)   // (the cross-lane operands below stay operands of the additions and products: DPP folds into them)
            const float t = gardner_quad<kQuadMidStrobe>(val) * (val - prev);
            const float ee = gardner_quad<kQuadOtherComponent>(t) + t;
            const float v2 = val * val;
            q = fmaf(0.99f, q, gardner_quad<kQuadOtherComponent>(v2) + v2);
            const float e = gardner_quad<kQuadFirst>(ee * __builtin_amdgcn_rcpf(fmaxf(q, 1e-10f)));
            // t + sps (1 - (k1 e + integ')), integ' = integ + k2 e: a late strobe makes e positive and shortens the period
            xu = __builtin_elementwise_fma(k_err, f32x2{e, e}, xu);
            integ = fmaf(k2_v, e, integ);
        }
        const f32x2 fl = {floorf(xu.x), floorf(xu.y)};
        const f32x2 fr = xu - fl;                              // {the strobe's fraction, the instant's}
        {                                                      // the next symbol's reads leave here
            const int m2 = m + (int)fl.x;
            // (the mid-symbol strobe is not taken before sample 1: the first chunk's business)
            pm = SLOW ? max(m2, 1) : m2;
            pu = (!SLOW || m2 >= 1) ? fr.x : 0.f;
            fetch();
        }
        // (all four lanes store: lanes s = 1 their partner's value to their partner's address -- cheaper than masking them out)
        *(float *)(wg_soft + off + 8 * T) = gardner_quad<kQuadSymbolPair>(val);
        if (SLOW) {
            const int k = (int)((off - off0) >> 3) + T;
            if (k == k_mid) { m_mid = m; mu_mid = mu; }
            if (k_in_rec < 0 && m >= seam_in) { k_in_rec = k; t_in_rec = (float)(m - seam_in) + mu; }
            if (k_out_rec < 0 && m >= seam_out) { k_out_rec = k; t_out_rec = (float)(m - seam_out) + mu; }
            if (k + 1 >= P.max_soft) hi_v = -0x7fffffff;     // the row is full
        }
        prev = val;
        m += (int)fl.y;
        mu = fr.y;
    };
    auto in_chunk = [&](int mm, float uu) { return mm < m_end || (mm == m_end && uu == 0.f); };
    bool active = mine && in_chunk(m, mu) && P.max_soft > 0;
    bool done = !__any(active);
    if constexpr (FUSED) {
        if (lane == 0) { sh_c0 = 0; sh_done = done ? 1 : 0; }
        __syncthreads();                            // the first hand-over: the producers have made the ring's eight chunks
    }
    int produced = GardnerRing<true>::chunks;
    if constexpr (FUSED) {
        // The pieces of a chunk behind the first (GardnerSeg) start next to the eye: their first instant is the one in
        // [1 + sps, 1 + 2 sps) at which the square-law estimate over the ring's 512 filter outputs puts a symbol --
        // tau = -arg(sum_n |y_n|^2 exp(-2 pi i n / sps)) sps / (2 pi), oracle/tetra_np.py _gardner_loop(ff_init) -- instead of
        // wherever the piece's first sample happens to lie: half a symbol off the eye the detector's error vanishes too, and a
        // loop started there can sit for hundreds of symbols before it pulls in (seen in 1 of ~30 pieces).  A quad's four
        // lanes take every fourth sample each.
        // (the first loop of a chunk only by the plan option gardner_ff_start, and like the definition only when the chunk fills the ring)
        const bool ff_here = (piece > 0 || S.ff_first != 0) && n >= kGRing;
        if (__any(ff_here)) {
            constexpr int H = (NT - 1) / 2;
            const int l4 = lane & 3;
            const float w = -6.28318530717958648f / sps_f;
            float sr, cr, ss, cs;
            sincosf(w * (float)(H + l4), &sr, &cr);
            sincosf(w * 4.f, &ss, &cs);
            const float2 *rp = ring + quad * kGPitch;
            float ar = 0.f, ai = 0.f;
#pragma unroll 4
            for (int nn = H + l4; nn < kGRing; nn += 4) {
                const float2 v = rp[nn];
                const float pw = fmaf(v.x, v.x, v.y * v.y);
                ar = fmaf(pw, cr, ar);
                ai = fmaf(pw, sr, ai);
                const float cn = fmaf(cr, cs, -(sr * ss));
                sr = fmaf(sr, cs, cr * ss);
                cr = cn;
            }
            ar += gardner_quad<kQuadOtherComponent>(ar);
            ai += gardner_quad<kQuadOtherComponent>(ai);
            ar += gardner_quad<kQuadOtherPair>(ar);
            ai += gardner_quad<kQuadOtherPair>(ai);
            const float tau = -atan2f(ai, ar) * sps_f * 0.159154943091895336f;
            const float base = 1.f + sps_f;
            float dd = fmodf(tau - base, sps_f);
            if (dd < 0.f) dd += sps_f;
            const float t0 = base + dd;
            if (ff_here) {
                m = (int)floorf(t0);
                mu = t0 - floorf(t0);
            }
        }
    }
    aim(std::true_type{}, m, mu - half_lane);       // the first symbol's strobes (its samples lie in the first chunks: m < 2 + 2 sps)
    if (active) {
        symbol(std::true_type{}, std::true_type{}, 0);
        off += 8;
    }
    // Turns come in blocks of kGBlock with nothing wavefront-wide -- and nothing that only changes slowly -- inside: whether a
    // carrier is still inside its chunk and its row, whether the ring can move on and whether anybody is still active is
    // settled between blocks (a carrier outside the resident chunks simply waits out the turn).
    constexpr int kGBlock = 16;
    const int max_blocks = (4 * P.max_soft + 64) / kGBlock;   // (bounded whatever the input: every turn advances the slowest active carrier)
    for (int blk = 0; !done; ++blk) {
        const int k = (int)((off - off0) >> 3);
        const bool corner = m == m_end && mu == 0.f;          // t = n - 3 exactly: the last instant inside the chunk
        active = mine && (m < m_end || corner) && k < P.max_soft;
        // a carrier takes its strobes when the samples both of them can touch lie in the resident chunks -- m - back >= 64 c0
        // (nothing lies before sample 0; settled here for the whole block: m does not go back in a loop that works) and
        // m + 2 < 64 (c0 + 4) -- and m is inside the chunk: per turn ONE comparison, m < hi_v; one that has run ahead of the
        // wavefront's slowest carrier by more than three chunks waits for the ring to move on
        hi_v = (active && (c0 == 0 || m - back >= kGChunk * c0)) ? min(kGChunk * (c0 + kGChunks) - 2, m_end + (corner ? 1 : 0)) : -0x7fffffff;
        const bool near = (k <= k_mid && k_mid < k + kGBlock) || k + kGBlock > P.max_soft ||
                          (k_in_rec < 0 && m + kGBlock * back >= seam_in) ||
                          (k_out_rec < 0 && m + kGBlock * back >= seam_out);     // (the block that may cross a seam)
        // The block's turns run with the active carriers' lanes enabled and NO per-lane decision while every one of them can
        // take its strobes (one wavefront-uniform branch per turn: carriers of a wavefront move in step unless their clocks
        // differ by more than three chunks); what is left of the block when one cannot is done lane by lane.
        fetch();                                             // (what was read ahead before a ring move, or by a carrier that waited, is read again)
        int turn = 0;
        const int first_active = __ffsll((unsigned long long)__ballot(active)) - 1;
        auto block = [&](auto slow) {
            if (active) {
#pragma unroll
                for (int t = 0; t < kGBlock; ++t) {
                    if (!__all(m < hi_v)) break;
                    symbol(std::false_type{}, slow, t);
                    ++turn;
                }
                off += 8 * turn;
            }
            turn = __builtin_amdgcn_readlane(turn, first_active);   // (lanes outside the branch hold 0: the count of a lane inside)
#pragma unroll 1
            for (; turn < kGBlock; ++turn)
                if (m < hi_v) {                              // (the same in a quad's four lanes)
                    symbol(std::false_type{}, slow, 0);
                    off += 8;
                }
        };
        if (c0 == 0 || __any(active && near)) block(std::true_type{});
        else block(std::false_type{});
        // the ring moves on while no active carrier needs its oldest chunk any more (wavefront-uniform votes)
        active = mine && in_chunk(m, mu) && (int)((off - off0) >> 3) < P.max_soft;
        done = !__any(active) || blk + 1 >= max_blocks;
        if constexpr (FUSED) {
            // hand-over: ONE barrier per block with the producers; behind it their counts of complete chunks are valid, and
            // `done`, set in front of it, is what every wavefront of the workgroup leaves on
            if (lane == 0 && done) sh_done = 1;
            __syncthreads();
            if (!done) {
                produced = min(*(volatile int *)&sh_prod[0], *(volatile int *)&sh_prod[1]);
#pragma unroll 1
                for (int hop = 0; hop < 4; ++hop) {
                    const bool can_drop = !active || (m - back) >= kGChunk * (c0 + 1);
                    if (!__all(can_drop) || produced < c0 + kGChunks + 1) break;
                    ++c0;                               // (chunk c0 + 4 is complete: the window moves on)
                }
                if (lane == 0) sh_c0 = c0;
            }
        } else {
#pragma unroll 1
            for (int hop = 0; hop < 4 && !done; ++hop) {
                const bool can_drop = !active || (m - back) >= kGChunk * (c0 + 1);
                if (!__all(can_drop)) break;
                __syncthreads();                         // (one wavefront: orders the ring reads above against the writes below)
                land(c0 + kGChunks);                     // into the slots chunk c0 held
                ++c0;
                request(c0 + kGChunks);
                __syncthreads();
            }
        }
    }
    const int k = (int)((off - off0) >> 3);
    if (mine && (lane & 3) == 0) {
        n_soft[row] = k;
        if (timing_milli) {
            const double u = ((double)piece * (double)S.seg_step + (double)m_mid + (double)mu_mid) / sps;
            timing_milli[row] = (int32_t)rint((u - rint(u)) * 1000.0);
        }
        if (S.pieces > 0) {
            S.k_in[row] = k_in_rec < 0 ? k : k_in_rec;
            S.t_in[row] = t_in_rec;
            S.k_out[row] = k_out_rec < 0 ? k : k_out_rec;
            S.t_out[row] = t_out_rec;
        }
    }
}

// ---- pieces of a chunk (GardnerSeg): the join.  Piece 0's symbols before its outgoing seam already lie in the carrier's row;
// every further piece's follow from the symbol that IS its predecessor's first one at or behind their seam (the recorded
// instants say which: they differ by a whole number of symbol periods, 0 unless the two loops place a symbol on different
// sides of the seam) up to its own outgoing seam.  A bulk copy of its own ahead of k_tetra_decide (grid: tiles of kJoinTile
// symbols x carriers x pieces - 1), 0.05 ms for 4096 carriers in two pieces.  Measured alternatives for two pieces, join +
// decisions together (separate launches: 0.121 ms): the copy at the head of a deciding workgroup 0.157, its stores among that
// workgroup's loads 0.170 (every wait for a load becomes a wait for the stores' acknowledgements as well: one counter), behind
// its decisions 0.220 (a second latency chain), as workgroups of their own in the deciding launch, which then reads the
// pieces where they lie, 0.152.
struct GardnerJoin {
    const float2 *b;          // [(pieces - 1) rows][cap_b] the symbols of pieces 1..
    int32_t cap_b;
    const int32_t *n_v;       // [pieces rows] the pieces' symbol counts
    const int32_t *timing_v;  // [pieces rows]
    int32_t *timing_milli;    // [rows] or null
    float sps;
};
constexpr int kJoinPer = 8, kJoinTile = 256 * kJoinPer;
__global__ __launch_bounds__(256) void k_tetra_gardner_join(float2 *__restrict__ soft, int max_soft, int32_t *__restrict__ n_soft,
                                                            const GardnerSeg S, const GardnerJoin J)
{
    const int row = blockIdx.y, tid = threadIdx.x, mine = (int)blockIdx.z + 1;
    const int R = S.rows_phys, K = S.pieces;
    // the chain of pieces: where piece q's kept symbols start (j) and end (e) in its own row, and where they go (dst)
    int dst = min(S.k_out[row], J.n_v[row]);     // piece 0 keeps [0, its outgoing seam)
    int j_mine = 0, e_mine = 0, dst_mine = 0;
    for (int q = 1; q < K; ++q) {
        const int nq = J.n_v[q * R + row];
        const int d = (int)rintf((S.t_in[q * R + row] - S.t_out[(q - 1) * R + row]) / J.sps);
        const int j = min(max(S.k_in[q * R + row] - d, 0), nq);
        const int e = max(q < K - 1 ? min(S.k_out[q * R + row], nq) : nq, j);
        if (q == mine) { j_mine = j; e_mine = e; dst_mine = dst; }
        dst += e - j;
    }
    const int ns = min(dst, max_soft);
    if (blockIdx.x == 0 && mine == 1 && tid == 0) {
        n_soft[row] = ns;
        if (J.timing_milli) J.timing_milli[row] = J.timing_v[S.piece_mid * R + row];
    }
    const int cnt = min(e_mine - j_mine, max_soft - dst_mine);      // (a full row: what does not fit is dropped, as the loop does)
    const float2 *__restrict__ B = J.b + ((int64_t)(mine - 1) * R + row) * J.cap_b + j_mine;
    float2 *__restrict__ sr = soft + (int64_t)row * max_soft + dst_mine;
    const int i0 = blockIdx.x * kJoinTile + tid;
    if (i0 >= cnt) return;
    float2 v[kJoinPer];
#pragma unroll
    for (int t = 0; t < kJoinPer; ++t) v[t] = B[min(i0 + 256 * t, cnt - 1)];
#pragma unroll
    for (int t = 0; t < kJoinPer; ++t)
        if (i0 + 256 * t < cnt) sr[i0 + 256 * t] = v[t];
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
    // A thread's products d_i (i = 1 + tid + 256 j) stay in registers between the estimate and the decisions -- the kernel waits
    // on memory for most of its time, so the soft symbols are read once, and with clamped indices all of a thread's loads
    // are in flight together.  (Rows longer than 1 + 256 kKeep symbols: the rest is read twice, as before.)
    constexpr int kKeep = 40;
    float2 dk[kKeep];
    const int last = max(ns - 1, 1);
#pragma unroll
    for (int j = 0; j < kKeep; ++j) dk[j] = product(min(1 + tid + 256 * j, last));
    float a_pp = 0.f, a_qq = 0.f, a_pq = 0.f;
    auto gather = [&](const float2 d) {
        const float p4 = fmaf(d.x, d.x, -(d.y * d.y)), q4 = d.x * d.y;
        a_pp = fmaf(p4, p4, a_pp);
        a_qq = fmaf(q4, q4, a_qq);
        a_pq = fmaf(p4, q4, a_pq);
    };
#pragma unroll
    for (int j = 0; j < kKeep; ++j)
        if (1 + tid + 256 * j < ns) gather(dk[j]);
    for (int i = 1 + tid + 256 * kKeep; i < ns; i += 256) gather(product(i));
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
    auto decide = [&](int i, const float2 d) {
        const float ddx = d.x * rc - d.y * rs, ddy = d.x * rs + d.y * rc;
        hr[i - 1] = (uint8_t)(((ddy < 0.f) ? 2u : 0u) | ((ddx < 0.f) ? 1u : 0u));
        const float lo = fminf(fabsf(ddx), fabsf(ddy)), hi = fmaxf(fabsf(ddx), fabsf(ddy));
        hmin = fminf(hmin, hi);
        const bool take = lo * mhi < mlo * hi;
        mlo = take ? lo : mlo;
        mhi = take ? hi : mhi;
    };
#pragma unroll
    for (int j = 0; j < kKeep; ++j)
        if (1 + tid + 256 * j < ns) decide(1 + tid + 256 * j, dk[j]);
    for (int i = 1 + tid + 256 * kKeep; i < ns; i += 256) decide(i, product(i));
    const float mratio = hmin == 0.f ? 0.f : mlo / mhi;
    float margin = mratio <= 1.f ? atanf(mratio) : 3.4e38f;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) margin = fminf(margin, __shfl_xor(margin, d, 64));
    if (lane == 0) sm[2][wv] = margin;
    __syncthreads();
    if (tid == 0 && min_margin) min_margin[row] = (double)fminf(fminf(sm[2][0], sm[2][1]), fminf(sm[2][2], sm[2][3]));
}

}  // namespace tdm
