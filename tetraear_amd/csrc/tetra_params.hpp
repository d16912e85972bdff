// TETRA mode: constants, kernel parameters and the launch entry (the kernels live in tetra_kernels.hpp, compiled in
// tdm_tetra.hip; the rest of the library sees only this header).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace tdm {

constexpr int kRrcMaxTaps = 96;
#ifndef TDM_TETRA_THREADS
#define TDM_TETRA_THREADS 256
#endif
constexpr int kRrcThreads = TDM_TETRA_THREADS;            // 256: three workgroups per CU fit in LDS (0.465 ms per 4096 x 32768); 512: two (0.48 ms)
#ifndef TDM_TETRA_PER
#define TDM_TETRA_PER 8
#endif
constexpr int kRrcPerThread = TDM_TETRA_PER;              // outputs per thread and tile (8: a wavefront owns 512 consecutive outputs)
constexpr int kRrcRun = 16;                               // outputs per row of the matched filter's matrix-core tiles
constexpr int kRrcTile = kRrcThreads * kRrcPerThread;     // samples per round of a workgroup
constexpr int kTimingBlock = 256;                         // samples per timing sub-block (TB)
constexpr int kTimingHalfWin = 2;                         // sub-blocks averaged each side (TW)
constexpr int kMaxTimingBlocks = 512;
constexpr int kTileBlocks = kRrcTile / kTimingBlock;      // sub-blocks per tile
// Matched-filter outputs kept in LDS: after tile i the ring holds samples [T(i+1) - kRing, T(i+1)), T = kRrcTile.  Round
// i emits the symbols whose nominal position lies in [T i - 640, T(i+1) - 640) (640 = 2.5 sub-blocks: their two timing
// estimates are final then), so the ring covers them with kRing - T - 640 = 128 samples to spare below and 640 above:
// that is how far the unwrapped timing estimate may carry a symbol instant from its nominal position (16 symbols
// behind / 80 ahead at 8 samples/symbol) before the symbol takes the direct path (its four filter outputs recomputed
// from the input).  Round 3: 1024 -> 768 extra samples, which with the unpadded staging planes brings a workgroup
// under 40 KB of LDS: four workgroups per compute unit, and 4096 carriers are exactly four dispatch rounds.
constexpr int kRing = kRrcTile + 768;
constexpr int kTauRing = 32;                              // timing estimates kept (a round reads at most kTileBlocks + 2 of them)

struct TetraParams {
    int32_t n;          // samples per carrier chunk
    int32_t ntaps;      // odd
    int32_t max_soft;   // capacity of per-carrier symbol outputs
    int32_t pad_;
    double sps;         // samples per symbol (sample_rate / 18000)
    double inv_sps;
    float ev_c[kRrcPerThread], ev_s[kRrcPerThread];   // exp(-2 pi i g / sps) at g = 256 (v >> 2) + 16 (v & 3): symbol-clock phasor of a lane's outputs inside a wavefront's two sub-blocks
    float tile_c, tile_s;                             // exp(-2 pi i kRrcTile / sps): advance per tile
    float taps[kRrcMaxTaps];
    // the matched filter's constant matrix-core operands as the lanes hold them (tetra_tap_operands below), in device
    // memory: [step][leading / trailing bf16 half][lane] x 16 bytes.  Null: every wavefront forms them from taps[] itself
    // (about 150 vector instructions at the start of every carrier).
    const uint32_t *tap_ops;
};

// ---- host side of the tap operands.  Lane l holds, for step s, the eight Toeplitz entries T[32 s + 8 (l >> 4) + e][l & 15]
// = h[32 s + 8 (l >> 4) + e - (l & 15)], e < 8 (zero outside the taps), as four dwords of bf16 pairs: the leading halves
// and the trailing halves (h = h1 + h2, h1 = bf16(h), h2 = bf16(h - h1), round to nearest even -- v_cvt_pk_bf16_f32).
inline uint32_t tetra_bf16_rne(float f)
{
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;   // (finite taps)
}
inline int tetra_tap_steps(int ntaps) { return (kRrcRun + ntaps - 1 + 31) / 32; }
inline size_t tetra_tap_operand_words(int ntaps) { return (size_t)tetra_tap_steps(ntaps) * 2 * 64 * 4; }
inline void tetra_tap_operands(const float *taps, int ntaps, uint32_t *out)
{
    const int ks = tetra_tap_steps(ntaps);
    for (int s = 0; s < ks; ++s)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
                uint32_t hi[2], lo[2];
                for (int c = 0; c < 2; ++c) {
                    const int t = 32 * s + 8 * (lane >> 4) + 2 * j + c - (lane & 15);
                    const float h = (t >= 0 && t < ntaps) ? taps[t] : 0.f;
                    hi[c] = tetra_bf16_rne(h);
                    const uint32_t hb = hi[c] << 16;
                    float h1;
                    __builtin_memcpy(&h1, &hb, 4);
                    lo[c] = tetra_bf16_rne(h - h1);
                }
                out[((size_t)(2 * s) * 64 + lane) * 4 + j] = hi[0] | (hi[1] << 16);
                out[((size_t)(2 * s + 1) * 64 + lane) * 4 + j] = lo[0] | (lo[1] << 16);
            }
}

// one launch of the fused receiver on `rows` carriers; returns false when no kernel is instantiated for tp.ntaps
#ifdef TDM_TETRA_TIMING
void tetra_timing_dump();
#endif
bool tetra_launch(const TetraParams &tp, int rows, const float2 *x, int64_t in_stride, float2 *soft, uint8_t *hard,
                  int32_t *n_soft, int32_t *timing_milli, double *min_margin, hipStream_t stream);

}  // namespace tdm
