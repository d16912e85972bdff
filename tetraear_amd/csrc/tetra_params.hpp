// TETRA mode: constants, kernel parameters and the launch entry (the kernels live in tetra_kernels.hpp, compiled in
// tdm_tetra.hip; the rest of the library sees only this header).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "tetra_taps.hpp"

namespace tdm {

constexpr int kRrcMaxTaps = 96;
#define TDM_TETRA_THREADS 256
constexpr int kRrcThreads = TDM_TETRA_THREADS;            // 256: three workgroups per CU fit in LDS (0.465 ms per 4096 x 32768); 512: two (0.48 ms)
#define TDM_TETRA_PER 8
constexpr int kRrcPerThread = TDM_TETRA_PER;              // outputs per thread and tile (8: a wavefront owns 512 consecutive outputs)
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

// one launch of the fused receiver on `rows` carriers; returns false when no kernel is instantiated for tp.ntaps
// row_list / n_rows (device, or null): the launch covers the rows listed -- workgroup i takes row row_list[i], workgroups
// past *n_rows leave at once -- instead of all `rows`
// fmt8: 0 cf32 input, 1 cu8, 2 cs8 (tetra_kernels.hpp TetraIn8)
bool tetra_launch(const TetraParams &tp, int rows, const void *x, int fmt8, int64_t in_stride, float2 *soft, uint8_t *hard,
                  int32_t *n_soft, int32_t *timing_milli, double *min_margin, hipStream_t stream, const int32_t *row_list = nullptr,
                  const int32_t *n_rows = nullptr);

// TDM_MODE_TETRA_GARDNER (tetra_gardner_kernels.hpp): the three launches, each on its own so that the caller can time them.
// y: [rows][y_pitch] cf32 matched-filter output (y_pitch even, >= tp.n); false when no kernel is instantiated for tp.ntaps
// fmt8: 0 cf32 input, 1 cu8, 2 cs8 (converted where the window is staged)
bool tetra_mf_launch(const TetraParams &tp, int rows, const void *x, int fmt8, int64_t in_stride, float2 *y, int64_t y_pitch, hipStream_t stream);
void tetra_gardner_loop_launch(const TetraParams &tp, int rows, const float2 *y, int64_t y_pitch, float2 *soft, int32_t *n_soft,
                               int32_t *timing_milli, hipStream_t stream);
// the matched filter and the loop in ONE kernel (the filter output stays in LDS); false when not instantiated for tp.ntaps
// (the caller then makes the three launches)
int tetra_gardner_fused_per_cu(int ntaps);                 // workgroups of the fused kernel a compute unit holds (0: not instantiated)
bool tetra_gardner_fused_available(int ntaps, int rows, int fmt8 = 0);   // fmt8: tetra_launch's   // instantiated for the tap count, and not slower than the three launches at this size
// seg (tetra_gardner_kernels.hpp GardnerSeg, or null): the carriers as two virtual carriers each (tp.n = a half's length,
// rows = 2 x the physical carriers, outputs into the caller's temporaries); tetra_decide_launch joins them afterwards
// Two segments per carrier (rounds of 4096 carriers or fewer: one loop wavefront per compute unit leaves seven eighths of the
// chip idle, and the loop's time is symbols x instructions whatever shares the unit).  The loop filter is a contraction: a
// second loop started anywhere converges onto the first one's trajectory with the loop's time constant (~75 symbols at 1 %
// noise bandwidth), so a carrier's chunk is walked as K = 2, 4 or 8 virtual carriers -- pieces of n_v samples, piece p starting p * seg_step samples
// into the chunk, n_v such that a piece -- started next to the eye by a feed-forward estimate -- has run for 384 warm-up symbols when it reaches the seam at which it takes over -- and
// the symbol streams are joined at the K - 1 seams, each `margin` samples before a piece's end (clear of its matched filter's
// edge): a piece records the index and the instant of its first symbol at or behind its incoming and its outgoing seam, and
// k_tetra_gardner_join chains them (the instants tell whether two loops mean the same symbol).  oracle/tetra_np.py
// gardner_segments / demod_gardner(segments=K) is the same construction in fp64.
struct GardnerSeg {
    int32_t rows_phys;        // physical carriers; the kernel's `rows` counts the virtual ones (pieces x rows_phys), piece 0's first
    int32_t pieces;           // K (0: no segments)
    int32_t seg_step;         // samples from a piece's first sample to the next piece's first
    int32_t seam_in, seam_out;   // the seams in a piece's own sample coordinates (the first piece has no seam_in, the last no seam_out)
    int32_t piece_mid, k_mid; // the PHYSICAL chunk's middle symbol: which piece keeps it and its index there (timing_milli comes from it)
    int32_t *k_in, *k_out;    // [rows] index of the piece's first symbol at or behind the seam (its symbol count if none)
    float *t_in, *t_out;      // [rows] that symbol's instant relative to the seam, samples
    // piece 0 writes straight into the caller's rows (its symbols before the seam are final where they land); only the other
    // pieces go through a temporary.  Needs rows_phys % 16 == 0 (a loop wavefront's sixteen carriers all of one kind);
    // null: all pieces into the temporary, [pieces rows_phys][max_soft]
    float2 *soft_a;           // [rows_phys][pitch_a]
    int32_t pitch_a;
    // plan option "gardner_ff_start": the FIRST loop of a chunk (the only one of a whole chunk) starts at the feed-forward
    // estimate as well (the later pieces always do); usable with pieces == 0
    int32_t ff_first;
};

bool tetra_gardner_fused_launch(const TetraParams &tp, int rows, const void *x, int fmt8, int64_t in_stride, float2 *soft, int32_t *n_soft,
                                int32_t *timing_milli, hipStream_t stream, const GardnerSeg *seg = nullptr);
// seg != null: the carriers' pieces are joined first (k_tetra_gardner_join): soft_b [(pieces - 1) rows][cap_b] the symbols of
// pieces 1.., n_v / timing_v [pieces rows] the pieces' counts and timing; soft / n_soft / timing_milli receive the joined carrier
void tetra_decide_launch(const TetraParams &tp, int rows, float2 *soft, int32_t *n_soft, uint8_t *hard, double *min_margin,
                         hipStream_t stream, const GardnerSeg *seg = nullptr, const float2 *soft_b = nullptr, int cap_b = 0,
                         const int32_t *n_v = nullptr, const int32_t *timing_v = nullptr, int32_t *timing_milli = nullptr);

}  // namespace tdm
