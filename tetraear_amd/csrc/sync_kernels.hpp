// Burst synchronisation on the hard-symbol stream: the immediate consumer of the demodulator
// (SURVEY.md section 8(f) N1).  Restates, for parallel evaluation,
//   TetraDecoder.symbols_to_bits  (tetraear/core/decoder.py:140-169, the 0..3 pass-through branch)
//   TetraDecoder.find_sync        (tetraear/core/decoder.py:171-295)
// find_sync slides a 22-bit window over the bit stream, counts matches against the two training
// sequences, accepts a position when matches/22 >= threshold, then skips 250 bits; if nothing is
// accepted it retries once with an adaptive threshold derived from the best correlation seen.
//   sync_count_body : one thread per bit position -> match counts against TS1 and TS2
//   sync_walk_body  : one thread per carrier      -> the reference's sequential accept/skip walk
#pragma once
#include "zp_common.hpp"

namespace tdm {

constexpr int kSyncLen = 22;
// decoder.py:192-195, first array element = most significant bit of the word below
constexpr uint32_t kTS1 = 0b1101000011101001110100u;
constexpr uint32_t kTS2 = 0b0111101001000011011100u;
constexpr int kSyncSkip = 250;

// bit i of the stream: symbol i/2, (val >> 1) for even i, (val & 1) for odd i  (decoder.py:156,167)
// from_bits != 0: `sym` already is the bit stream, one byte per bit (find_sync's own argument);
// a byte that is neither 0 nor 1 can match no pattern element and is flagged in `bad`.
TDM_HD uint32_t sync_window(const uint8_t *sym, int64_t pos, int from_bits, uint32_t &bad)
{
    uint32_t w = 0;
    bad = 0;
#pragma unroll
    for (int k = 0; k < kSyncLen; ++k) {
        const int64_t i = pos + k;
        uint32_t bit;
        if (from_bits) {
            const uint32_t v = sym[i];
            bit = v & 1u;
            bad = (bad << 1) | (v > 1u ? 1u : 0u);
        } else {
            const uint32_t v = sym[i >> 1] & 3u;
            bit = (i & 1) ? (v & 1u) : (v >> 1);
            bad <<= 1;
        }
        w = (w << 1) | bit;
    }
    return w;
}

TDM_HD int popc22(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}

// counts[pos] = (matches with TS1) | (matches with TS2) << 8
TDM_HD void sync_count_body(const uint8_t *sym, int64_t n_bits, int64_t pos, int from_bits, uint16_t *counts)
{
    if (pos + kSyncLen > n_bits) return;
    uint32_t bad;
    const uint32_t w = sync_window(sym, pos, from_bits, bad);
    const int c1 = kSyncLen - popc22((w ^ kTS1) | bad);
    const int c2 = kSyncLen - popc22((w ^ kTS2) | bad);
    counts[pos] = (uint16_t)(c1 | (c2 << 8));
}

// The reference's walk (decoder.py:225-283), one carrier.  Returns the number of positions written.
TDM_HD int sync_walk_body(const uint16_t *counts, int64_t n_bits, double threshold, int32_t *positions,
                          int max_pos, double *max_corr_out)
{
    int n_found = 0;
    double max_corr = 0.0;
    if (n_bits < kSyncLen) {
        *max_corr_out = 0.0;
        return 0;
    }
    const int64_t num_windows = n_bits - kSyncLen + 1;
    int64_t i = 0;
    while (i < num_windows) {
        const int c1 = counts[i] & 0xff, c2 = counts[i] >> 8;
        bool found = false;
        // patterns are tried in dict order, TS1 then TS2; TS2 is not looked at once TS1 matched
        const double corr1 = (double)c1 / (double)kSyncLen;
        if (corr1 > max_corr) max_corr = corr1;
        if (corr1 >= threshold) {
            found = true;
        } else {
            const double corr2 = (double)c2 / (double)kSyncLen;
            if (corr2 > max_corr) max_corr = corr2;
            if (corr2 >= threshold) found = true;
        }
        if (found) {
            if (n_found < max_pos) positions[n_found] = (int32_t)i;
            ++n_found;
            i += kSyncSkip;
            continue;
        }
        ++i;
    }
    if (n_found == 0 && max_corr > 0.75 && max_corr >= (threshold - 0.15)) {
        double adaptive = max_corr - 0.02;
        if (adaptive < 0.75) adaptive = 0.75;
        if (adaptive < threshold) {
            // nothing was accepted, so every position was visited: re-scan the stored correlations
            int64_t blocked_until = -1;  // positions < blocked_until were marked "seen"
            for (int64_t pos = 0; pos < num_windows; ++pos) {
                const int c1 = counts[pos] & 0xff, c2 = counts[pos] >> 8;
                const int cb = c1 > c2 ? c1 : c2;
                if (cb == 0) continue;  // best_corr_at_pos > 0 filter
                const double corr = (double)cb / (double)kSyncLen;
                if (corr >= adaptive && pos >= blocked_until) {
                    if (n_found < max_pos) positions[n_found] = (int32_t)pos;
                    ++n_found;
                    blocked_until = pos + kSyncSkip;
                }
            }
        }
    }
    *max_corr_out = max_corr;
    return n_found;
}

}  // namespace tdm
