// TETRA mode, host side of the matched filter's constant operands (no HIP types: also compiled by the CPU test tier,
// tests/emul).  See tetra_kernels.hpp for the layout the lanes of a wavefront hold.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace tdm {

constexpr int kRrcRun = 16;                               // outputs per row of the matched filter's matrix-core tiles

// Lane l holds, for step s, the eight Toeplitz entries T[32 s + 8 (l >> 4) + e][l & 15]
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


}  // namespace tdm
