// libtetrahip.so, TETRA-mode translation unit: the fused receiver kernel and its launch (gfx950 only).
#include <cstdio>

#include "tetra_kernels.hpp"
#include "tetra_gardner_kernels.hpp"

namespace tdm {


bool tetra_launch(const TetraParams &tp, int rows, const void *x, int fmt8, int64_t in_stride, float2 *soft, uint8_t *hard,
                  int32_t *n_soft, int32_t *timing_milli, double *min_margin, hipStream_t stream, const int32_t *row_list,
                  const int32_t *n_rows)
{
    switch (tp.ntaps) {
#define TDM_RRC_ARGS dim3(rows), dim3(kRrcThreads), 0, stream, x, in_stride, tp, soft, hard, n_soft, timing_milli, min_margin, row_list, n_rows
#define TDM_RRC_CASE(NT) case NT:                                                      \
        if (fmt8 == 1) hipLaunchKernelGGL((k_tetra_fused<NT, 1>), TDM_RRC_ARGS);         \
        else if (fmt8 == 2) hipLaunchKernelGGL((k_tetra_fused<NT, 2>), TDM_RRC_ARGS);    \
        else hipLaunchKernelGGL((k_tetra_fused<NT, 0>), TDM_RRC_ARGS);                   \
        return true;
        TDM_RRC_CASE(17) TDM_RRC_CASE(25) TDM_RRC_CASE(33) TDM_RRC_CASE(35) TDM_RRC_CASE(41) TDM_RRC_CASE(49) TDM_RRC_CASE(57) TDM_RRC_CASE(65)
#undef TDM_RRC_CASE
#undef TDM_RRC_ARGS
    default: return false;
    }
}

bool tetra_mf_launch(const TetraParams &tp, int rows, const void *x, int fmt8, int64_t in_stride, float2 *y, int64_t y_pitch, hipStream_t stream)
{
    const int tiles = (tp.n + kMfTile - 1) / kMfTile;
    const dim3 grid((unsigned)((tiles + kMfTilesPerWg - 1) / kMfTilesPerWg), (unsigned)rows);
    switch (tp.ntaps) {
#define TDM_MF_CASE(NT) case NT:                                                                                                             \
        if (fmt8 == 1) hipLaunchKernelGGL((k_tetra_mf<NT, 1>), grid, dim3(kMfThreads), 0, stream, x, in_stride, tp, y, y_pitch);              \
        else if (fmt8 == 2) hipLaunchKernelGGL((k_tetra_mf<NT, 2>), grid, dim3(kMfThreads), 0, stream, x, in_stride, tp, y, y_pitch);         \
        else hipLaunchKernelGGL((k_tetra_mf<NT, 0>), grid, dim3(kMfThreads), 0, stream, x, in_stride, tp, y, y_pitch);                        \
        return true;
        TDM_MF_CASE(17) TDM_MF_CASE(25) TDM_MF_CASE(33) TDM_MF_CASE(35) TDM_MF_CASE(41) TDM_MF_CASE(49) TDM_MF_CASE(57) TDM_MF_CASE(65)
#undef TDM_MF_CASE
    default: return false;
    }
}

static GardnerConsts gardner_gains()
{
    // loop filter gains of the definition (oracle/tetra_np.py demod_gardner: noise bandwidth 1 % of the symbol rate,
    // damping 0.7071, detector gain 2.7 per symbol; Rice, Digital Communications, eq. C.61); x 100: see GardnerConsts
    const double bn_t = 0.01, zeta = 0.7071, kp = 2.7;
    const double th = bn_t / (zeta + 0.25 / zeta);
    const double den = 1.0 + 2.0 * zeta * th + th * th;
    return GardnerConsts{(float)(100.0 * 4.0 * zeta * th / den / kp), (float)(100.0 * 4.0 * th * th / den / kp)};
}

void tetra_gardner_loop_launch(const TetraParams &tp, int rows, const float2 *y, int64_t y_pitch, float2 *soft, int32_t *n_soft,
                               int32_t *timing_milli, hipStream_t stream)
{
    const GardnerConsts G = gardner_gains();
    hipLaunchKernelGGL(k_tetra_gardner<0>, dim3((unsigned)((rows + kGQuads - 1) / kGQuads)), dim3(64), 0, stream, y, y_pitch, tp, G, rows,
                       soft, n_soft, timing_milli, GardnerSeg{});
}

static const void *gardner_fused_kernel(int ntaps)
{
    switch (ntaps) {
#define TDM_GF_CASE(NT) case NT: return (const void *)k_tetra_gardner<NT>;
        TDM_GF_CASE(17) TDM_GF_CASE(25) TDM_GF_CASE(33) TDM_GF_CASE(35) TDM_GF_CASE(41) TDM_GF_CASE(49) TDM_GF_CASE(57) TDM_GF_CASE(65)
#undef TDM_GF_CASE
    default: return nullptr;
    }
}

// The fused kernel's time is one workgroup's (the loop's serial chain) times the ROUNDS the launch takes: workgroups over what
// the device holds at once (compute units x workgroups per unit: two up to 41 taps -- 79 KB of LDS each --, one above).
// With two per unit it stays ahead of the three launches at any size (4096 / 8192 / 16 384 carriers at 4 samples per
// symbol: 1.35 / 1.86 / 3.51 ms against 1.85 / 2.4 / 3.57); with one per unit only while the launch is a single round (the
// loop on its own needs 33 KB and runs four workgroups per unit: 8192 carriers at 8 samples per symbol 1.96 ms).
bool tetra_gardner_fused_available(int ntaps, int rows, int fmt8)
{
    const void *fn = gardner_fused_kernel(ntaps);
    if (!fn) return false;
    if (fmt8 && ntaps != 33 && ntaps != 35) return false;   // (tetra_gardner_fused_launch: the 8-bit instantiations)
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return true;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * kGWaves, 0) != hipSuccess || per_cu < 1) return true;
    return per_cu >= 2 || (int64_t)(rows + kGQuads - 1) / kGQuads <= (int64_t)cus * per_cu;
}

int tetra_gardner_fused_per_cu(int ntaps)
{
    const void *fn = gardner_fused_kernel(ntaps);
    int per_cu = 0;
    if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * kGWaves, 0) != hipSuccess) return 0;
    return per_cu;
}

bool tetra_gardner_fused_launch(const TetraParams &tp, int rows, const void *x_, int fmt8, int64_t in_stride, float2 *soft, int32_t *n_soft,
                                int32_t *timing_milli, hipStream_t stream, const GardnerSeg *seg)
{
    const GardnerSeg S = seg ? *seg : GardnerSeg{};
    const GardnerConsts G = gardner_gains();
    const dim3 grid((unsigned)((rows + kGQuads - 1) / kGQuads)), block(64 * kGWaves);
    const float2 *x = (const float2 *)x_;
    if (fmt8) {   // 8-bit input: the tap counts of 4 and 4.44 samples per symbol (72 / 80 kS/s) -- the others take the three launches
        switch (tp.ntaps) {
#define TDM_GF8_CASE(NT) case NT:                                                                                                                  \
            if (fmt8 == 1) hipLaunchKernelGGL((k_tetra_gardner<NT, 1>), grid, block, 0, stream, x, in_stride, tp, G, rows, soft, n_soft, timing_milli, S); \
            else hipLaunchKernelGGL((k_tetra_gardner<NT, 2>), grid, block, 0, stream, x, in_stride, tp, G, rows, soft, n_soft, timing_milli, S);           \
            return true;
            TDM_GF8_CASE(33) TDM_GF8_CASE(35)
#undef TDM_GF8_CASE
        default: return false;
        }
    }
    switch (tp.ntaps) {
#define TDM_GF_CASE(NT) case NT: hipLaunchKernelGGL((k_tetra_gardner<NT>), grid, block, 0, stream, x, in_stride, tp, G, rows, soft, n_soft, timing_milli, S); return true;
        TDM_GF_CASE(17) TDM_GF_CASE(25) TDM_GF_CASE(33) TDM_GF_CASE(35) TDM_GF_CASE(41) TDM_GF_CASE(49) TDM_GF_CASE(57) TDM_GF_CASE(65)
#undef TDM_GF_CASE
    default: return false;
    }
}

void tetra_decide_launch(const TetraParams &tp, int rows, float2 *soft, int32_t *n_soft, uint8_t *hard, double *min_margin,
                         hipStream_t stream, const GardnerSeg *seg, const float2 *soft_b, int cap_b, const int32_t *n_v,
                         const int32_t *timing_v, int32_t *timing_milli)
{
    if (seg) {
        const GardnerJoin J{soft_b, cap_b, n_v, timing_v, timing_milli, (float)tp.sps};
        const dim3 grid((unsigned)((cap_b + kJoinTile - 1) / kJoinTile), (unsigned)rows, (unsigned)(seg->pieces - 1));
        hipLaunchKernelGGL(k_tetra_gardner_join, grid, dim3(256), 0, stream, soft, (int)tp.max_soft, n_soft, *seg, J);
    }
    hipLaunchKernelGGL(k_tetra_decide, dim3((unsigned)rows), dim3(256), 0, stream, soft, (int)tp.max_soft, n_soft, hard, min_margin);
}

}  // namespace tdm
