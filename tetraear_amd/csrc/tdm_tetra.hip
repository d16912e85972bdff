// libtetrahip.so, TETRA-mode translation unit: the fused receiver kernel and its launch (gfx950 only).
#include <cstdio>

#include "tetra_kernels.hpp"

namespace tdm {

#ifdef TDM_TETRA_TIMING
void tetra_timing_dump()
{
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tetra_dbg), sizeof(h)) != hipSuccess) return;
    static const char *names[12] = {"loop top", "barrier 1", "rrc", "statistic", "ring store", "barrier 2", "stage+fetch", "estimates", "-", "symbol range", "farrow", "finish"};
    unsigned long long tot = 0;
    for (int i = 0; i < 12; ++i) tot += h[i];
    for (int i = 0; i < 12; ++i) fprintf(stderr, "tetra phase %-12s %5.1f %%\n", names[i], 100.0 * (double)h[i] / (double)(tot ? tot : 1));
}
#endif

bool tetra_launch(const TetraParams &tp, int rows, const float2 *x, int64_t in_stride, float2 *soft, uint8_t *hard,
                  int32_t *n_soft, int32_t *timing_milli, double *min_margin, hipStream_t stream)
{
    switch (tp.ntaps) {
#define TDM_RRC_CASE(NT) case NT: hipLaunchKernelGGL((k_tetra_fused<NT>), dim3(rows), dim3(kRrcThreads), 0, stream, x, in_stride, tp, soft, hard, n_soft, timing_milli, min_margin); return true;
        TDM_RRC_CASE(17) TDM_RRC_CASE(25) TDM_RRC_CASE(33) TDM_RRC_CASE(35) TDM_RRC_CASE(41) TDM_RRC_CASE(49) TDM_RRC_CASE(57) TDM_RRC_CASE(65)
#undef TDM_RRC_CASE
    default: return false;
    }
}

}  // namespace tdm
