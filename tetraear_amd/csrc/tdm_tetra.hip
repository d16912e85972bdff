// libtetrahip.so, TETRA-mode translation unit: the fused receiver kernel and its launch (gfx950 only).
#include "tetra_kernels.hpp"

namespace tdm {

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
