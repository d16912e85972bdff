// libtetrahip.so, parallel-form decimator translation unit (samples as doubles): one kernel per decimation factor;
// compiled twice, -DTDM_PZ_SHIFT=0 (no input-rate pre-shift) and =1 (gfx950 only).
#include <cstdio>

#include "dev_comm.hpp"
#include "launch.hpp"

namespace tdm {

// parallel-form decimator (pz_kernels.hpp): one wavefront per block of 64 lanes x Q*S samples
template <int Q, int S, int EDGE, bool SHIFT>
__global__ __launch_bounds__(64, (Q * S <= 32 ? 2 : 1)) void k_pz_block(const ZpParams P, const RawLoaderRT<SHIFT> ld)
{
    __shared__ __attribute__((aligned(16))) double stg[PzEdgeGeom<Q * S, EDGE>::kDoubles];
    WaveComm cm{stg};
    pz_block_body<Q, S, EDGE>(P, ld, cm, (int)threadIdx.x, (int)blockIdx.x, (int)blockIdx.y);
}

template <int Q, int S, int EDGE, bool SHIFT>
void launch_pz_block(const ZpParams &P, const RawLoaderRT<SHIFT> &ld, int nb, int rows, hipStream_t st)
{
    hipLaunchKernelGGL((k_pz_block<Q, S, EDGE, SHIFT>), dim3(nb, rows), dim3(64), 0, st, P, ld);
}

#define TDM_PZ_INST(Q, S) template void launch_pz_block<Q, S, kEdgeSos, (TDM_PZ_SHIFT != 0)>(const ZpParams &, const RawLoaderRT<(TDM_PZ_SHIFT != 0)> &, int, int, hipStream_t);
TDM_PZ_CASES(TDM_PZ_INST)
#undef TDM_PZ_INST

}  // namespace tdm
