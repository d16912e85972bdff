// libtetrahip.so, raw-integer decimator translation unit: k_pz_raw per decimation factor, compiled in two
// halves (-DTDM_RAW_PART=0 / 1) (gfx950 only).
#include <cstdio>

#include "dev_comm.hpp"
#include "launch.hpp"

namespace tdm {

// raw-integer decimator (pz_raw_body), one launch for all blocks of all rows: blocks without extension samples run the
// narrow body (bytes as they come), the first block and the block(s) with the tail extension the wide one (int16 pairs)
template <int Q, int S, int EDGE, int FMT8>
__global__ __launch_bounds__(64, 2) void k_pz_raw(const ZpParams P, const void *iq, int64_t stride, int b_tail)
{
    __shared__ __attribute__((aligned(16))) double stg[PzEdgeGeom<Q * S, EDGE>::kDoubles];
    WaveComm cm{stg};
    const int blk = (int)blockIdx.x;
    if (blk == 0 || blk >= b_tail)
        pz_raw_body<Q, S, EDGE, FMT8, true>(P, iq, stride, cm, (int)threadIdx.x, blk, (int)blockIdx.y);
    else
        pz_raw_body<Q, S, EDGE, FMT8, false>(P, iq, stride, cm, (int)threadIdx.x, blk, (int)blockIdx.y);
}

template <int Q, int S, int EDGE, int FMT8>
void launch_pz_raw(const ZpParams &P, const void *iq, int64_t stride, int b_tail, int rows, hipStream_t st)
{
    hipLaunchKernelGGL((k_pz_raw<Q, S, EDGE, FMT8>), dim3(P.nb, rows), dim3(64), 0, st, P, iq, stride, b_tail);
}

#define TDM_PZR_INST(Q, S) template void launch_pz_raw<Q, S, kEdgeSos, FMT_CU8>(const ZpParams &, const void *, int64_t, int, int, hipStream_t);
#if TDM_RAW_PART == 0
TDM_PZR_CASES_A(TDM_PZR_INST)
#else
TDM_PZR_CASES_B(TDM_PZR_INST)
#endif
#undef TDM_PZR_INST

}  // namespace tdm
