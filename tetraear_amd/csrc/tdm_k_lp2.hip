// libtetrahip.so, low-rate stage translation unit: k_lp2 for both sample sources (gfx950 only).
#include <cstdio>

#include "dev_comm.hpp"
#include "launch.hpp"

namespace tdm {

// low-rate stage in one kernel (lp2_kernels.hpp): grid = (chunks, rows), one workgroup of four wavefronts per chunk
template <class Src>
__global__ __launch_bounds__(kLp2Lanes, kLp2Waves / 2) void k_lp2(const Lp2Params P, const Src src)
{
    __shared__ __attribute__((aligned(16))) double stg[Lp2Lds::kStage];
    __shared__ __attribute__((aligned(16))) double sml[Lp2Lds::kSmall];
    WgComm cm;
    cm.stg = stg;
    cm.sml = sml;
    lp2_body(P, src, cm, (int)blockIdx.x, (int)blockIdx.y);
}

template <class Src>
void launch_lp2(const Lp2Params &P, const Src &src, int rows, hipStream_t st)
{
    hipLaunchKernelGGL((k_lp2<Src>), dim3(P.n_chunks, rows), dim3(kLp2Lanes), 0, st, P, src);
}
template void launch_lp2<Lp2SrcDec>(const Lp2Params &, const Lp2SrcDec &, int, hipStream_t);
template void launch_lp2<Lp2SrcPlain>(const Lp2Params &, const Lp2SrcPlain &, int, hipStream_t);

}  // namespace tdm
