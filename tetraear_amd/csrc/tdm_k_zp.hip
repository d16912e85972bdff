// libtetrahip.so, cascade-engine translation unit: k_zp_block for every loader the pipeline uses (gfx950 only).
#include <cstdio>

#include "dev_comm.hpp"
#include "launch.hpp"

namespace tdm {


#define TDM_BLOCK_WAVES 2  // waves per SIMD the block kernel is register-budgeted for
template <int K, int NSEC, int L, int EDGE, class Loader>
__global__ __launch_bounds__(64, (L <= 16 ? 4 : (L <= 24 ? 3 : TDM_BLOCK_WAVES))) void k_zp_block(const ZpParams P, const Loader ld)
{
    __shared__ __attribute__((aligned(16))) double stg[Loader::kStaged ? StageGeom<L>::kDoubles : 2];
    WaveComm cm{stg};
    zp_block_body<K, NSEC, L, EDGE>(P, ld, cm, (int)threadIdx.x, (int)blockIdx.x, (int)blockIdx.y);
}

template <int K, int NSEC, int L, int EDGE, class Loader>
void launch_zp_block(const ZpParams &P, const Loader &ld, int nb, int rows, hipStream_t st)
{
    hipLaunchKernelGGL((k_zp_block<K, NSEC, L, EDGE, Loader>), dim3(nb, rows), dim3(64), 0, st, P, ld);
}

#define TDM_ZP_DEC(FMT, SH) template void launch_zp_block<2, 4, kLDec, kEdgeSos, RawLoader<FMT, SH>>(const ZpParams &, const RawLoader<FMT, SH> &, int, int, hipStream_t);
TDM_ZP_DEC(FMT_CU8, false) TDM_ZP_DEC(FMT_CU8, true) TDM_ZP_DEC(FMT_CS8, false) TDM_ZP_DEC(FMT_CS8, true)
TDM_ZP_DEC(FMT_CF32, false) TDM_ZP_DEC(FMT_CF32, true) TDM_ZP_DEC(FMT_CF64, false) TDM_ZP_DEC(FMT_CF64, true)
#undef TDM_ZP_DEC
template void launch_zp_block<2, 2, kLLpf, kEdgeTf, StagedLoader<DecFixSrc<kLDec>>>(const ZpParams &, const StagedLoader<DecFixSrc<kLDec>> &, int, int, hipStream_t);
template void launch_zp_block<2, 2, kLLpf, kEdgeTf, StagedLoader<PlainC128Src>>(const ZpParams &, const StagedLoader<PlainC128Src> &, int, int, hipStream_t);

}  // namespace tdm
