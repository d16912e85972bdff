// libtetrahip.so, low-rate stage translation unit: k_lp2 for both sample sources (gfx950 only).
#include <cstdio>

#include "dev_comm.hpp"
#include "launch.hpp"

#ifdef TDM_LP2_TIMING
__device__ unsigned long long g_lp2_dbg[16];   // (global namespace: lp2_kernels.hpp declares it there)
#endif

namespace tdm {

#ifdef TDM_LP2_TIMING
void lp2_timing_dump()
{
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lp2_dbg), sizeof(h)) == hipSuccess)
    {
        static const char *nm[16] = {"stage-in", "fixup+nco", "ext+pass1", "(scans total: see parts)", "pass2", "stage-out", "store+power",
                                     "scan0 in-row steps", "scan0 row totals", "scan0 prefix loop", "scan0 apply+shift",
                                     "scan1 in-row steps", "scan1 row totals", "scan1 prefix loop", "scan1 apply+shift", "scan barrier"};
        double tot = 0;
        for (int i = 0; i < 16; ++i) tot += i == 3 ? 0.0 : (double)h[i];
        for (int i = 0; i < 16; ++i)
            if (i != 3) fprintf(stderr, "lp2 phase %-24s %5.1f %%\n", nm[i], 100.0 * (double)h[i] / tot);
    }
}
#endif

// low-rate stage in one kernel (lp2_kernels.hpp): grid = (chunks, rows), one workgroup of 8 wavefronts per chunk
template <class Src>
__global__ __launch_bounds__(kLp2Lanes, kLp2Waves / 2) void k_lp2(const Lp2Params P, const Src src)
{
    __shared__ __attribute__((aligned(16))) double stg[Lp2Lds::kStage];
    __shared__ __attribute__((aligned(16))) double sml[Lp2Lds::kSmall];
    WgComm cm;
    cm.stg = stg;
    cm.sml = sml;
#ifdef TDM_LP2_STAGGER
    {
        // experiment: the workgroups of the first dispatch round start spread over one workgroup period instead of all at once
        const unsigned id = blockIdx.y * gridDim.x + blockIdx.x;
        if (id < 2u * 256u) {
            const unsigned long long wait = (unsigned long long)((id * 2654435761u) >> 28) * (TDM_LP2_STAGGER / 16);
            const unsigned long long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(32);
        }
    }
#endif
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
