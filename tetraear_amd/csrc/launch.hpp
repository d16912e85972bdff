// Host-side launchers of the heavy kernels.  Each family is compiled in its own translation unit (tdm_k_*.hip, explicit
// instantiations of these templates) so that the library builds in parallel and a change to one kernel recompiles one
// unit; tdm_hip.hip only sees the declarations.
#pragma once
#include <hip/hip_runtime.h>

#include "ref_pipeline.hpp"

namespace tdm {

// cascade engine (zp_kernels.hpp): grid (blocks, rows), one wavefront per block
template <int K, int NSEC, int L, int EDGE, class Loader>
void launch_zp_block(const ZpParams &P, const Loader &ld, int nb, int rows, hipStream_t st);
// parallel-form decimator, samples held as doubles (pz_kernels.hpp)
template <int Q, int S, int EDGE, bool SHIFT>
void launch_pz_block(const ZpParams &P, const RawLoaderRT<SHIFT> &ld, int nb, int rows, hipStream_t st);
// parallel-form decimator on the raw bytes
template <int Q, int S, int EDGE, int FMT8>
void launch_pz_raw(const ZpParams &P, const void *iq, int64_t stride, int b_tail, int rows, hipStream_t st);
// low-rate stage in one kernel (lp2_kernels.hpp)
template <class Src>
void launch_lp2(const Lp2Params &P, const Src &src, int rows, hipStream_t st);


}  // namespace tdm
