// Device-side communication objects the kernel bodies are written against (gfx950): wavefront shuffles as DPP moves,
// workgroup barriers, LDS areas.  tests/emul has the lock-step CPU counterparts.
#pragma once
#include <hip/hip_runtime.h>

#include "zp_common.hpp"

namespace tdm {

struct WaveComm {
    double *stg;
    __device__ __forceinline__ double *stage() { return stg; }
    __device__ __forceinline__ double *edge_slots() { return stg; }
    // ---- lane-row shuffles of the parallel-form scan (DPP: VALU moves, no LDS traffic)
    template <int CTRL, int ROW_MASK, bool BOUND>
    static __device__ __forceinline__ double dpp(double v)
    {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, BOUND);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, BOUND);
        return __hiloint2double(hi, lo);
    }
    template <int K, int CTRL, int ROW_MASK, bool BOUND>
    static __device__ __forceinline__ void dpp2(const double *a, const double *b, double *oa, double *ob)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) { oa[k] = dpp<CTRL, ROW_MASK, BOUND>(a[k]); ob[k] = dpp<CTRL, ROW_MASK, BOUND>(b[k]); }
    }
    // out[lane] = in[lane - d] inside the lane's row of 16, 0 where the row has no such lane (d = 1, 2, 4, 8)
    template <int K>
    __device__ __forceinline__ void row_shr2(const double *a, const double *b, double *oa, double *ob, int d)
    {
        switch (d) {
        case 1: dpp2<K, 0x111, 0xf, true>(a, b, oa, ob); break;
        case 2: dpp2<K, 0x112, 0xf, true>(a, b, oa, ob); break;
        case 4: dpp2<K, 0x114, 0xf, true>(a, b, oa, ob); break;
        default: dpp2<K, 0x118, 0xf, true>(a, b, oa, ob); break;
        }
    }
    template <int K>
    __device__ __forceinline__ void row_shl2(const double *a, const double *b, double *oa, double *ob, int d)
    {
        switch (d) {
        case 1: dpp2<K, 0x101, 0xf, true>(a, b, oa, ob); break;
        case 2: dpp2<K, 0x102, 0xf, true>(a, b, oa, ob); break;
        case 4: dpp2<K, 0x104, 0xf, true>(a, b, oa, ob); break;
        default: dpp2<K, 0x108, 0xf, true>(a, b, oa, ob); break;
        }
    }
    // step 0: rows 1,3 <- lane 15 of rows 0,2 (row_bcast:15); step 1: rows 2,3 <- lane 31 (row_bcast:31); 0 elsewhere
    template <int K>
    __device__ __forceinline__ void row_total_prev2(const double *a, const double *b, double *oa, double *ob, int step)
    {
        if (step == 0) dpp2<K, 0x142, 0xa, false>(a, b, oa, ob);
        else dpp2<K, 0x143, 0xc, false>(a, b, oa, ob);
    }
    // mirror image (no DPP form exists): step 0: rows 0,2 <- lane 0 of rows 1,3; step 1: rows 0,1 <- lane 32; 0 elsewhere
    template <int K>
    __device__ __forceinline__ void row_total_next2(const double *a, const double *b, double *oa, double *ob, int step)
    {
        const int lane = threadIdx.x & 63;
        const int src = step == 0 ? (lane & ~15) + 16 : 32;
        const bool ok = step == 0 ? ((lane & 16) == 0) : (lane < 32);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double x = __shfl(a[k], src, 64), y = __shfl(b[k], src, 64);
            oa[k] = ok ? x : 0.0;
            ob[k] = ok ? y : 0.0;
        }
    }
    // whole-wave shift by one lane, 0 shifted in
    template <int K>
    __device__ __forceinline__ void wave_shr1(const double *a, const double *b, double *oa, double *ob) { dpp2<K, 0x138, 0xf, true>(a, b, oa, ob); }
    template <int K>
    __device__ __forceinline__ void wave_shl1(const double *a, const double *b, double *oa, double *ob) { dpp2<K, 0x130, 0xf, true>(a, b, oa, ob); }
    __device__ __forceinline__ void wave_sync() { __syncthreads(); }  // one wavefront per workgroup
    template <int K>
    __device__ __forceinline__ void shfl_up2(const double *a, const double *b, double *oa, double *ob, int d)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) { oa[k] = __shfl_up(a[k], d, 64); ob[k] = __shfl_up(b[k], d, 64); }
    }
    template <int K>
    __device__ __forceinline__ void shfl_down2(const double *a, const double *b, double *oa, double *ob, int d)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) { oa[k] = __shfl_down(a[k], d, 64); ob[k] = __shfl_down(b[k], d, 64); }
    }
};

// workgroup of kLp2Waves wavefronts (lp2_kernels.hpp): the wavefront shuffles of WaveComm + barrier and two LDS areas
struct WgComm : WaveComm {
    double *sml;
    __device__ __forceinline__ double *small() { return sml; }
    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ void sync() { __syncthreads(); }
};

constexpr int kFinishThreads = 256;

struct BlockComm {
    double *sm;  // [kFinishThreads / 64] LDS (reductions)
    double *buf; // [kPowThreads] LDS (power_fixup scratch) or null
    double *big = nullptr;  // large workgroup scratch (gate FFT) or null
    __device__ __forceinline__ double *smem() { return big; }
    __device__ __forceinline__ double &lds(int i) { return buf[i]; }
    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ int nthreads() const { return blockDim.x; }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    template <class F>
    __device__ __forceinline__ double reduce(double v, F f)
    {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v = f(v, __shfl_xor(v, d, 64));
        const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
        if ((threadIdx.x & 63) == 0) sm[w] = v;
        __syncthreads();
        double r = sm[0];
        for (int i = 1; i < nw; ++i) r = f(r, sm[i]);
        __syncthreads();
        return r;
    }
    __device__ __forceinline__ double reduce_sum(double v) { return reduce(v, [](double a, double b) { return a + b; }); }
    __device__ __forceinline__ double reduce_max(double v) { return reduce(v, [](double a, double b) { return fmax(a, b); }); }
    __device__ __forceinline__ double reduce_min(double v) { return reduce(v, [](double a, double b) { return fmin(a, b); }); }
    // numpy's np.max: a NaN operand wins
    __device__ __forceinline__ double reduce_max_nan(double v) { return reduce(v, [](double a, double b) { return (a != a) ? a : ((b != b) ? b : fmax(a, b)); }); }
};

}  // namespace tdm
