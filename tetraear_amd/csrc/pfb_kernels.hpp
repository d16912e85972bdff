// Tetra-mode channeliser: oversampled uniform-DFT polyphase filter bank (definition:
// oracle/pfb_np.py; the reference has no channeliser).
//   y_k[m] = sum_l h[l] x[mD - l] exp(-2 pi i k (mD - l) / M)
//          = sum_{r'} u_m[r'] exp(+2 pi i k r' / M),   u_m[r'] = v_m[(r' + mD) mod M],
//   v_m[r] = sum_p h[r + pM] x[mD - r - pM]                       (P taps per branch)
// One workgroup produces T consecutive output times of all M channels:
//   stage 0  coalesced load + format conversion of (T-1)D + MP input samples into LDS
//   stage A  polyphase branch sums v, written circularly shifted (u)
//   stage B  M-point DFT as M1 x M2 Cooley-Tukey
// Output layout [M][n_out] cf32 (channel-major), which is the [carriers][n] input of TETRA mode.
// Two kernels: k_pfb_fft (register-resident mixed-radix small DFTs, any D whose input window fits
// LDS) and k_pfb (direct small DFTs, T chosen to fit) as the general fallback.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "small_dft.hpp"

namespace tdm {

constexpr int kPfbThreads = 256;

struct PfbParams {
    int32_t D;        // decimation
    int32_t T;        // output times per workgroup
    int32_t fmt;      // TDM_CU8 / TDM_CS8 / TDM_CF32
    int32_t pad_;
    int64_t n_in, n_out;
    const float *h;       // [M*P] prototype
    const float2 *W1;     // [M1][M1]  exp(+2 pi i k1 n1 / M1)
    const float2 *WM;     // [M1][M2]  exp(+2 pi i k1 n2 / M)
    const float2 *W2;     // [M2][M2]  exp(+2 pi i k2 n2 / M2)
    int32_t G;            // k_pfb_fft: rounds of TB output times per workgroup
    int32_t pad2_;
    int64_t in_stride;    // bytes between the input streams of a batch (grid.y)
    int64_t out_batch;    // float2 elements between the outputs of a batch
};


__device__ __forceinline__ float2 pfb_load(const void *iq, int fmt, int64_t n)
{
    if (fmt == 0) {  // cu8, pyrtlsdr scaling
        const uint8_t *p = (const uint8_t *)iq + 2 * n;
        return make_float2((float)p[0] * (1.f / 127.5f) - 1.f, (float)p[1] * (1.f / 127.5f) - 1.f);
    } else if (fmt == 1) {
        const int8_t *p = (const int8_t *)iq + 2 * n;
        return make_float2((float)p[0] * (1.f / 128.f), (float)p[1] * (1.f / 128.f));
    }
    return ((const float2 *)iq)[n];
}

__device__ __forceinline__ float2 cfma(float2 a, float2 b, float2 acc)
{
    acc.x = fmaf(a.x, b.x, acc.x);
    acc.x = fmaf(-a.y, b.y, acc.x);
    acc.y = fmaf(a.x, b.y, acc.y);
    acc.y = fmaf(a.y, b.x, acc.y);
    return acc;
}

template <int M1, int M2, int P>
__global__ __launch_bounds__(kPfbThreads) void k_pfb(const void *__restrict__ iq, float2 *__restrict__ out,
                                                      int64_t out_stride, const PfbParams Q)
{
    constexpr int M = M1 * M2;
    constexpr int L = M * P;
    constexpr int RS = M + 1;  // padded row stride of u / A
    extern __shared__ float2 smem[];
    const int T = Q.T, D = Q.D;
    const int nxs = (T - 1) * D + L;
    float2 *xs = smem;                 // [nxs]
    float2 *u = xs + nxs;              // [T][RS]
    float2 *A = u + T * RS;            // [T][RS]
    float2 *w1 = A + T * RS;           // [M1*M1]
    float2 *wm = w1 + M1 * M1;         // [M]
    float2 *w2 = wm + M;               // [M2*M2]
    const int tid = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * T;
    iq = (const char *)iq + (int64_t)blockIdx.y * Q.in_stride;
    out += (int64_t)blockIdx.y * Q.out_batch;
    // ---- stage 0: inputs n = m0*D - (L-1) + i
    const int64_t nbase = m0 * D - (L - 1);
    for (int i = tid; i < nxs; i += kPfbThreads) {
        const int64_t n = nbase + i;
        xs[i] = (n >= 0 && n < Q.n_in) ? pfb_load(iq, Q.fmt, n) : make_float2(0.f, 0.f);
    }
    for (int i = tid; i < M1 * M1; i += kPfbThreads) w1[i] = Q.W1[i];
    for (int i = tid; i < M; i += kPfbThreads) wm[i] = Q.WM[i];
    for (int i = tid; i < M2 * M2; i += kPfbThreads) w2[i] = Q.W2[i];
    __syncthreads();
    // ---- stage A: branch sums, stored circularly shifted by s = (m*D) mod M
    for (int idx = tid; idx < T * M; idx += kPfbThreads) {
        const int mi = idx / M, r = idx - mi * M;
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float hv = Q.h[r + p * M];
            const float2 xv = xs[mi * D + (L - 1) - r - p * M];
            acc.x = fmaf(hv, xv.x, acc.x);
            acc.y = fmaf(hv, xv.y, acc.y);
        }
        const int s = (int)(((m0 + mi) * (int64_t)D) % M);
        int rp = r - s;
        if (rp < 0) rp += M;
        u[mi * RS + rp] = acc;
    }
    __syncthreads();
    // ---- stage B1: for each (mi, n2): M1-point DFT over n1, times the middle twiddle
    for (int idx = tid; idx < T * M2; idx += kPfbThreads) {
        const int mi = idx / M2, n2 = idx - mi * M2;
        float2 x[M1];
#pragma unroll
        for (int n1 = 0; n1 < M1; ++n1) x[n1] = u[mi * RS + M2 * n1 + n2];
#pragma unroll
        for (int k1 = 0; k1 < M1; ++k1) {
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int n1 = 0; n1 < M1; ++n1) acc = cfma(x[n1], w1[k1 * M1 + n1], acc);
            const float2 tw = wm[k1 * M2 + n2];
            A[mi * RS + k1 * M2 + n2] = make_float2(acc.x * tw.x - acc.y * tw.y, acc.x * tw.y + acc.y * tw.x);
        }
    }
    __syncthreads();
    // ---- stage B2: for each (k1, mi): M2-point DFT over n2 -> channels k = k1 + M1*k2
    for (int idx = tid; idx < T * M1; idx += kPfbThreads) {
        const int k1 = idx / T, mi = idx - k1 * T;
        float2 a[M2];
#pragma unroll
        for (int n2 = 0; n2 < M2; ++n2) a[n2] = A[mi * RS + k1 * M2 + n2];
        const int64_t m = m0 + mi;
#pragma unroll 4
        for (int k2 = 0; k2 < M2; ++k2) {
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int n2 = 0; n2 < M2; ++n2) acc = cfma(a[n2], w2[k2 * M2 + n2], acc);
            if (m < Q.n_out) out[(int64_t)(k1 + M1 * k2) * out_stride + m] = acc;
        }
    }
}

// ---- register-FFT variant ---------------------------------------------------------------------
// Workgroup = TB*M2 threads, TB consecutive output times per round, G rounds, any decimation D.
//   load     the round's input window arrives as 4-sample units prefetched into registers during
//            the previous round (one 8-byte load per unit for cu8/cs8, two 16-byte loads for cf32),
//            converted and written to LDS with 16-byte stores
//   stage A  item = (branch r, group of 4 output times): the P taps of the branch stay in registers,
//            lanes walk consecutive r (conflict-free LDS reads and writes); the circular shift
//            (m*D) mod M of each output time is applied to the write index
//   pass 1   thread (mi, n2): SmallDft<M1> over n1 in place in the exchange tile, times the middle twiddle
//   pass 2   thread (k1, mi): SmallDft<M2> over n2, stores channels k1 + M1*k2; consecutive lanes hold
//            consecutive output times, so each channel row receives TB*8 contiguous bytes.
template <int FMT>
struct PfbUnit;   // four consecutive input samples in wire format + validity
template <>
struct PfbUnit<2> {
    float4 a, b;
    uint32_t ok;
    __device__ __forceinline__ void load(const char *base, int64_t n0, int64_t n_in)
    {
        ok = 0;
        a = b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 >= 0 && n0 + 3 < n_in) {
            const float4 *p = (const float4 *)(base + n0 * 8);   // cf32 streams are 8-byte aligned
            __builtin_memcpy(&a, p, 16);
            __builtin_memcpy(&b, p + 1, 16);
            ok = 15u;
        } else {
            float2 t0 = make_float2(0.f, 0.f), t1 = t0, t2 = t0, t3 = t0;
            if (n0 >= 0 && n0 < n_in) t0 = ((const float2 *)base)[n0];
            if (n0 + 1 >= 0 && n0 + 1 < n_in) t1 = ((const float2 *)base)[n0 + 1];
            if (n0 + 2 >= 0 && n0 + 2 < n_in) t2 = ((const float2 *)base)[n0 + 2];
            if (n0 + 3 >= 0 && n0 + 3 < n_in) t3 = ((const float2 *)base)[n0 + 3];
            a = make_float4(t0.x, t0.y, t1.x, t1.y);
            b = make_float4(t2.x, t2.y, t3.x, t3.y);
            ok = 15u;   // zeros already in place
        }
    }
    __device__ __forceinline__ void store(cf32v *dst) const
    {
        ((float4 *)dst)[0] = a;
        ((float4 *)dst)[1] = b;
    }
};
template <int FMT>
struct PfbUnit {   // cu8 (FMT 0) / cs8 (FMT 1): 8 bytes
    uint2 v;
    uint32_t ok;
    __device__ __forceinline__ void load(const char *base, int64_t n0, int64_t n_in)
    {
        v = make_uint2(0u, 0u);
        if (n0 >= 0 && n0 + 3 < n_in) {
            __builtin_memcpy(&v, base + n0 * 2, 8);
            ok = 15u;
        } else {
            ok = 0;
            uint32_t w0 = 0u, w1 = 0u;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (n0 + c >= 0 && n0 + c < n_in) {
                    const uint32_t h = ((const uint16_t *)base)[n0 + c];
                    if (c < 2) w0 |= h << (16 * (c & 1));
                    else w1 |= h << (16 * (c & 1));
                    ok |= 1u << c;
                }
            v = make_uint2(w0, w1);
        }
    }
    __device__ __forceinline__ static cf32v conv(uint32_t h)   // low 16 bits: I, Q
    {
        if (FMT == 0) return cv((float)(h & 255u), (float)((h >> 8) & 255u)) * (1.f / 127.5f) - cv(1.f, 1.f);
        return cv((float)(int8_t)(h & 255u), (float)(int8_t)((h >> 8) & 255u)) * (1.f / 128.f);
    }
    __device__ __forceinline__ void store(cf32v *dst) const
    {
        cf32v s0 = conv(v.x), s1 = conv(v.x >> 16), s2 = conv(v.y), s3 = conv(v.y >> 16);
        if (ok != 15u) {
            if (!(ok & 1u)) s0 = cv(0.f, 0.f);
            if (!(ok & 2u)) s1 = cv(0.f, 0.f);
            if (!(ok & 4u)) s2 = cv(0.f, 0.f);
            if (!(ok & 8u)) s3 = cv(0.f, 0.f);
        }
        float4 lo, hi;
        lo.x = s0.x; lo.y = s0.y; lo.z = s1.x; lo.w = s1.y;
        hi.x = s2.x; hi.y = s2.y; hi.z = s3.x; hi.w = s3.y;
        ((float4 *)dst)[0] = lo;
        ((float4 *)dst)[1] = hi;
    }
};

// waves per SIMD that let `wgs` workgroups of nt threads share a CU
constexpr int pfb_waves_per_simd(int nt, int wgs) { return (wgs * ((nt + 63) / 64) + 3) / 4; }

template <int M1, int M2>
struct PfbFftGeom {
    static constexpr int FS = (M1 * M2) | 1;      // odd stride between output times of the exchange tile
};
// LDS footprint in float2 elements: input window + exchange tile + middle twiddles
template <int M1, int M2, int P, int TB>
constexpr size_t pfb_fft_lds(int D)
{
    const size_t xs = ((size_t)(TB - 1) * D + (size_t)M1 * M2 * P + 3) / 4 * 4;
    return xs + (size_t)TB * PfbFftGeom<M1, M2>::FS + (size_t)M1 * M2;
}

// ---- the three compute phases of a round (TB output times), shared by the kernel variants ----
// stage A: item = tid + it*NT -> (group of 4 output times, branch r).  All LDS reads of an item come
// before its writes (window and tile live in the same LDS array, so the compiler keeps their order);
// the opaque copy of tid keeps the per-item address arithmetic inside the round instead of hoisted
// into spilled registers.
template <int M1, int M2, int P, int TB>
__device__ __forceinline__ void pfb_stage_a(const cf32v *xs, cf32v *A, const float (*hv)[P], int tid, int D, int dmod, int s0)
{
    constexpr int M = M1 * M2, L = M * P, NT = TB * M2, FS = PfbFftGeom<M1, M2>::FS, FG = 4, NI = M1 / 4;
    int tid_v = tid;
    asm volatile("" : "+v"(tid_v));
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int item = tid_v + it * NT;
        const int f = (item / M) * FG, r = item - (item / M) * M;
        const cf32v *px = xs + f * D + (L - 1) - r;
        cf32v acc[FG];
#pragma unroll
        for (int j = 0; j < FG; ++j) {
            acc[j] = px[j * D] * hv[it][0];
#pragma unroll
            for (int p = 1; p < P; ++p) acc[j] += px[j * D - p * M] * hv[it][p];
        }
        int sft = (int)(((uint32_t)s0 + (uint32_t)f * (uint32_t)dmod) % (uint32_t)M);
        cf32v *pa = A + f * FS;
#pragma unroll
        for (int j = 0; j < FG; ++j) {
            int rp = r - sft;   // branch r lands at (r - shift) mod M of output time f + j
            rp += rp < 0 ? M : 0;
            pa[j * FS + rp] = acc[j];
            sft += dmod;
            sft -= sft >= M ? M : 0;
        }
    }
}
// pass 1, in place: thread (mi, n2) owns column n2 of its output time
template <int M1, int M2>
__device__ __forceinline__ void pfb_pass1(cf32v *A, const cf32v *wml, int mi, int n2)
{
    cf32v x[M1];
    cf32v *col = A + mi * PfbFftGeom<M1, M2>::FS + n2;
#pragma unroll
    for (int n1 = 0; n1 < M1; ++n1) x[n1] = col[n1 * M2];
    SmallDft<M1>::run(x);
#pragma unroll
    for (int k1 = 0; k1 < M1; ++k1) col[k1 * M2] = cmulv(x[k1], wml[k1 * M2 + n2]);
}
// pass 2: thread (k1b, mib) transforms row k1b of output time mib and stores channels k1b + M1*k2
template <int M1, int M2>
__device__ __forceinline__ void pfb_pass2(const cf32v *A, cf32v *out, int64_t out_stride, int64_t m0, int64_t n_out,
                                          int k1b, int mib)
{
    cf32v a[M2];
    const cf32v *row = A + mib * PfbFftGeom<M1, M2>::FS + k1b * M2;
#pragma unroll
    for (int j = 0; j < M2; ++j) a[j] = row[j];
    SmallDft<M2>::run(a);
    const int64_t m = m0 + mib;
    if (m < n_out) {
        cf32v *po = out + (int64_t)k1b * out_stride + m;
        const int64_t step = (int64_t)M1 * out_stride;
#pragma unroll
        for (int k2 = 0; k2 < M2; ++k2) {
            __builtin_nontemporal_store(a[k2], po);
            po += step;
        }
    }
}

// NPF: 4-sample units per thread held in registers for the next round; WGS: workgroups per CU aimed at
template <int M1, int M2, int P, int TB, int FMT, int NPF, int WGS>
__global__ __launch_bounds__(TB *M2, pfb_waves_per_simd(TB *M2, WGS)) void k_pfb_fft(
    const void *__restrict__ iq_, cf32v *__restrict__ out_, int64_t out_stride, const PfbParams Q)
{
    static_assert(M1 <= M2, "pass 2 uses the first TB*M1 threads");
    static_assert(TB % 4 == 0 && M1 % 4 == 0, "stage A: M1/4 items of four output times per thread");
    constexpr int M = M1 * M2, L = M * P, NT = TB * M2;
    constexpr int NI = M1 / 4;   // M*(TB/4) items over NT threads
    extern __shared__ cf32v smem_v[];
    const int D = Q.D;
    const int nxs = (TB - 1) * D + L;
    const int nu = (nxs + 3) >> 2;
    const int dmod = D % M;
    cf32v *xs = smem_v;              // [4*nu]
    cf32v *A = smem_v + 4 * nu;      // [TB][FS]: row k1 (or n1) of an output time at k1*M2
    cf32v *wml = smem_v + (pfb_fft_lds<M1, M2, P, TB>(D) - M);   // [M1][M2] middle twiddles
    // (in LDS rather than re-read from memory: loads and stores share vmcnt, so a global load issued after
    //  pass 2's stores would wait for their write acknowledgements)
    const char *iq = (const char *)iq_ + (int64_t)blockIdx.y * Q.in_stride;
    cf32v *out = out_ + (int64_t)blockIdx.y * Q.out_batch;
    const int tid = threadIdx.x;
    const int mi = tid / M2, n2 = tid - mi * M2;
    const int k1b = tid / TB, mib = tid - k1b * TB;
    float hv[NI][P];   // taps of this thread's stage-A branches (the same in every round)
#pragma unroll
    for (int it = 0; it < NI; ++it) {
        const int item = tid + it * NT;
        const int r = item - (item / M) * M;
#pragma unroll
        for (int p = 0; p < P; ++p) hv[it][p] = Q.h[r + p * M];
    }
    for (int i = tid; i < M; i += NT) {
        const float2 w = Q.WM[i];
        wml[i] = cv(w.x, w.y);
    }
    const int64_t round0 = (int64_t)blockIdx.x * Q.G;
    PfbUnit<FMT> pf[NPF];
#pragma unroll
    for (int k = 0; k < NPF; ++k)
        if (tid + k * NT < nu) pf[k].load(iq, round0 * TB * D - (L - 1) + 4 * (int64_t)(tid + k * NT), Q.n_in);
    for (int g = 0; g < Q.G; ++g) {
        const int64_t m0 = (round0 + g) * TB;
        if (m0 >= Q.n_out) break;
        const int64_t nbase = m0 * D - (L - 1);
        if (g == 0) {
#pragma unroll
            for (int k = 0; k < NPF; ++k)
                if (tid + k * NT < nu) pf[k].store(xs + 4 * (tid + k * NT));
            for (int u = tid + NPF * NT; u < nu; u += NT) {   // windows longer than the prefetch depth
                PfbUnit<FMT> t;
                t.load(iq, nbase + 4 * (int64_t)u, Q.n_in);
                t.store(xs + 4 * u);
            }
        }
        __syncthreads();
        if (g + 1 < Q.G) {
#pragma unroll
            for (int k = 0; k < NPF; ++k)
                if (tid + k * NT < nu) pf[k].load(iq, nbase + (int64_t)TB * D + 4 * (int64_t)(tid + k * NT), Q.n_in);
        }
        pfb_stage_a<M1, M2, P, TB>(xs, A, hv, tid, D, dmod, (int)((m0 * (int64_t)D) % M));
        __syncthreads();
        pfb_pass1<M1, M2>(A, wml, mi, n2);
        __syncthreads();
        // The next round's window lands in LDS here, BEFORE this round's stores are issued: loads and stores
        // share vmcnt and retire out of order with respect to each other, so waiting for a load while
        // stores are in flight means waiting for every store acknowledgement.
        if (g + 1 < Q.G) {
#pragma unroll
            for (int k = 0; k < NPF; ++k)
                if (tid + k * NT < nu) pf[k].store(xs + 4 * (tid + k * NT));
            for (int u = tid + NPF * NT; u < nu; u += NT) {
                PfbUnit<FMT> t;
                t.load(iq, nbase + (int64_t)TB * D + 4 * (int64_t)u, Q.n_in);
                t.store(xs + 4 * u);
            }
        }
        if (tid < TB * M1) pfb_pass2<M1, M2>(A, out, out_stride, m0, Q.n_out, k1b, mib);
    }
}

// (Round 4 also built a half-tile variant for the 8-bit formats -- fp16 window, branch sums fused with a radix-2 split of pass 1,
// 80 KB of LDS, two workgroups per compute unit: correct, 0.335 ms against this kernel's 0.237 -- and round 5 one with 16 output
// times per round: 0.340.  Both removed from the source in round 6; docs/HISTORY.md A.2, profiles/r04r_*.)

}  // namespace tdm
