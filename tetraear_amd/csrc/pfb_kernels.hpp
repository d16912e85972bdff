// Tetra-mode channeliser: oversampled uniform-DFT polyphase filter bank (definition:
// oracle/pfb_np.py; the reference has no channeliser).
//   y_k[m] = sum_l h[l] x[mD - l] exp(-2 pi i k (mD - l) / M)
//          = sum_{r'} u_m[r'] exp(+2 pi i k r' / M),   u_m[r'] = v_m[(r' + mD) mod M],
//   v_m[r] = sum_p h[r + pM] x[mD - r - pM]                       (P taps per branch)
// One workgroup produces T consecutive output times of all M channels:
//   stage 0  coalesced load + format conversion of (T-1)D + MP input samples into LDS
//   stage A  polyphase branch sums v, written circularly shifted (u)
//   stage B  M-point DFT as M1 x M2 Cooley-Tukey with direct small DFTs, twiddles from LDS
// Output layout [M][n_out] cf32 (channel-major), which is the [carriers][n] input of TETRA mode.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

}  // namespace tdm
