// Discrete Fourier transforms of ARBITRARY length for SignalProcessor.resample (processor.py:35-49 -> scipy.signal.resample,
// FFT method) on long inputs.  tdm_resample evaluates short inputs as direct sums (zp_kernels.hpp dft_terms_body: exact
// twiddles, O(n x kept bins)); at 131 072 samples that is 10^10 terms, so from 2^24 terms on the transforms run as
//   * power-of-two lengths: Stockham autosort radix-2 passes through two buffers (one launch per pass, a thread per
//     butterfly, every twiddle an exact sincospi of a dyadic fraction);
//   * any other length L: Bluestein's chirp-z form -- exp(-+2 pi i nk / L) = c(n) c(k) conj(c)(k - n) with the chirp
//     c(m) = exp(-+ i pi m^2 / L), m^2 taken modulo 2 L in integers so that the phase argument is exact --: one cyclic
//     convolution of length M = the power of two >= 2 L - 1, i.e. three power-of-two transforms.
// fp64 throughout; against scipy's pocketfft-based resample the result differs by ~1e-14 of the largest output
// (tests/test_gpu_parity.py test_resample_long_inputs_vs_scipy).  The reference never calls resample() on the process() path
// (SURVEY 8(a) row a8); this keeps the kept method usable on capture-sized arrays (131 072 samples: milliseconds).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tdm {

typedef double f64c __attribute__((ext_vector_type(2)));   // (re, im)

__device__ __forceinline__ f64c fft_cmul(f64c a, f64c b) { return f64c{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// exp(sign i pi m / L) for an integer 0 <= m < 2 L
__device__ __forceinline__ f64c fft_root2(uint64_t m, int64_t L, double sign)
{
    double s, c;
    sincospi((double)m / (double)L, &s, &c);
    return f64c{c, sign * s};
}
// the chirp exp(sign i pi k^2 / L), k < 2^31
__device__ __forceinline__ f64c fft_chirp(int64_t k, int64_t L, double sign)
{
    return fft_root2(((uint64_t)k * (uint64_t)k) % (uint64_t)(2 * L), L, sign);
}

// one Stockham radix-2 pass: sub-transforms of length p become sub-transforms of length 2 p (p = 1, 2, 4, ..., M / 2)
__global__ void k_fft2_pass(const f64c *__restrict__ x, f64c *__restrict__ y, int64_t half, int64_t p, double sign)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const int64_t k = i & (p - 1), j = ((i - k) << 1) + k;
    const f64c u0 = x[i], u1 = fft_cmul(x[i + half], fft_root2((uint64_t)k, p, sign));   // exp(sign i pi k / p)
    y[j] = u0 + u1;
    y[j + p] = u0 - u1;
}

// a[k] = in[k] * chirp(k) (k < L), 0 beyond, scaled; b[k] = conj(chirp)(|k|) wrapped around M
__global__ void k_bluestein_pre(const f64c *__restrict__ in, f64c *__restrict__ a, f64c *__restrict__ b, int64_t L, int64_t M, double sign)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    f64c av{0.0, 0.0}, bv{0.0, 0.0};
    if (k < L) {
        const f64c c = fft_chirp(k, L, sign);
        av = fft_cmul(in[k], c);
        bv = f64c{c.x, -c.y};
    } else if (M - k < L) {
        const f64c c = fft_chirp(M - k, L, sign);
        bv = f64c{c.x, -c.y};
    }
    a[k] = av;
    b[k] = bv;
}
__global__ void k_fft_cmul(f64c *__restrict__ a, const f64c *__restrict__ b, int64_t M)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < M) a[k] = fft_cmul(a[k], b[k]);
}
// out[k] = chirp(k) * conv[k] * scale   (conv: the unnormalised inverse transform of the product; scale carries 1 / M)
__global__ void k_bluestein_post(const f64c *__restrict__ conv, f64c *__restrict__ out, int64_t L, double sign, double scale)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= L) return;
    const f64c v = fft_cmul(conv[k], fft_chirp(k, L, sign));
    out[k] = f64c{v.x * scale, v.y * scale};
}
__global__ void k_fft_scale_copy(const f64c *__restrict__ in, f64c *__restrict__ out, int64_t n, double scale)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = f64c{in[k].x * scale, in[k].y * scale};
}
// resample's spectrum bookkeeping (resample_plan.hpp): Y[dst] += w X[bins[src]]; at most two terms share a dst (the folded
// Nyquist bin), and a sum of two is the same in either order
__global__ void k_resample_terms(const f64c *__restrict__ X, f64c *__restrict__ Y, const int64_t *__restrict__ bins, const int64_t *__restrict__ src,
                                 const int64_t *__restrict__ dst, const double *__restrict__ w, int64_t n_terms)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms) return;
    const f64c v = X[bins[src[t]]];
    double *y = (double *)(Y + dst[t]);
    atomicAdd(y, v.x * w[t]);
    atomicAdd(y + 1, v.y * w[t]);
}

}  // namespace tdm
