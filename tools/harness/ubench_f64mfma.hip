// microbenchmark (not part of the product): does gfx950's fp64 matrix instruction run BESIDE dependent fp64 vector FMA chains,
// or on the same datapath?  (round-3 VERDICT item 4: could the decimator's non-recurrent 24 % move to the matrix pipe for free?)
//   (a) v_mfma_f64_16x16x4_f64 alone, independent / dependent accumulators: cycles per instruction per SIMD
//   (b) ONE wavefront: an MFMA followed by NV v_fma_f64 on independent chains
//   (c) TWO wavefronts of a SIMD (512-thread workgroups): wavefronts 0-3 only MFMAs, 4-7 only v_fma_f64 of equal stand-alone time
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NV>
__global__ __launch_bounds__(256) void k_mix(double *out, int iters, double a, double b)
{
    f64x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f64x4{a + i, b, a, b - i};
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (NACC > 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[r % (NACC > 0 ? NACC : 1)]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[i & 7]) : "v"(a), "v"(b));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV>
__global__ __launch_bounds__(256) void k_mix44(double *out, int iters, double a, double b)
{
    double acc[4] = {a, b, a + 1, b - 1};
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[r & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[i & 7]) : "v"(a), "v"(b));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + acc[0] + acc[1] + acc[2] + acc[3];
}
// ROLE 1: only the matrix wavefronts work, 2: only the vector ones (VPER v_fma_f64 per group), 3: both
template <int ROLE, int VPER>
__global__ __launch_bounds__(512) void k_two(double *out, int iters, double a, double b)
{
    f64x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f64x4{a + i, b, a, b - i};
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        if (ROLE & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[r & 3]) : "v"(a), "v"(b));
            }
    } else {
        if (ROLE & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
#pragma unroll
                    for (int i = 0; i < VPER; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[i & 7]) : "v"(a), "v"(b));
                }
            }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].w;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <class K> void run_k(const char *name, K kern, int threads, int wgs_per_cu, double groups_per_iter)
{
    double *out;
    hipMalloc(&out, 256 * 512 * 16 * 8);
    const int iters = 1000, grid = 256 * wgs_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, iters, 1.0001, 0.9999);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, iters, 1.0001, 0.9999);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    printf("%-52s %d WG/CU: %.3f ms, %.2f cycles per group per SIMD at 2.4 GHz\n", name, wgs_per_cu, ms,
           ms * 1e-3 * 2.4e9 / ((double)wgs_per_cu * iters * groups_per_iter));
    hipFree(out);
}
int main()
{
    for (int w : {1, 2}) {
        run_k("16 v_fma_f64 alone", k_mix<0, 16>, 256, w, 8);
        run_k("mfma_f64_16x16x4 alone, 4 accumulators", k_mix<4, 0>, 256, w, 8);
        run_k("mfma_f64_16x16x4 alone, 1 accumulator (dependent)", k_mix<1, 0>, 256, w, 8);
        run_k("mfma_f64_16x16x4 + 4 v_fma_f64", k_mix<4, 4>, 256, w, 8);
        run_k("mfma_f64_16x16x4 + 8 v_fma_f64", k_mix<4, 8>, 256, w, 8);
        run_k("mfma_f64_16x16x4 + 16 v_fma_f64", k_mix<4, 16>, 256, w, 8);
        run_k("mfma_f64_4x4x4_4b alone", k_mix44<0>, 256, w, 8);
        run_k("mfma_f64_4x4x4_4b + 4 v_fma_f64", k_mix44<4>, 256, w, 8);
        run_k("two waves/SIMD: matrix wave only", k_two<1, 16>, 512, w, 8);
        run_k("two waves/SIMD: vector wave only (16 v_fma_f64)", k_two<2, 16>, 512, w, 8);
        run_k("two waves/SIMD: both (16 v_fma_f64 per mfma)", k_two<3, 16>, 512, w, 8);
        run_k("two waves/SIMD: vector wave only (8 v_fma_f64)", k_two<2, 8>, 512, w, 8);
        run_k("two waves/SIMD: both (8 v_fma_f64 per mfma)", k_two<3, 8>, 512, w, 8);
    }
    return 0;
}
