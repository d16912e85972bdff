// microbenchmark (not part of the product): issue cost of vector instructions on gfx950, cycles per wave-instruction per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    f32x2 acc[8];
    f32x2 w[8];
    const unsigned long long msk = 0x5555555555555555ull;
    for (int i = 0; i < 8; ++i) {
        acc[i] = f32x2{(float)threadIdx.x, 1.f};
        w[i] = f32x2{a + i, b - i};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w[(i + 1) & 7]), "v"(w[i]));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(w[i].y), "v"(w[i].x));
                if (MODE == 2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w[(i + 1) & 7]), "v"(w[i]));
                if (MODE == 3) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "+v"(acc[i].x) : "v"(w[i].y), "v"(w[i].x));
                if (MODE == 4) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i].x) : "v"(w[i].y));
                if (MODE == 5) asm volatile("v_rcp_f32 %0, %1" : "+v"(acc[i].x) : "v"(w[i].y));
                if (MODE == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(acc[i].x) : "v"(w[i].y), "v"(w[i].x));
                if (MODE == 7) asm volatile("v_add_u32 %0, %1, %2" : "+v"(acc[i].x) : "v"(w[i].y), "v"(w[i].x));
                if (MODE == 8) asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "+v"(acc[i]) : "v"(w[(i + 1) & 7]), "v"(w[i]));
                if (MODE == 9) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(w[(i + 1) & 7]), "v"(w[i]));
                if (MODE == 10) asm volatile("v_mul_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(w[(i + 1) & 7]), "v"(w[i]));
                if (MODE == 11) asm volatile("v_add_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(w[(i + 1) & 7]), "v"(w[i]));
                if (MODE == 12) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(acc[i]) : "v"(w[i].x));
                if (MODE == 13) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "+v"(acc[i].x) : "v"(w[i].x));
                if (MODE == 15) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %1, %0" : "+v"(*(float __attribute__((ext_vector_type(4))) *)&acc[i & 6]) : "v"(w[i]));
                if (MODE == 16) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "+v"(acc[i].x) : "v"(w[i].y), "v"(w[i].x), "s"(msk));
                if (MODE == 17) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "+v"(acc[i].x) : "v"(w[i].y), "v"(w[i].x) : "vcc");
                if (MODE == 18) asm volatile("v_max_f32 %0, %1, %2" : "+v"(acc[i].x) : "v"(w[i].y), "v"(w[i].x));
                if (MODE == 19) asm volatile("v_floor_f32 %0, %1" : "+v"(acc[i].x) : "v"(w[i].y));
                if (MODE == 20) asm volatile("v_cvt_i32_f32 %0, %1" : "+v"(acc[i].x) : "v"(w[i].y));
                if (MODE == 21) asm volatile("v_alignbit_b32 %0, %1, %2, 8" : "+v"(acc[i].x) : "v"(w[i].y), "v"(w[i].x));
                if (MODE == 22) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(w[i].y), "v"(w[i].x));
                if (MODE == 23) asm volatile("v_mul_f32 %0, %1, %2" : "+v"(acc[i].x) : "v"(w[i].y), "v"(w[i].x));
                if (MODE == 24) asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "+v"(acc[i].x) : "v"(w[i].y), "v"(w[i].x));
                if (MODE == 25) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(*(float __attribute__((ext_vector_type(4))) *)&acc[i & 2]) : "v"(*(float __attribute__((ext_vector_type(4))) *)&w[i & 6]));
                if (MODE == 26) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(*(float __attribute__((ext_vector_type(4))) *)&acc[0]) : "v"(*(float __attribute__((ext_vector_type(4))) *)&w[i & 6]));
                if (MODE == 27) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[i].x) : "v"(w[i].y), "v"(w[i].x));
                if (MODE == 28) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[i].x) : "v"(w[i].y), "v"(w[i].x));
                if (MODE == 29) asm volatile("v_cvt_f32_f16 %0, %1" : "+v"(acc[i].x) : "v"(w[i].y));
                if (MODE == 14) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(*(float __attribute__((ext_vector_type(4))) *)&acc[i & 6]) : "v"(*(float __attribute__((ext_vector_type(4))) *)&w[i & 6]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// Do matrix and vector instructions overlap?  (a) in ONE wavefront: an MFMA followed by NV independent v_fma_f32 / 4-cycle
// instructions; (b) in TWO wavefronts of a SIMD: 512-thread workgroups, wavefronts 0-3 issue MFMAs only, 4-7 vector only
// (ROLE 1: only the matrix wavefronts work, 2: only the vector ones, 3: both)
template <int NV, int KIND>
__global__ __launch_bounds__(256) void k_mix(float *out, int iters, float a, float b)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[2] = {f32x4{a, b, a, b}, f32x4{b, a, b, a}};
    f32x4 w = f32x4{a, b, b, a};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc[r & 1]) : "v"(w));
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i & 7]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[i & 7]) : "v"(a));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + acc[0].x + acc[1].y;
}
template <int ROLE>
__global__ __launch_bounds__(512) void k_two(float *out, int iters, float a, float b)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[2] = {f32x4{a, b, a, b}, f32x4{b, a, b, a}};
    f32x4 w = f32x4{a, b, b, a};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) {
        if (ROLE & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc[r & 1]) : "v"(w));
            }
    } else {
        if (ROLE & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[i & 7]) : "v"(a));   // 4 x 4 cycles = one MFMA
                }
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s + acc[0].x + acc[1].y;
}
template <class K> void run_k(const char *name, K kern, int threads, int wgs_per_cu, double groups_per_iter)
{
    float *out;
    hipMalloc(&out, 256 * 512 * 16 * 4);
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, iters, 1.0001f, 0.9999f);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, iters, 1.0001f, 0.9999f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    // cycles per group (one MFMA [+ its vector instructions]) per wavefront-slot of a SIMD
    printf("%-44s %d WG/CU: %.3f ms, %.2f cycles per group per SIMD at 2.4 GHz\n", name, wgs_per_cu, ms,
           ms * 1e-3 * 2.4e9 / ((double)wgs_per_cu * iters * groups_per_iter));
    hipFree(out);
}
template <int MODE> void run(const char *name, int wgs_per_cu)
{
    float *out;
    hipMalloc(&out, 256 * 256 * 16 * 4);
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.9999f);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.9999f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double per_simd = (double)grid * 4 * iters * 64 / 1024;   // wave-instructions per SIMD
    printf("%-22s %d WG/CU: %.3f ms, %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, wgs_per_cu, ms, ms * 1e-3 * 2.4e9 / per_simd);
    hipFree(out);
}
int main(int argc, char **argv)
{
    for (int w : {1, 2, 4}) {
        run_k("mfma alone (k_mix NV=0)", k_mix<0, 0>, 256, w, 16);
        run_k("mfma + 2 v_fma_f32", k_mix<2, 0>, 256, w, 16);
        run_k("mfma + 4 v_fma_f32", k_mix<4, 0>, 256, w, 16);
        run_k("mfma + 6 v_fma_f32", k_mix<6, 0>, 256, w, 16);
        run_k("mfma + 8 v_fma_f32", k_mix<8, 0>, 256, w, 16);
        run_k("mfma + 2 v_max_f32", k_mix<2, 1>, 256, w, 16);
        run_k("mfma + 3 v_max_f32", k_mix<3, 1>, 256, w, 16);
        run_k("mfma + 4 v_max_f32", k_mix<4, 1>, 256, w, 16);
        run_k("two waves/SIMD: matrix wave only", k_two<1>, 512, w, 16);
        run_k("two waves/SIMD: vector wave only (4 v_max)", k_two<2>, 512, w, 16);
        run_k("two waves/SIMD: both", k_two<3>, 512, w, 16);
    }
    if (argc > 1 && argv[1][0] == 'm') return 0;
    if (argc > 1 && argv[1][0] == 'x') {   // fp16-mix forms only
        run<27>("v_fma_mix_f32 lo", 4); run<28>("v_fma_mix_f32 hi", 4); run<29>("v_cvt_f32_f16", 4); run<1>("v_fma_f32", 4); run<0>("v_pk_fma_f32", 4);
        return 0;
    }
    for (int w : {1, 4}) {
#define R(M, N) if (w == 1) run<M>(N, 1); else run<M>(N, 4);
        R(0, "v_pk_fma_f32") R(1, "v_fma_f32") R(2, "v_fma_f64") R(3, "v_cndmask_b32") R(4, "v_mov_b32_dpp row_shr") R(5, "v_rcp_f32")
        R(6, "v_cvt_pk_bf16_f32") R(7, "v_add_u32") R(8, "v_lshl_add_u64") R(9, "v_pk_mul_f32") R(10, "v_mul_f64") R(11, "v_add_f64")
        R(12, "v_cvt_f64_f32") R(13, "v_cvt_f32_ubyte0") R(14, "v_mfma_16x16x32_bf16") R(15, "v_mfma_16x16x16_bf16") R(16, "v_cndmask sgpr mask") R(17, "v_cmp+v_cndmask vcc (x2)") R(18, "v_max_f32") R(19, "v_floor_f32") R(20, "v_cvt_i32_f32") R(21, "v_alignbit_b32") R(22, "v_med3_f32") R(23, "v_mul_f32") R(24, "v_lshl_or_b32") R(25, "mfma 16x16x32, 2 accumulators alternating") R(26, "mfma 16x16x32, 1 accumulator (dependent)")
    }
    return 0;
}
