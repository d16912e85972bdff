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
                if (MODE == 14) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(*(float __attribute__((ext_vector_type(4))) *)&acc[i & 6]) : "v"(*(float __attribute__((ext_vector_type(4))) *)&w[i & 6]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
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
int main()
{
    for (int w : {1, 4}) {
#define R(M, N) if (w == 1) run<M>(N, 1); else run<M>(N, 4);
        R(0, "v_pk_fma_f32") R(1, "v_fma_f32") R(2, "v_fma_f64") R(3, "v_cndmask_b32") R(4, "v_mov_b32_dpp row_shr") R(5, "v_rcp_f32")
        R(6, "v_cvt_pk_bf16_f32") R(7, "v_add_u32") R(8, "v_lshl_add_u64") R(9, "v_pk_mul_f32") R(10, "v_mul_f64") R(11, "v_add_f64")
        R(12, "v_cvt_f64_f32") R(13, "v_cvt_f32_ubyte0") R(14, "v_mfma_16x16x32_bf16") R(15, "v_mfma_16x16x16_bf16") R(16, "v_cndmask sgpr mask") R(17, "v_cmp+v_cndmask vcc (x2)") R(18, "v_max_f32") R(19, "v_floor_f32") R(20, "v_cvt_i32_f32") R(21, "v_alignbit_b32") R(22, "v_med3_f32") R(23, "v_mul_f32") R(24, "v_lshl_or_b32") R(25, "mfma 16x16x32, 2 accumulators alternating") R(26, "mfma 16x16x32, 1 accumulator (dependent)")
    }
    return 0;
}
