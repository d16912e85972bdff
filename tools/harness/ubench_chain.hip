// microbenchmark (not part of the product): latency of DEPENDENT vector instructions in ONE wavefront on gfx950
// (cycles of s_memrealtime's 100 MHz clock are too coarse: the kernel is timed by HIP events over many iterations and
// divided by the chain length; one wavefront per CU, so nothing else shares the SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters, float a, float b)
{
    float x = (float)threadIdx.x, y = a;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 xp = {x, y}, yp = {a, b}, zp = {b, a};
    __shared__ float sm[256];
    sm[threadIdx.x] = 0.f;
    __syncthreads();
    int idx = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(b));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %3, %3, %1, %2" : "+v"(x) : "v"(y), "v"(b), "v"(a));   // two independent chains
            if (MODE == 2) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));
            if (MODE == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
            if (MODE == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(y), "v"(b) : "vcc");
            if (MODE == 5) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(idx * 4) : "memory");
            if (MODE == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(idx) : "v"(idx));
            if (MODE == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(xp) : "v"(yp));
            if (MODE == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %2, %2, %1, %1" : "+v"(xp), "+v"(zp) : "v"(yp));   // two independent chains
            if (MODE == 11) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp) : "v"(yp));
            if (MODE == 12) asm volatile("v_pk_mul_f32 %0, %1, %1\n\tv_mul_f32 %2, %2, %3" : "=v"(xp), "+v"(yp), "+v"(x) : "v"(y));      // packed and plain, independent
            if (MODE == 8) asm volatile("v_floor_f32 %0, %0\n\tv_cvt_i32_f32 %0, %0\n\tv_cvt_f32_i32 %0, %0" : "+v"(x));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x + (float)idx + xp.x + xp.y + zp.x + zp.y;
}
template <int MODE>
void run(const char *name, int per_iter)
{
    float *out;
    hipMalloc(&out, 256 * 64 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 64>>>(out, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, 64>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.2f ns per step = %.1f cycles at 2.4 GHz (%d instructions per step)\n", name, ms * 1e6 / ((double)iters * 64), ms * 1e6 / ((double)iters * 64) * 2.4, per_iter);
    hipFree(out);
}
int main()
{
    run<0>("dependent v_fma_f32", 1);
    run<1>("two independent v_fma_f32 chains", 2);
    run<2>("dependent v_mov_b32_dpp quad_perm", 1);
    run<3>("dependent v_rcp_f32", 1);
    run<4>("dependent v_cmp + v_cndmask", 2);
    run<5>("ds_read_b32 + wait", 1);
    run<7>("dependent v_add_u32", 1);
    run<8>("v_floor + v_cvt_i32 + v_cvt_f32", 3);
    run<9>("dependent v_pk_mul_f32", 1);
    run<10>("two independent v_pk_fma_f32 chains", 2);
    run<11>("dependent v_pk_add_f32", 1);
    run<12>("v_pk_mul_f32 + v_mul_f32, independent", 2);
    return 0;
}
