// experiment: which copy kernel reaches the highest HBM rate on this box (the roofline's measured denominator)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_stride(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n)
{
    const size_t S = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * S < n; i += U * S) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(a + i + u * S) : a[i + u * S];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NTS) __builtin_nontemporal_store(v[u], b + i + u * S); else b[i + u * S] = v[u]; }
    }
}
// each workgroup owns a contiguous tile of T KiB
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_tile(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n)
{
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(a + base + u * 256) : a[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NTS) __builtin_nontemporal_store(v[u], b + base + u * 256); else b[base + u * 256] = v[u]; }
}
template <typename F> double timeit(F f, size_t bytes)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * bytes * 20 / (ms * 1e-3) / 1e9;
}
int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    f4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    printf("hipMemcpyDtoD          %.0f GB/s\n", timeit([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, bytes));
#define ST(U, L, S, G) printf("stride U=%d ntl=%d nts=%d grid=%5d  %.0f GB/s\n", U, L, S, G, timeit([&] { hipLaunchKernelGGL((k_stride<U, L, S>), dim3(G), dim3(256), 0, 0, a, b, n); }, bytes));
    for (int g : {512, 1024, 2048, 4096, 8192, 16384}) {
        ST(1, false, false, g) ST(4, false, false, g) ST(8, false, false, g) ST(4, true, true, g) ST(4, true, false, g) ST(4, false, true, g)
    }
#define TL(U, L, S) printf("tile   U=%d ntl=%d nts=%d              %.0f GB/s\n", U, L, S, timeit([&] { hipLaunchKernelGGL((k_tile<U, L, S>), dim3((unsigned)(n / (256 * U))), dim3(256), 0, 0, a, b, n); }, bytes));
    TL(1, false, false) TL(2, false, false) TL(4, false, false) TL(8, false, false) TL(16, false, false) TL(4, true, true) TL(8, true, true) TL(4, true, false) TL(8, false, true)
    return 0;
}
