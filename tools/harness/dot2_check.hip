// experiment (not part of the product): is v_dot2c_f32_bf16 with a (-1, 0) / (0, -1) constant an exact "a - bf16(a)"?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, uint32_t *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = x[2 * i], b = x[2 * i + 1];
    const bf16x2 v = __builtin_convertvector((f32x2{a, b}), bf16x2);
    const uint32_t hi = __builtin_bit_cast(uint32_t, v);
    const float a1 = __builtin_bit_cast(float, hi << 16), b1 = __builtin_bit_cast(float, hi & 0xffff0000u);
    const float ra = a - a1, rb = b - b1;
    float da = a, db = b;
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(da) : "s"(0x0000bf80u), "v"(hi));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(db) : "s"(0xbf800000u), "v"(hi));
    {   // variants: constant in a VGPR; the VOP3P form
        float ea = a, fa;
        uint32_t cv = 0x0000bf80u;
        asm volatile("v_mov_b32 %0, 0xbf80" : "=v"(cv));
        asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(ea) : "v"(hi), "v"(cv));
        asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(fa) : "v"(hi), "v"(cv), "v"(a));
        if (i < 4) printf("i %d a %.9g want %.9g: dot2c(vgpr const) %.9g, dot2 vop3p %.9g\n", i, a, ra, ea, fa);
    }
    out[4 * i] = __builtin_bit_cast(uint32_t, ra);
    out[4 * i + 1] = __builtin_bit_cast(uint32_t, da);
    out[4 * i + 2] = __builtin_bit_cast(uint32_t, rb);
    out[4 * i + 3] = __builtin_bit_cast(uint32_t, db);
}
int main()
{
    const int n = 1 << 16;
    float *hx = new float[2 * n];
    unsigned s = 1;
    for (int i = 0; i < 2 * n; ++i) { s = s * 1664525u + 1013904223u; hx[i] = ((float)(s >> 8) / 8388608.f - 1.f) * std::ldexp(1.f, (int)(s % 40) - 20); }
    float *dx; uint32_t *dout;
    hipMalloc(&dx, 2 * n * 4); hipMalloc(&dout, 4 * n * 4);
    hipMemcpy(dx, hx, 2 * n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    uint32_t *ho = new uint32_t[4 * n];
    hipMemcpy(ho, dout, 4 * n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 2; ++c)
            if (ho[4 * i + 2 * c] != ho[4 * i + 2 * c + 1]) {
                if (bad < 12) printf("x %.9g: a - hi = %.9g (%08x), dot2c = %.9g (%08x)\n", hx[2 * i + c], __builtin_bit_cast(float, ho[4 * i + 2 * c]), ho[4 * i + 2 * c],
                                     __builtin_bit_cast(float, ho[4 * i + 2 * c + 1]), ho[4 * i + 2 * c + 1]);
                ++bad;
            }
    printf("%d of %d differ\n", bad, 2 * n);
    return 0;
}
