// Register-resident small DFTs (N = 2,3,4,5 butterflies; composites by Cooley-Tukey, or by the
// Good-Thomas prime-factor map when the factors are coprime, which needs no inner twiddles).
// Sign convention: X[k] = sum_n x[n] exp(+2 pi i k n / N)   (the channeliser's synthesis sign,
// oracle/pfb_np.py).  Everything is resolved at compile time: indices, twiddle values (constexpr
// series), trivial twiddles (1, i, -1, -i) become moves/negations.
#pragma once
#include <type_traits>

#ifndef TDM_HD
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TDM_HD __host__ __device__ __forceinline__
#else
#define TDM_HD inline
#endif
#endif

namespace tdm {

#if defined(__clang__)
typedef float cf32v __attribute__((ext_vector_type(2)));  // (re, im); a+b maps to v_pk_add_f32
#else
struct cf32v {
    float x, y;
};
inline cf32v operator+(cf32v a, cf32v b) { return cf32v{a.x + b.x, a.y + b.y}; }
inline cf32v operator-(cf32v a, cf32v b) { return cf32v{a.x - b.x, a.y - b.y}; }
inline cf32v operator-(cf32v a) { return cf32v{-a.x, -a.y}; }
inline cf32v operator*(cf32v a, float s) { return cf32v{a.x * s, a.y * s}; }
#endif

TDM_HD cf32v cv(float re, float im)
{
    cf32v r;
    r.x = re;
    r.y = im;
    return r;
}
TDM_HD cf32v mul_i(cf32v a) { return cv(-a.y, a.x); }
TDM_HD cf32v cmulv(cf32v a, cf32v w) { return cv(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }

namespace dftc {
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double cos_series(double x)
{
    double t = 1.0, s = 1.0;
    for (int k = 1; k < 24; ++k) {
        t *= -x * x / ((2.0 * k - 1.0) * (2.0 * k));
        s += t;
    }
    return s;
}
constexpr double sin_series(double x)
{
    double t = x, s = x;
    for (int k = 1; k < 24; ++k) {
        t *= -x * x / ((2.0 * k) * (2.0 * k + 1.0));
        s += t;
    }
    return s;
}
// exp(+2 pi i j / n), argument folded into (-pi, pi]
constexpr double tw_re(int j, int n)
{
    j = ((j % n) + n) % n;
    if (2 * j > n) j -= n;
    return cos_series(2.0 * kPi * j / n);
}
constexpr double tw_im(int j, int n)
{
    j = ((j % n) + n) % n;
    if (2 * j > n) j -= n;
    return sin_series(2.0 * kPi * j / n);
}
constexpr int gcd(int a, int b) { return b == 0 ? a : gcd(b, a % b); }
constexpr int pick_factor(int n) { return n % 4 == 0 ? 4 : (n % 2 == 0 ? 2 : (n % 3 == 0 ? 3 : (n % 5 == 0 ? 5 : n))); }
// the k in [0, A*B) with k = ka (mod A), k = kb (mod B); A, B coprime
constexpr int crt(int ka, int kb, int A, int B)
{
    for (int k = 0; k < A * B; ++k)
        if (k % A == ka && k % B == kb) return k;
    return -1;
}

template <int I, int N, class F>
TDM_HD void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
}  // namespace dftc

// a * exp(+2 pi i J / N), J and N compile-time
template <int N, int J>
TDM_HD cf32v mul_tw(cf32v a)
{
    constexpr int j = ((J % N) + N) % N;
    if constexpr (j == 0) {
        return a;
    } else if constexpr (4 * j == N) {
        return mul_i(a);
    } else if constexpr (2 * j == N) {
        return -a;
    } else if constexpr (4 * j == 3 * N) {
        return cv(a.y, -a.x);
    } else {
        constexpr float c = (float)dftc::tw_re(j, N), s = (float)dftc::tw_im(j, N);
        return cv(a.x * c - a.y * s, a.x * s + a.y * c);
    }
}

template <int N>
struct SmallDft;

template <>
struct SmallDft<1> {
    TDM_HD static void run(cf32v (&)[1]) {}
};
template <>
struct SmallDft<2> {
    TDM_HD static void run(cf32v (&x)[2])
    {
        const cf32v a = x[0], b = x[1];
        x[0] = a + b;
        x[1] = a - b;
    }
};
template <>
struct SmallDft<3> {
    TDM_HD static void run(cf32v (&x)[3])
    {
        constexpr float kS = (float)dftc::tw_im(1, 3);  // sqrt(3)/2
        const cf32v t = x[1] + x[2], d = x[1] - x[2];
        const cf32v m = x[0] - t * 0.5f;
        const cf32v s = mul_i(d) * kS;
        x[0] = x[0] + t;
        x[1] = m + s;
        x[2] = m - s;
    }
};
template <>
struct SmallDft<4> {
    TDM_HD static void run(cf32v (&x)[4])
    {
        const cf32v t0 = x[0] + x[2], t1 = x[0] - x[2], t2 = x[1] + x[3], t3 = mul_i(x[1] - x[3]);
        x[0] = t0 + t2;
        x[1] = t1 + t3;
        x[2] = t0 - t2;
        x[3] = t1 - t3;
    }
};
template <>
struct SmallDft<5> {
    TDM_HD static void run(cf32v (&x)[5])
    {
        constexpr float c1 = (float)dftc::tw_re(1, 5), c2 = (float)dftc::tw_re(2, 5);
        constexpr float s1 = (float)dftc::tw_im(1, 5), s2 = (float)dftc::tw_im(2, 5);
        const cf32v t1 = x[1] + x[4], t2 = x[2] + x[3], t3 = x[1] - x[4], t4 = x[2] - x[3];
        const cf32v m1 = x[0] + t1 * c1 + t2 * c2, m2 = x[0] + t1 * c2 + t2 * c1;
        const cf32v u1 = mul_i(t3 * s1 + t4 * s2), u2 = mul_i(t3 * s2 - t4 * s1);
        x[0] = x[0] + t1 + t2;
        x[1] = m1 + u1;
        x[4] = m1 - u1;
        x[2] = m2 + u2;
        x[3] = m2 - u2;
    }
};

template <int N>
struct SmallDft {
    static constexpr int A = dftc::pick_factor(N), B = N / A;
    static_assert(A > 1 && A < N, "SmallDft: prime size without a butterfly");
    static constexpr bool kPfa = dftc::gcd(A, B) == 1;
    static constexpr int in_index(int a, int b) { return kPfa ? (B * a + A * b) % N : B * a + b; }
    static constexpr int out_index(int ka, int kb) { return kPfa ? dftc::crt(ka, kb, A, B) : ka + A * kb; }

    TDM_HD static void run(cf32v (&x)[N])
    {
        cf32v y[N];  // y[ka*B + b]
        dftc::static_for<0, B>([&](auto bb) {
            constexpr int b = decltype(bb)::value;
            cf32v t[A];
            dftc::static_for<0, A>([&](auto aa) {
                constexpr int a = decltype(aa)::value;
                t[a] = x[in_index(a, b)];
            });
            SmallDft<A>::run(t);
            dftc::static_for<0, A>([&](auto kk) {
                constexpr int ka = decltype(kk)::value;
                if constexpr (kPfa)
                    y[ka * B + b] = t[ka];
                else
                    y[ka * B + b] = mul_tw<N, b * ka>(t[ka]);
            });
        });
        dftc::static_for<0, A>([&](auto kk) {
            constexpr int ka = decltype(kk)::value;
            cf32v t[B];
            dftc::static_for<0, B>([&](auto bb) {
                constexpr int b = decltype(bb)::value;
                t[b] = y[ka * B + b];
            });
            SmallDft<B>::run(t);
            dftc::static_for<0, B>([&](auto kk2) {
                constexpr int kb = decltype(kk2)::value;
                x[out_index(ka, kb)] = t[kb];
            });
        });
    }
};

}  // namespace tdm
