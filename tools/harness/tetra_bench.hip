// Experiment harness (not part of the product): times k_tetra_fused<33> alone on rows x n cf32 noise, HIP events,
// `steps` launches back to back after a settling run.  The kernel header is taken from TDM_KERNEL_HEADER so that several
// variants (other -D switches, an older revision of the header) can be built side by side and run in ONE gpurun call:
//     hipcc -O3 -std=c++17 --offload-arch=gfx950 -DTDM_KERNEL_HEADER='"../../tetraear_amd/csrc/tetra_kernels.hpp"' ...
// usage: tetra_bench [rows] [n] [label] [extra dynamic LDS bytes per workgroup]
#include TDM_KERNEL_HEADER
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace tdm;

static std::vector<double> rrc_taps(double sps)
{
    const double alpha = 0.35;
    const int half = (int)std::floor(8 * sps / 2);
    std::vector<double> h(2 * half + 1);
    double e = 0;
    for (int i = -half; i <= half; ++i) {
        const double t = i / sps;
        double v;
        if (std::fabs(t) < 1e-9) v = 1.0 - alpha + 4 * alpha / M_PI;
        else if (std::fabs(std::fabs(t) - 1.0 / (4 * alpha)) < 1e-9)
            v = (alpha / std::sqrt(2.0)) * ((1 + 2 / M_PI) * std::sin(M_PI / (4 * alpha)) + (1 - 2 / M_PI) * std::cos(M_PI / (4 * alpha)));
        else v = (std::sin(M_PI * t * (1 - alpha)) + 4 * alpha * t * std::cos(M_PI * t * (1 + alpha))) / (M_PI * t * (1 - (4 * alpha * t) * (4 * alpha * t)));
        h[i + half] = v;
        e += v * v;
    }
    for (auto &v : h) v /= std::sqrt(e);
    return h;
}

int main(int argc, char **argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 4096, n = argc > 2 ? atoi(argv[2]) : 32768;
    const int dyn = argc > 4 ? atoi(argv[4]) : 0;
    const int settle = 150, steps = 100;
    const double sps = 4.0;
    TetraParams tp{};
    auto h = rrc_taps(sps);
    tp.n = n;
    tp.ntaps = (int)h.size();
    tp.sps = sps;
    tp.inv_sps = 1 / sps;
    tp.max_soft = (int)(n / sps) + 4;
    for (int v = 0; v < kRrcPerThread; ++v) {
        const double g = 256.0 * (v >> 2) + 16.0 * (v & 3);
        tp.ev_c[v] = (float)cos(-2 * M_PI * g / sps);
        tp.ev_s[v] = (float)sin(-2 * M_PI * g / sps);
    }
    tp.tile_c = (float)cos(-2 * M_PI * kRrcTile / sps);
    tp.tile_s = (float)sin(-2 * M_PI * kRrcTile / sps);
    for (size_t i = 0; i < h.size(); ++i) tp.taps[i] = (float)h[i];
#ifndef TDM_HARNESS_NO_TAPOPS
    {
        std::vector<uint32_t> ops(tetra_tap_operand_words(tp.ntaps));
        tetra_tap_operands(tp.taps, tp.ntaps, ops.data());
        uint32_t *d_ops;
        hipMalloc((void **)&d_ops, ops.size() * 4);
        hipMemcpy(d_ops, ops.data(), ops.size() * 4, hipMemcpyHostToDevice);
        tp.tap_ops = d_ops;
    }
#endif
    // pi/4-DQPSK-like rows: random symbols through the same RRC at 4 samples/symbol plus noise, 8 distinct rows
    std::vector<float2> x((size_t)n * 8);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 8388608.f - 1.f; };
    for (int r = 0; r < 8; ++r) {
        const int nsym = n / 4 + 16;
        std::vector<float2> sym(nsym);
        double ph = 0;
        for (auto &q : sym) {
            s = s * 1664525u + 1013904223u;
            ph += M_PI / 4 * (2 * ((s >> 20) & 3) + 1);
            q = make_float2((float)cos(ph), (float)sin(ph));
        }
        const int half = (int)h.size() / 2;
        for (int i = 0; i < n; ++i) {
            double ar = 0, ai = 0;
            for (int k = (i - half + 3) / 4; 4 * k <= i + half; ++k) {
                if (k < 0 || k >= nsym) continue;
                const double w = h[i - 4 * k + half];
                ar += w * sym[k].x;
                ai += w * sym[k].y;
            }
            x[(size_t)r * n + i] = make_float2((float)ar + 0.07f * rnd(), (float)ai + 0.07f * rnd());
        }
    }
    float2 *dx, *dsoft;
    uint8_t *dhard;
    int32_t *dns, *dtm;
    double *dmm;
    hipMalloc(&dx, (size_t)rows * n * 8);
    hipMalloc(&dsoft, (size_t)rows * tp.max_soft * 8);
    hipMalloc(&dhard, (size_t)rows * tp.max_soft);
    hipMalloc(&dns, rows * 4);
    hipMalloc(&dtm, rows * 4);
    hipMalloc(&dmm, rows * 8);
    for (int r = 0; r < rows; ++r) hipMemcpy(dx + (size_t)r * n, x.data() + (size_t)(r % 8) * n, (size_t)n * 8, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto launch = [&]() { hipLaunchKernelGGL((k_tetra_fused<33>), dim3(rows), dim3(kRrcThreads), dyn, 0, dx, (int64_t)n, tp, dsoft, dhard, dns, dtm, dmm); };
    for (int it = 0; it < settle; ++it) launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int it = 0; it < steps; ++it) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= steps;
    std::vector<int32_t> ns(rows);
    hipMemcpy(ns.data(), dns, rows * 4, hipMemcpyDeviceToHost);
    std::vector<uint8_t> hd((size_t)tp.max_soft);
    hipMemcpy(hd.data(), dhard, tp.max_soft, hipMemcpyDeviceToHost);
    unsigned long long chk = 1469598103934665603ull;
    for (int i = 0; i + 1 < ns[0]; ++i) chk = (chk ^ hd[i]) * 1099511628211ull;
    long long tot = 0;
    for (int v : ns) tot += v;
    const double bytes = (double)rows * n * 8 + 9.0 * tot;
    int nblk = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_tetra_fused<33>, kRrcThreads, dyn);
    printf("%-14s rows %d n %d: %.4f ms  %.1f GB/s  frac(8TB/s) %.3f  nsym %lld  wg/CU %d  row0 hash %016llx  %s\n", argc > 3 ? argv[3] : "", rows, n, ms,
           bytes / ms / 1e6, bytes / ms / 1e6 / 8000, tot, nblk, chk, hipGetErrorString(hipGetLastError()));
#ifdef TDM_TETRA_TIMING
    unsigned long long hdg[16];
    hipMemcpyFromSymbol(hdg, HIP_SYMBOL(g_tetra_dbg), sizeof(hdg));
    const char *nm[12] = {"loop top", "wait barrier 1", "rrc (mfma)", "statistic", "ring store", "wait barrier 2", "stage+fetch", "estimates", "-", "symbol range", "symbols (farrow)", "finish"};
    double tot_ = 0;
    for (int q = 0; q < 12; ++q) tot_ += (double)hdg[q];
    for (int q = 0; q < 12; ++q) printf("    %-18s %5.1f %%\n", nm[q], 100.0 * hdg[q] / tot_);
#endif
    return 0;
}
