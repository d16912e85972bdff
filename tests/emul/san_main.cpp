// TEST INFRASTRUCTURE: sanitizer run (ASan + UBSan) of the host-side plan / table builders and of the kernel
// bodies in CPU lock-step emulation.  Built and run by tests/test_sanitizers.py (CPU tier only; GPU sanitizers are
// not available on the pool).  Exercises both decimator engines, the fall-back branches and ragged lengths.
#include <cstdint>
#include <cstdio>
#include <vector>

extern "C" int emu_process(double sample_rate, int64_t n, int rows, int fmt, const void *iq, int64_t stride,
                           const double *pre_shift, const double *freq_offset, uint8_t *hard, double *soft,
                           int32_t *n_soft, int32_t *best_phase, double *min_margin, int32_t *max_soft_out);
extern "C" void emu_allow_parallel_form(int on);

int main()
{
    const double rates[] = {2.4e6, 1.8e6, 10e6, 4.9e6, 2.2222e6, 240000.0};
    const int64_t lens[] = {1, 15, 16, 27, 28, 29, 1000, 5003};
    uint32_t lcg = 12345;
    long total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        emu_allow_parallel_form(pass == 0);
        for (double fs : rates)
            for (int64_t n : lens) {
                if (pass == 1 && n > 1000) continue;
                const int rows = 2;
                std::vector<uint8_t> iq((size_t)rows * n * 2);
                for (auto &b : iq) { lcg = lcg * 1664525u + 1013904223u; b = (uint8_t)(lcg >> 24); }
                int32_t ms = 0;
                emu_process(fs, n, rows, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &ms);
                std::vector<uint8_t> hard((size_t)rows * ms);
                std::vector<double> soft((size_t)rows * ms * 2), mm(rows);
                std::vector<int32_t> ns(rows), bp(rows);
                const double fo[2] = {0.0, 1171.875}, pre[2] = {0.0, -25000.0};
                emu_process(fs, n, rows, 0, iq.data(), n, n > 100 ? pre : nullptr, fo, hard.data(), soft.data(), ns.data(),
                            bp.data(), mm.data(), nullptr);
                total += ns[0] + ns[1];
            }
    }
    std::printf("sanitizer run ok, %ld soft symbols\n", total);
    return 0;
}
