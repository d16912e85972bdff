// Host-side bin bookkeeping of scipy.signal.resample's FFT method (scipy/signal/_signaltools.py,
// `resample`, complex-input branch) as called by SignalProcessor.resample (processor.py:46-48).
#pragma once
#include <cstdint>
#include <unordered_map>
#include <vector>

namespace tdm {

struct ResamplePlan {
    int64_t n = 0, num = 0;
    std::vector<int64_t> src_bins;  // unique input-spectrum bins that are needed (stage 1 outputs)
    std::vector<int64_t> term_src;  // stage 2 terms: index into src_bins ...
    std::vector<int64_t> term_dst;  // ... output-spectrum bin it lands on ...
    std::vector<double> term_w;     // ... and its weight
};

inline ResamplePlan build_resample_plan(int64_t n, int64_t num)
{
    ResamplePlan p;
    p.n = n;
    p.num = num;
    if (n <= 0 || num <= 0) return p;
    const int64_t N = num < n ? num : n;
    const int64_t nyq = N / 2 + 1;
    std::unordered_map<int64_t, int64_t> index_of;
    auto add = [&](int64_t src, int64_t dst, double w) {
        auto it = index_of.find(src);
        int64_t idx;
        if (it == index_of.end()) {
            idx = (int64_t)p.src_bins.size();
            p.src_bins.push_back(src);
            index_of.emplace(src, idx);
        } else {
            idx = it->second;
        }
        p.term_src.push_back(idx);
        p.term_dst.push_back(dst);
        p.term_w.push_back(w);
    };
    const bool even = (N % 2 == 0);
    const bool down = num < n, up = n < num;
    // Y[:nyq] = X[:nyq]
    for (int64_t k = 0; k < nyq; ++k) {
        double w = 1.0;
        if (even && up && k == N / 2) w = 0.5;  // "select the component at frequency +N/2 and halve it"
        add(k, k, w);
    }
    // Y[nyq-N:] = X[nyq-N:]   (negative frequencies), only if N > 2
    if (N > 2)
        for (int64_t i = nyq - N; i < 0; ++i) add(n + i, num + i, 1.0);
    if (even) {
        // Y[-N/2] += X[-N/2]; scipy writes it as the slice [-N//2 : -N//2+1], which is EMPTY for N == 2
        if (down && N > 2) add(n - N / 2, num - N / 2, 1.0);
        else if (up) add(N / 2, num - N / 2, 0.5);       // Y[num-N/2] = Y[N/2] (already halved)
    }
    return p;
}

}  // namespace tdm
