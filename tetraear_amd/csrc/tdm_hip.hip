// libtetrahip.so -- gfx950 build of the C-ABI in include/tetrahip.h.
//
// __global__ wrappers around the kernel bodies of zp_kernels.hpp, the HIP backend of
// ref_pipeline.hpp, plans, and the extern "C" entry points.  There is no CPU compute path in
// this library: every compute entry point needs a HIP device and fails loudly without one.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <string>
#include <vector>

#include "../../include/tetrahip.h"
#include "ref_pipeline.hpp"
#include "dev_comm.hpp"
#include "launch.hpp"
#include "resample_plan.hpp"
#include "fft_kernels.hpp"
#include "sync_kernels.hpp"
#include "tetra_params.hpp"
#include "pfb_kernels.hpp"
#include "gate_kernels.hpp"
#include "detect_kernels.hpp"
#include "occupancy_kernels.hpp"

using namespace tdm;

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
// stream used by the device-pointer forms of the stand-alone entry points (gate, channeliser, find_sync):
// the null stream unless the caller chose one with tdm_set_stream -- typically a plan's own stream, which puts
// those launches in order with tdm_process_device without any host synchronisation
static thread_local hipStream_t g_cur_stream = nullptr;

// ------------------------------------------------------------------------------------------
// debug / experiment switches (tdm_debug_set, include/tetrahip.h).  The library never reads the environment: a switch is
// set by a call the header documents, or not at all.
// ------------------------------------------------------------------------------------------
struct DebugSwitch { const char *key; std::atomic<long long> value; long long dflt; };
static DebugSwitch g_debug[] = {
    {"no_raw", {0}, 0},                  // 1: cu8 plans made from now on never take the raw-integer decimator
    {"raw_min_blocks", {-1}, -1},        // >= 0: blocks below which a batch stays on the double-based decimator (plans made from now on)
    {"gardner_fused", {1}, 1},           // 0: Gardner mode as three launches (matched filter -> HBM -> loop -> decisions)
    {"pfb_direct", {0}, 0},              // 1: channeliser plans made from now on use the direct-DFT kernel
    {"pfb_rounds", {0}, 0},              // > 0: rounds per channeliser workgroup (plans made from now on)
    {"gardner_segments", {1}, 1},        // tdm_plan_option "gardner_segments" for TDM_MODE_TETRA_GARDNER plans made from now on (0, 1, K, -1)
};
static DebugSwitch *debug_find(const char *key)
{
    if (!key) return nullptr;
    for (auto &d : g_debug)
        if (std::strcmp(d.key, key) == 0) return &d;
    return nullptr;
}
static long long debug_value(const char *key)
{
    DebugSwitch *d = debug_find(key);
    return d ? d->value.load(std::memory_order_relaxed) : 0;
}

static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(TDM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

static int use_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(TDM_ERR_NO_DEVICE, std::string("no HIP device available (") +
                                           (e == hipSuccess ? "count=0" : hipGetErrorString(e)) +
                                           "); libtetrahip has no CPU path");
    if (device < 0 || device >= n) return fail(TDM_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    return TDM_OK;
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
template <int K, int NSEC, bool FWD>
__global__ __launch_bounds__(256) void k_zp_carry(const ZpParams P, int nb, int rows)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ch = (int)(idx & 1);
    const int64_t rb = idx >> 1;
    const int b = (int)(rb % nb);
    const int row = (int)(rb / nb);
    if (row >= rows) return;
    if (FWD)
        zp_carry_fwd_body<K, NSEC>(P, row, b, ch);
    else
        zp_carry_bwd_body<K, NSEC>(P, row, b, ch);
}

template <int NSEC>
__global__ __launch_bounds__(256) void k_pz_carry(const ZpParams P, int nb, int rows)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ch = (int)(idx & 1);
    const int64_t rb = idx >> 1;
    const int b = (int)(rb % nb);
    const int row = (int)(rb / nb);
    if (row >= rows) return;
    pz_carry_body<NSEC>(P, row, b, ch);
}

template <int D, int L>
__global__ __launch_bounds__(256) void k_zp_fixup(const ZpParams P, double *out, int64_t out_row_stride,
                                                  const double *freq_offset, double fs_out)
{
    const int row = blockIdx.y;
    zp_fixup_body<D, L>(P, row, (int)blockIdx.x, (int)threadIdx.x, (int)blockDim.x,
                        out + (int64_t)row * out_row_stride * 2, freq_offset, fs_out);
}

template <class Loader>
__global__ __launch_bounds__(256) void k_convert(const Loader ld, int64_t n, double *out, const double *freq_offset,
                                                 double fs)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (j >= n) return;
    convert_body(ld, row, j, out + (int64_t)row * n * 2, freq_offset, fs);
}

__global__ __launch_bounds__(kFinishThreads) void k_finish(FinishArgs fa)
{
    __shared__ double sm[kFinishThreads / 64];
    __shared__ double buf[kFinishThreads];
    BlockComm cm{sm, buf};
    finish_body(fa, cm, (int)blockIdx.x);
}

template <int D, int L>
__global__ __launch_bounds__(kPowThreads) void k_power_fixup(const ZpParams P, int64_t n, double *z, int sps,
                                                            double *partials, int n_pblk)
{
    static_assert(kPowSub * kMaxSps <= kPowThreads, "one thread per (power block, phase)");
    __shared__ double buf[kPowSub * kPowThreads];
    BlockComm cm{nullptr, buf};
    const int row = blockIdx.y;
    power_fixup_body<D, L>(P, cm, row, (int)blockIdx.x, z ? z + (int64_t)row * n * 2 : nullptr, n, sps,
                           partials + (int64_t)row * n_pblk * kMaxSps, n_pblk);
}

__global__ __launch_bounds__(kFinishThreads) void k_dft_terms(const int64_t *o_list, const double *in, int64_t n_terms,
                                                           const int64_t *src, const int64_t *freq,
                                                           const double *weight, int64_t n, double sign, double scale,
                                                           double *out)
{
    __shared__ double sm[kFinishThreads / 64];
    BlockComm cm{sm, nullptr};
    dft_terms_body(cm, (int64_t)blockIdx.x, o_list, in, n_terms, src, freq, weight, n, sign, scale, out);
}

__global__ __launch_bounds__(kFinishThreads) void k_gate(const GateArgs A)
{
    __shared__ double sm[kFinishThreads / 64];
    __shared__ double big[4 * kGateFft];
    BlockComm cm{sm, nullptr, big};
    gate_body(A, cm, (int)blockIdx.x);
}

__global__ __launch_bounds__(kFinishThreads) void k_detect(const DetectArgs A)
{
    __shared__ double sm[kFinishThreads / 64];
    BlockComm cm{sm, nullptr, nullptr};
    detect_body(A, cm, (int)blockIdx.x);
}

// from_bits bit 0: units are bits (else dibit symbols); bit 1: n_units holds n_soft of tdm_process_device (symbols + 1)
// bits of a row, never more than the scratch row holds (max_bits - 1): with device pointers the counts stay on the
// device and the documented bound "rows are bounded by row_stride" is enforced here, not assumed
__device__ __forceinline__ int64_t sync_row_bits(int32_t n_units, int from_bits, int64_t max_bits)
{
    int64_t n = n_units;
    if (from_bits & 2) n = n > 0 ? n - 1 : 0;
    if (n < 0) n = 0;
    n = (from_bits & 1) ? n : 2 * n;
    return n < max_bits - 1 ? n : max_bits - 1;
}

__global__ __launch_bounds__(256) void k_sync_count(const uint8_t *sym, int64_t row_stride, const int32_t *n_units,
                                                    int from_bits, int64_t max_bits, uint16_t *counts)
{
    const int row = blockIdx.y;
    const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n_bits = sync_row_bits(n_units[row], from_bits, max_bits);
    sync_count_body(sym + (int64_t)row * row_stride, n_bits, pos, from_bits & 1, counts + (int64_t)row * max_bits);
}

__global__ __launch_bounds__(64) void k_sync_walk(const uint16_t *counts, const int32_t *n_units, int from_bits,
                                                  int64_t max_bits, int rows, double threshold, int32_t *positions,
                                                  int max_pos, int32_t *n_pos, double *max_corr)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const int64_t n_bits = sync_row_bits(n_units[row], from_bits, max_bits);
    double mc;
    n_pos[row] = sync_walk_body(counts + (int64_t)row * max_bits, n_bits, threshold,
                                positions + (int64_t)row * max_pos, max_pos, &mc);
    max_corr[row] = mc;
}

__global__ __launch_bounds__(256) void k_shift(const double *x, double *y, int64_t n, double f, double fs)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) shift_body(x, y, j, f, fs);
}

// ------------------------------------------------------------------------------------------
// stage timing (HIP events around each launch, on the stream the kernels run on)
// ------------------------------------------------------------------------------------------
enum Stage { ST_DEC_BLOCK = 0, ST_DEC_CARRY, ST_DEC_FIXUP, ST_CONVERT, ST_LPF_BLOCK, ST_LPF_CARRY, ST_LPF_FIXUP, ST_FINISH, ST_TETRA, ST_TETRA_MF, ST_TETRA_LOOP, ST_TETRA_DECIDE, ST_COUNT };
static const char *kStageNames[ST_COUNT] = {"dec_block", "dec_carry", "dec_fixup", "convert",
                                            "lpf_block", "lpf_carry", "lpf_fixup", "finish",
                                            "tetra_fused", "tetra_mf", "tetra_gardner_loop", "tetra_decide"};

struct StageTimer {
    bool on = false;
    struct Rec { int stage; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
    void release_all()
    {
        for (auto &r : recs) { pool.push_back(r.a); pool.push_back(r.b); }
        recs.clear();
    }
    ~StageTimer()
    {
        release_all();
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

struct HipBackend {
    hipStream_t stream = nullptr;
    StageTimer *timer = nullptr;
    hipError_t err = hipSuccess;
    int device = 0;

    struct Scope {
        HipBackend &be; int stage; hipEvent_t a{}, b{}; bool on;
        Scope(HipBackend &be_, int st) : be(be_), stage(st), on(be_.timer && be_.timer->on)
        {
            if (on) { a = be.timer->get(); b = be.timer->get(); (void)hipEventRecord(a, be.stream); }
        }
        ~Scope()
        {
            if (on) { (void)hipEventRecord(b, be.stream); be.timer->recs.push_back({stage, a, b}); }
            hipError_t e = hipGetLastError();
            if (e != hipSuccess && be.err == hipSuccess) be.err = e;
        }
    };

    template <int K, int NSEC, int L, int EDGE, class Loader>
    void zp_block(const ZpParams &P, Loader ld, int nb, int rows)
    {
        Scope s(*this, NSEC == 4 ? ST_DEC_BLOCK : ST_LPF_BLOCK);
        launch_zp_block<K, NSEC, L, EDGE, Loader>(P, ld, nb, rows, stream);
    }
    template <int Q, int S, int EDGE, bool SHIFT>
    void pz_block(const ZpParams &P, const RawLoaderRT<SHIFT> &ld, int nb, int rows)
    {
        Scope s(*this, ST_DEC_BLOCK);
        launch_pz_block<Q, S, EDGE, SHIFT>(P, ld, nb, rows, stream);
    }
    template <int Q, int S, int EDGE, int FMT8>
    void pz_raw(const ZpParams &P, const void *iq, int64_t stride, int b_tail, int rows)
    {
        Scope s(*this, ST_DEC_BLOCK);
        launch_pz_raw<Q, S, EDGE, FMT8>(P, iq, stride, b_tail, rows, stream);
    }
    template <class Src>
    void lp2(const Lp2Params &P, const Src &src, int rows)
    {
        Scope s(*this, ST_LPF_BLOCK);
        launch_lp2<Src>(P, src, rows, stream);
    }
    // low-rate kernel, then the finish stage.  (Attaching the finish to the low-rate kernel -- the last chunk of a carrier
    // to complete runs it, a ticket per carrier -- was measured: the release fence every workgroup then needs is an
    // agent-scope one, i.e. an L2 write-back on this 8-XCD part, and the launch took 2.8 ms instead of 0.37 + 0.03.)
    template <class Src>
    void lp2_finish(const Lp2Params &P, const Src &src, const FinishArgs &fa, int rows)
    {
        lp2(P, src, rows);
        finish(fa, rows);
    }
    template <int K, int NSEC>
    void zp_carry(const ZpParams &P, int nb, int rows)
    {
        Scope s(*this, NSEC == 4 ? ST_DEC_CARRY : ST_LPF_CARRY);
        const int64_t threads = (int64_t)rows * nb * 2;
        const unsigned blocks = (unsigned)((threads + 255) / 256);
        if (P.pform) {   // both carries in one launch
            hipLaunchKernelGGL((k_pz_carry<NSEC>), dim3(blocks), dim3(256), 0, stream, P, nb, rows);
            return;
        }
        hipLaunchKernelGGL((k_zp_carry<K, NSEC, true>), dim3(blocks), dim3(256), 0, stream, P, nb, rows);
        hipLaunchKernelGGL((k_zp_carry<K, NSEC, false>), dim3(blocks), dim3(256), 0, stream, P, nb, rows);
    }
    template <int D, int L>
    void zp_fixup(const ZpParams &P, int nb, int rows, double *out, int64_t out_row_stride,
                  const double *freq_offset, double fs_out)
    {
        Scope s(*this, D == 8 ? ST_DEC_FIXUP : ST_LPF_FIXUP);
        hipLaunchKernelGGL((k_zp_fixup<D, L>), dim3(nb, rows), dim3(256), 0, stream, P, out, out_row_stride,
                           freq_offset, fs_out);
    }
    template <class Loader>
    void convert(Loader ld, int rows, int64_t n, double *out, const double *freq_offset, double fs)
    {
        Scope s(*this, ST_CONVERT);
        hipLaunchKernelGGL((k_convert<Loader>), dim3((unsigned)((n + 255) / 256), rows), dim3(256), 0, stream, ld, n,
                           out, freq_offset, fs);
    }
    template <int D, int L>
    void power_fixup(const ZpParams &P, int rows, int64_t n, double *z, int sps, double *partials, int n_pblk)
    {
        Scope s(*this, ST_LPF_FIXUP);
        hipLaunchKernelGGL((k_power_fixup<D, L>), dim3((n_pblk + kPowSub - 1) / kPowSub, rows), dim3(kPowThreads), 0,
                           stream, P, n, z, sps, partials, n_pblk);
    }
    void finish(const FinishArgs &fa, int rows)
    {
        Scope s(*this, ST_FINISH);
        hipLaunchKernelGGL(k_finish, dim3(rows), dim3(kFinishThreads), 0, stream, fa);
    }
};

// centred root-raised-cosine taps, alpha 0.35, span 8 symbols, odd length, unit energy
// (the definition in oracle/tetra_np.py rrc_taps)
static std::vector<double> tetra_rrc_taps(double sps)
{
    const double alpha = 0.35;
    const int span = 8;
    const int half = (int)std::floor(span * sps / 2);
    std::vector<double> h(2 * half + 1);
    double e = 0;
    for (int i = -half; i <= half; ++i) {
        const double t = (double)i / sps;
        double v;
        if (std::fabs(t) < 1e-9)
            v = 1.0 - alpha + 4 * alpha / M_PI;
        else if (std::fabs(std::fabs(t) - 1.0 / (4 * alpha)) < 1e-9)
            v = (alpha / std::sqrt(2.0)) * ((1 + 2 / M_PI) * std::sin(M_PI / (4 * alpha)) + (1 - 2 / M_PI) * std::cos(M_PI / (4 * alpha)));
        else
            v = (std::sin(M_PI * t * (1 - alpha)) + 4 * alpha * t * std::cos(M_PI * t * (1 + alpha))) /
                (M_PI * t * (1 - (4 * alpha * t) * (4 * alpha * t)));
        h[i + half] = v;
        e += v * v;
    }
    for (auto &v : h) v /= std::sqrt(e);
    // 16-bit coefficients (oracle/tetra_np.py coeff16): each tap is the sum of two bfloat16, the way the kernel splits it
    // for the matrix cores, so that the split leaves no coefficient rounding behind
    auto bf16 = [](float x) {
        uint32_t u;
        std::memcpy(&u, &x, 4);
        u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
        std::memcpy(&x, &u, 4);
        return x;
    };
    for (auto &v : h) {
        const float f = (float)v, hi = bf16(f), lo = bf16(f - hi);
        v = (double)hi + (double)lo;
    }
    return h;
}

// ------------------------------------------------------------------------------------------
// plan.  A plan is made for (sample rate, wire format, carriers) and a chunk length; it serves any other chunk length
// through tdm_plan_resize: per length a VARIANT (the host plan + one small device allocation with the tables that depend
// on the length), while the large length-independent tables are uploaded once and the work buffers are shared and only
// ever grow.  Ragged reads (scanner.py:347, rtl_auto_capture.py:182) therefore cost a fraction of a millisecond the
// first time a length is seen and nothing afterwards.
// ------------------------------------------------------------------------------------------
struct Variant {
    RefPlanHost h;
    double *d_tab = nullptr;     // the length-dependent tables of every stage, one allocation
    ZpParams dec{}, lpf{}, dec_raw{};
    Lp2Params lp2{}, lp2_raw{};
    double *d_y = nullptr, *d_z = nullptr, *d_partials = nullptr;
    uint64_t stamp = 0;          // last use (eviction order)
    uint64_t id = 0;             // unique within the plan
    ~Variant() { if (d_tab) (void)hipFree(d_tab); }
};

constexpr size_t kMaxVariants = 32;

struct tdm_plan {
    int rows = 0, fmt = 0, mode = 0, device = 0;
    double sample_rate = 0.0;
    bool allow_raw = true;
    int32_t fast_pre_shift = 0;   // tdm_plan_option "fast_pre_shift"
    int32_t rows_per_chunk = 1;   // tdm_plan_option "rows_per_chunk"
    int64_t raw_min_blocks = 0;
    std::map<int64_t, std::unique_ptr<Variant>> variants;
    Variant *cur = nullptr;
    uint64_t clock = 0;
    std::map<const ZpSharedTables *, std::pair<std::shared_ptr<const ZpSharedTables>, double *>> d_shared;   // uploaded once per plan
    double *d_work = nullptr;    // work buffers of the stages, carved per variant, sized for the longest chunk so far
    size_t work_doubles = 0;
    const RefPlanHost &h() const { return cur->h; }
    // TETRA mode
    TetraParams tp{};
    uint32_t *d_tapops = nullptr;   // TetraParams::tap_ops
    float2 *d_gy = nullptr;         // TDM_MODE_TETRA_GARDNER, three-launch path only: matched-filter output (allocated the first time that path runs)
    int64_t gy_pitch = 0;
    int gardner_fused_ok = -1;      // does the fused Gardner kernel serve this plan (tap count, carriers, device)?  -1: not asked yet
    // two segments per carrier (GardnerSeg): geometry and temporaries, made the first time the plan runs that way
    int gardner_ff_first = 0;       // tdm_plan_option "gardner_ff_start"
    int gardner_ntaps_design = 0;   // the RRC filter's designed length (before the padding to an instantiated one)
    int gardner_seg = 0;            // pieces per carrier's chunk (TDM_MODE_TETRA_GARDNER: 1 whole chunks, 2 / 4 / 8: GardnerSeg)
    GardnerSeg gseg{};
    TetraParams gtp{};              // the plan's parameters with a half's length and row capacity
    float2 *d_gsoft = nullptr;      // [K or K - 1][rows][gtp.max_soft] the pieces' symbols
    int32_t *d_gint = nullptr;      // [4][K rows]: symbol counts, timing, seam indices (in, out)
    float *d_gts = nullptr;         // [2][K rows] seam instants (in, out)
    // staging for the host-pointer entry point
    void *d_iq = nullptr;
    size_t d_iq_bytes = 0;
    bool staging_ready = false;
    size_t staging_soft = 0;     // rows * max_soft the staging buffers were sized for
    double *d_pre = nullptr, *d_foff = nullptr, *d_soft = nullptr, *d_margin = nullptr;
    uint8_t *d_hard = nullptr;
    int32_t *d_nsoft = nullptr, *d_bp = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_order = nullptr;   // tdm_plan_wait_for: "everything enqueued on this plan's stream so far" (made on first use)
    StageTimer timer;
};

static inline size_t even(size_t x) { return (x + 1) & ~(size_t)1; }   // 16-byte granules

// work doubles of one zero-phase stage: y0, Ef, Eb, Gf, Hb, flast, Elast
static size_t zp_work_doubles(const ZpParams &p, int rows)
{
    const int D = p.nsec * p.K;
    const size_t n_y0 = even((size_t)rows * p.n_out * 2), n_e = (size_t)rows * p.nb * D * 2;
    return n_y0 + 4 * n_e + (size_t)rows * 2 + (size_t)rows * D * 2;
}
static double *zp_bind_work(ZpParams &p, int rows, double *w)
{
    const int D = p.nsec * p.K;
    const size_t n_y0 = even((size_t)rows * p.n_out * 2), n_e = (size_t)rows * p.nb * D * 2;
    p.y0 = w;
    p.Ef = p.y0 + n_y0;
    p.Eb = p.Ef + n_e;
    p.Gf = p.Eb + n_e;
    p.Hb = p.Gf + n_e;
    p.flast = p.Hb + n_e;
    p.Elast = p.flast + (size_t)rows * 2;
    return w + zp_work_doubles(p, rows);
}

// the shared (length-independent) tables of a stage on the device, uploaded the first time the plan sees them
static int shared_on_device(tdm_plan *plan, const std::shared_ptr<const ZpSharedTables> &sh, const double **out)
{
    *out = nullptr;
    if (!sh) return TDM_OK;
    auto it = plan->d_shared.find(sh.get());
    if (it == plan->d_shared.end()) {
        double *d = nullptr;
        HIP_TRY(hipMalloc(&d, sh->blob.size() * sizeof(double)));
        hipError_t e = hipMemcpy(d, sh->blob.data(), sh->blob.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); return fail(TDM_ERR_HIP, std::string("hipMemcpy(shared tables): ") + hipGetErrorString(e)); }
        it = plan->d_shared.emplace(sh.get(), std::make_pair(sh, d)).first;
    }
    *out = it->second.second;
    return TDM_OK;
}

// Make the variant for chunk length n current (reference mode).
// shared tables no variant refers to any more (only this map's own reference is left) are released; called where the
// plan's stream has just been synchronised.  The variant under construction holds its tables through its RefPlanHost.
static void shared_gc(tdm_plan *plan)
{
    for (auto it = plan->d_shared.begin(); it != plan->d_shared.end();) {
        if (it->second.first.use_count() == 1) {
            (void)hipFree(it->second.second);
            it = plan->d_shared.erase(it);
        } else {
            ++it;
        }
    }
}

static int plan_select(tdm_plan *plan, double sample_rate, int64_t n)
{
    auto hit = plan->variants.find(n);
    if (hit != plan->variants.end()) {
        plan->cur = hit->second.get();
        plan->cur->stamp = ++plan->clock;
        return TDM_OK;
    }
    const int rows = plan->rows;
    std::unique_ptr<Variant> v(new Variant);
    // designs and length-independent tables come from a variant that has them: the current one, else any other (a short
    // chunk that skips the decimator must not make the next long one rebuild and re-upload the shared tables)
    const RefPlanHost *base = plan->cur ? &plan->cur->h : nullptr;
    if (!base || !base->dec.shared)
        for (auto &kv : plan->variants)
            if (kv.second->h.dec.shared) { base = &kv.second->h; break; }
    v->h = build_ref_plan(sample_rate, n, 25000.0, true, plan->allow_raw ? plan->fmt : -1, base);
    const RefPlanHost &h = v->h;
    // ---- tables: the length-dependent ones of all stages in one allocation and one copy
    const bool use_lpf = h.lpf && !h.lp2.ok;
    std::vector<double> tab;
    auto put = [&](const std::vector<double> &src) { const size_t o = tab.size(); tab.insert(tab.end(), src.begin(), src.end()); if (tab.size() & 1) tab.push_back(0.0); return o; };
    const size_t o_dec = h.decimated ? put(h.dec.blob) : 0;
    const size_t o_lpf = use_lpf ? put(h.lpf_t.blob) : 0;
    const size_t o_raw = h.raw_S ? put(h.dec_raw.blob) : 0;
    const size_t o_lm = h.lp2.ok ? put(h.lp2.lane_m) : 0, o_lc = h.lp2.ok ? put(h.lp2.cst) : 0, o_ls = h.lp2.ok ? put(h.lp2.seeds) : 0;
    const size_t o_rc = h.raw_S ? put(h.lp2_raw.cst) : 0, o_rs = h.raw_S ? put(h.lp2_raw.seeds) : 0;
    const size_t o_li = h.lp2.ok ? put(h.lp2.items) : 0, o_ri = h.raw_S ? put(h.lp2_raw.items) : 0;
    if (!tab.empty()) {
        HIP_TRY(hipMalloc(&v->d_tab, tab.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(v->d_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    int rc;
    const double *sb = nullptr;
    if (h.decimated) {
        if ((rc = shared_on_device(plan, h.dec.shared, &sb))) return rc;
        v->dec = h.dec.p;
        h.dec.bind(v->dec, v->d_tab + o_dec, sb);
    }
    if (use_lpf) {
        v->lpf = h.lpf_t.p;
        h.lpf_t.bind(v->lpf, v->d_tab + o_lpf);
    }
    if (h.raw_S) {
        if ((rc = shared_on_device(plan, h.dec_raw.shared, &sb))) return rc;
        v->dec_raw = h.dec_raw.p;
        h.dec_raw.bind(v->dec_raw, v->d_tab + o_raw, sb);
    }
    // ---- work: carved out of one buffer that only grows
    // y: the low-rate signal when nothing downstream forms it on the fly (no decimation, or no channel filter);
    // z / partials: only the cascade-engine fallback of the low-rate stage materialises them
    const size_t nd = even((size_t)rows * h.n_dec * 2);
    const bool need_y = !h.decimated || !h.lpf;
    const bool need_z = !h.lp2.ok && h.lpf && !(h.sps > 1 && h.sps <= kMaxSps);
    const size_t n_part = !h.lp2.ok ? even((size_t)rows * (h.n_dec / kPowThreads + 16) * kMaxSps) : 0;
    const size_t n_zt = h.lp2.ok ? even((size_t)rows * h.sps * h.lp2.p.zt_k * 2) : 0;
    const int max_chunks = h.raw_S && h.lp2_raw.p.n_chunks > h.lp2.p.n_chunks ? h.lp2_raw.p.n_chunks : h.lp2.p.n_chunks;
    const size_t n_lp2p = h.lp2.ok ? even((size_t)rows * max_chunks * kMaxSps) : 0;
    const size_t need = (h.decimated ? zp_work_doubles(v->dec, rows) : 0) + (use_lpf ? zp_work_doubles(v->lpf, rows) : 0) +
                        (h.raw_S ? zp_work_doubles(v->dec_raw, rows) : 0) + n_zt + n_lp2p + (need_y ? nd : 0) + (need_z ? nd : 0) + n_part;
    if (need > plan->work_doubles) {
        // the new buffer first: if the allocation fails the plan keeps serving the lengths it has.  Then: kernels of
        // earlier calls may still be using the old buffer, and the other variants' pointers into it go stale
        double *grown = nullptr;
        if (hipMalloc(&grown, need * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            return fail(TDM_ERR_NOMEM, "plan work buffer: hipMalloc of " + std::to_string(need * sizeof(double)) + " bytes failed");
        }
        if (hipStreamSynchronize(plan->stream) != hipSuccess) {
            (void)hipFree(grown);
            return fail(TDM_ERR_HIP, "hipStreamSynchronize(plan->stream)");
        }
        plan->cur = nullptr;
        plan->variants.clear();
        shared_gc(plan);
        if (plan->d_work) (void)hipFree(plan->d_work);
        plan->d_work = grown;
        plan->work_doubles = need;
    }
    double *w = plan->d_work;
    if (h.decimated) w = zp_bind_work(v->dec, rows, w);
    if (use_lpf) w = zp_bind_work(v->lpf, rows, w);
    if (h.raw_S) w = zp_bind_work(v->dec_raw, rows, w);
    if (h.lp2.ok) {
        v->lp2 = h.lp2.p;
        v->lp2.zt = w; w += n_zt;
        v->lp2.partials = w; w += n_lp2p;
        v->lp2.lane_m = v->d_tab + o_lm;
        v->lp2.cst = v->d_tab + o_lc;
        if (!h.lp2.seeds.empty()) v->lp2.seeds = v->d_tab + o_ls;
        if (!h.lp2.items.empty()) v->lp2.items = (const int32_t *)(v->d_tab + o_li);
        if (h.raw_S) {
            // the raw-integer decimator's geometry: own tables and work buffers, same low-rate outputs
            // (the edge constants depend on the lane grid's offset, which follows the decimator's block geometry)
            v->lp2_raw = h.lp2_raw.p;
            v->lp2_raw.zt = v->lp2.zt;
            v->lp2_raw.partials = v->lp2.partials;
            v->lp2_raw.lane_m = v->lp2.lane_m;
            v->lp2_raw.cst = v->d_tab + o_rc;
            v->lp2_raw.seeds = v->d_tab + o_rs;
            v->lp2_raw.items = (const int32_t *)(v->d_tab + o_ri);
        }
    }
    if (need_y) { v->d_y = w; w += nd; }
    if (need_z) { v->d_z = w; w += nd; }
    if (n_part) { v->d_partials = w; w += n_part; }
    // ---- keep it
    if (plan->variants.size() >= kMaxVariants) {
        auto old = plan->variants.begin();
        for (auto it = plan->variants.begin(); it != plan->variants.end(); ++it)
            if (it->second->stamp < old->second->stamp) old = it;
        // (its tables may still be read by kernels in flight)
        HIP_TRY(hipStreamSynchronize(plan->stream));
        plan->variants.erase(old);
        shared_gc(plan);
    }
    v->stamp = ++plan->clock;
    v->id = plan->clock;
    v->h.raw_min_blocks = plan->raw_min_blocks;
    plan->cur = v.get();
    plan->variants[n] = std::move(v);
    return TDM_OK;
}

static size_t fmt_bytes(int fmt) { return fmt == TDM_CU8 || fmt == TDM_CS8 ? 2 : (fmt == TDM_CF32 ? 8 : 16); }
static int tetra_fmt8(int fmt) { return fmt == TDM_CU8 ? 1 : (fmt == TDM_CS8 ? 2 : 0); }   // the TETRA-mode kernels' FMT8

static void sync_scratch_release(int device, hipStream_t st);   // (find_sync's per-stream scratch, below)

// Geometry of a chunk walked in K pieces (oracle/tetra_np.py gardner_segments is the same arithmetic): false when the chunk is
// too short (a piece's own part, n / K, under 1.9 warm-ups)
struct GardnerGeom {
    int margin, lead, n_v, step, seam_in, seam_out;
};
static bool gardner_geometry(int64_t n, double sps, int ntaps_design, int K, GardnerGeom *g)
{
    const int warm = 384;             // (oracle/tetra_np.py GARDNER_WARMUP_SYMBOLS)
    g->margin = (ntaps_design - 1) / 2 + 4 * (int)std::ceil(sps) + 8;
    g->lead = (int)std::ceil(warm * sps) + g->margin;
    if (K < 2 || 10 * n < 19 * (int64_t)K * g->lead) return false;
    const int n_v0 = (int)((n + (int64_t)(K - 1) * g->lead + K - 1) / K);
    g->step = (int)((n - n_v0) / (K - 1));
    g->n_v = (int)(n - (int64_t)(K - 1) * g->step);
    g->seam_out = g->n_v - g->margin;
    g->seam_in = g->seam_out - g->step;
    return true;
}

// TDM_MODE_TETRA_GARDNER: K pieces per carrier's chunk (K independently started loops joined at seams).  The DEFAULT depends
// on the chunk alone -- its length, rate and tap count: the largest power of two up to 8 whose pieces leave every loop its
// 384 warm-up symbols (gardner_geometry) -- so a carrier's symbols do not depend on how many other carriers share the plan or
// on the device's size (round-5 review: the default used to be fitted to the batch); a batch whose K x rows / 16 workgroups
// are not all resident at once runs in more than one round of them.  The tap counts above 41 (one 79 KB workgroup per
// compute unit: the fused kernel only serves a launch of one round there) keep whole chunks by default.
// allow (tdm_plan_option / tdm_debug_set "gardner_segments"): 0 whole chunks; 1 the default above; K = 2, 4, 8 at most K
// pieces (the chunk's rule, capped); -1 FITTED TO THE BATCH, the round-5 rule: the power of two up to 8 with the shortest
// pieces -- a piece's time is its length, times 1.18 when two loops share a compute unit (measured) -- among those whose
// workgroups are all resident at once: the fastest for this plan's row count on this device, and the one setting under
// which the same carrier gives (slightly) different soft symbols behind a seam in plans of different size.
// Called when the plan is made and when the option changes (the device idle): sets plan->gardner_seg, the geometry and the
// temporaries.
static int gardner_choose_pieces(tdm_plan *p, long long allow)
{
    const TetraParams &tp = p->tp;
    for (void **q : {(void **)&p->d_gsoft, (void **)&p->d_gint, (void **)&p->d_gts}) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    p->gseg = GardnerSeg{};
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device);
    const int per_cu = tetra_gardner_fused_per_cu(tp.ntaps);   // (0: no fused kernel for this tap count)
    int best_k = 1;
    GardnerGeom best{};
    if (allow == -1) {
        const int per_cu_ok = p->gardner_fused_ok == 1 ? per_cu : 0;
        double best_cost = (double)tp.n * ((int64_t)(p->rows + 15) / 16 > cus ? 1.18 : 1.0);
        for (int K = 2; K <= 8 && per_cu_ok >= 1; K *= 2) {
            const int64_t wgs = ((int64_t)K * p->rows + 15) / 16;
            if (wgs > (int64_t)cus * (per_cu_ok >= 2 ? 2 : 1)) break;            // (all pieces' workgroups resident at once)
            GardnerGeom g;
            if (!gardner_geometry(tp.n, tp.sps, p->gardner_ntaps_design, K, &g)) break;
            const double cost = (double)g.n_v * (wgs > cus ? 1.18 : 1.0);
            if (cost < 0.9 * best_cost) { best_cost = cost; best_k = K; best = g; }
        }
    } else if (allow != 0 && per_cu >= 2 && p->gardner_fused_ok == 1 && debug_value("gardner_fused") != 0) {   // (fused_ok: with per_cu >= 2 it depends on the tap count and the wire format, not on the row count)
        for (int K = 2; K <= 8 && (allow == 1 || K <= allow); K *= 2) {
            GardnerGeom g;
            if (!gardner_geometry(tp.n, tp.sps, p->gardner_ntaps_design, K, &g)) break;
            best_k = K;
            best = g;
        }
    }
    p->gardner_seg = 1;   // (whole chunks unless everything below succeeds: a failed allocation leaves a plan that works)
    if (best_k > 1) {
        const int R = p->rows, K = best_k, n_v = best.n_v;
        GardnerSeg &S = p->gseg;
        S.rows_phys = R;
        S.pieces = K;
        S.seg_step = best.step;
        S.seam_out = best.seam_out;
        S.seam_in = best.seam_in;
        S.piece_mid = K / 2 - 1;
        S.k_mid = (int)((0.5 * (double)tp.n - (double)S.piece_mid * (double)best.step) / tp.sps);
        p->gtp = tp;
        p->gtp.n = n_v;
        p->gtp.max_soft = (int32_t)(1.02 * (double)n_v / tp.sps) + 8;
        const bool direct = R % 16 == 0;   // (piece 0 straight into the caller's rows: GardnerSeg::soft_a)
        S.pitch_a = direct ? tp.max_soft : 0;
        if (hipMalloc((void **)&p->d_gsoft, (size_t)(direct ? K - 1 : K) * R * p->gtp.max_soft * sizeof(float2)) != hipSuccess ||
            hipMalloc((void **)&p->d_gint, (size_t)4 * K * R * sizeof(int32_t)) != hipSuccess ||
            hipMalloc((void **)&p->d_gts, (size_t)2 * K * R * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            for (void **q : {(void **)&p->d_gsoft, (void **)&p->d_gint, (void **)&p->d_gts}) {
                if (*q) (void)hipFree(*q);
                *q = nullptr;
            }
            p->gseg = GardnerSeg{};
            return fail(TDM_ERR_NOMEM, "TETRA_GARDNER plan: no memory for the temporaries of " + std::to_string(K) + " pieces per chunk (the plan walks whole chunks)");
        }
        S.k_in = p->d_gint + (size_t)2 * K * R;
        S.k_out = p->d_gint + (size_t)3 * K * R;
        S.t_in = p->d_gts;
        S.t_out = p->d_gts + (size_t)K * R;
        p->gardner_seg = K;
    }
    return TDM_OK;
}

static void plan_free(tdm_plan *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->stream) sync_scratch_release(p->device, p->stream);   // the stream goes away: so does the scratch keyed by it
    p->cur = nullptr;
    p->variants.clear();
    for (auto &kv : p->d_shared)
        if (kv.second.second) (void)hipFree(kv.second.second);
    void *ptrs[] = {p->d_work, p->d_gsoft, p->d_gint, p->d_gts, p->d_iq, p->d_pre, p->d_foff, p->d_soft, p->d_margin, p->d_hard, p->d_nsoft, p->d_bp, p->d_tapops, p->d_gy};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->ev_order) (void)hipEventDestroy(p->ev_order);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

extern "C" {

int tdm_version(void) { return TDM_VERSION; }

int tdm_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(TDM_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

int tdm_last_error(char *buf, size_t buflen)
{
    if (buf && buflen) {
        std::snprintf(buf, buflen, "%s", g_err.c_str());
    }
    return (int)g_err.size();
}

int tdm_debug_set(const char *key, int64_t value)
{
    DebugSwitch *d = debug_find(key);
    if (!d) return fail(TDM_ERR_INVALID, std::string("tdm_debug_set: unknown switch '") + (key ? key : "(null)") + "'");
    d->value.store(value, std::memory_order_relaxed);
    return TDM_OK;
}

int tdm_debug_get(const char *key, int64_t *value)
{
    DebugSwitch *d = debug_find(key);
    if (!d || !value) return fail(TDM_ERR_INVALID, std::string("tdm_debug_get: unknown switch '") + (key ? key : "(null)") + "'");
    *value = d->value.load(std::memory_order_relaxed);
    return TDM_OK;
}

int tdm_plan_create(double sample_rate, int64_t n_samples, int32_t n_carriers, int32_t in_fmt, int32_t mode,
                    int32_t device, tdm_plan **out)
{
    if (!out) return fail(TDM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (!(sample_rate > 0) || n_samples < 1 || n_samples > (int64_t(1) << 31) || n_carriers < 1 || n_carriers > 65535 ||
        in_fmt < 0 || in_fmt > 3)
        return fail(TDM_ERR_INVALID, "bad sample_rate / n_samples / n_carriers (1..65535) / in_fmt");
    if (mode != TDM_MODE_REFERENCE && mode != TDM_MODE_TETRA && mode != TDM_MODE_TETRA_GARDNER) return fail(TDM_ERR_INVALID, "bad mode");
    int rc = use_device(device);
    if (rc) return rc;
    std::unique_ptr<tdm_plan, void (*)(tdm_plan *)> p(new tdm_plan, plan_free);
    p->device = device;
    p->rows = n_carriers;
    p->fmt = in_fmt;
    p->mode = mode;
    if (mode == TDM_MODE_TETRA || mode == TDM_MODE_TETRA_GARDNER) {
        // channelised baseband in: sample_rate is the per-carrier rate, >= 2 samples per symbol
        // channelised baseband as cf32 (the channeliser's output), or straight off the wire as cu8 / cs8 (round 6: converted where
        // the kernels stage their window; one bf16 plane per component in the fused receiver)
        if (in_fmt == TDM_CF64) return fail(TDM_ERR_UNSUPPORTED, "TETRA mode takes cf32, cu8 or cs8 baseband");
        const double sps = sample_rate / kSymbolRate;
        if (sps < 2.0 || sps > 8.0) return fail(TDM_ERR_UNSUPPORTED, "TETRA mode needs 2..8 samples per symbol");
        if (n_samples < 64 || n_samples > (int64_t)kMaxTimingBlocks * kTimingBlock)
            return fail(TDM_ERR_UNSUPPORTED, "TETRA mode chunk length must be 64..131072 samples");
        std::vector<double> h = tetra_rrc_taps(sps);
        if ((int)h.size() > kRrcMaxTaps) return fail(TDM_ERR_UNSUPPORTED, "too many RRC taps");
        const int ntaps_design = (int)h.size();   // (before the padding to an instantiated length)
        {
            // the matched-filter kernel is instantiated for these (odd) lengths: centre the taps in the next one up,
            // zeros either side (same filter, same alignment), so that every rate in the 2..8 samples/symbol contract runs
            static const int kInst[] = {17, 25, 33, 35, 41, 49, 57, 65};
            int nt = 0;
            for (int c : kInst)
                if (c >= (int)h.size()) { nt = c; break; }
            if (!nt) return fail(TDM_ERR_UNSUPPORTED, "no RRC kernel for this tap count");
            const int pad = (nt - (int)h.size()) / 2;
            std::vector<double> hp((size_t)nt, 0.0);
            for (size_t i = 0; i < h.size(); ++i) hp[i + pad] = h[i];
            h.swap(hp);
        }
        TetraParams &tp = p->tp;
        tp.n = (int32_t)n_samples;
        tp.ntaps = (int32_t)h.size();
        tp.sps = sps;
        tp.inv_sps = 1.0 / sps;
        // symbol-clock phasors exp(-2 pi i g / sps) at a lane's eight outputs of a tile (g = 256 (v >> 2) + 16 (v & 3)) and
        // at g = one tile (argument reduced before the call)
        auto clock = [&](double g, float &c, float &s_) {
            const double ph = g / sps, fr = ph - std::floor(ph);
            c = (float)std::cos(-2.0 * M_PI * fr);
            s_ = (float)std::sin(-2.0 * M_PI * fr);
        };
        for (int v = 0; v < kRrcPerThread; ++v) clock((double)(256 * (v >> 2) + kRrcRun * (v & 3)), tp.ev_c[v], tp.ev_s[v]);
        clock((double)kRrcTile, tp.tile_c, tp.tile_s);
        // capacity of a carrier's output row: the feed-forward receiver emits the nominal count; the Gardner loop follows the
        // carrier's own symbol clock, so its rows leave room for a clock 2 % fast
        tp.max_soft = mode == TDM_MODE_TETRA_GARDNER ? (int32_t)(1.02 * (double)n_samples / sps) + 8 : (int32_t)(n_samples / sps) + 4;
        for (size_t i = 0; i < h.size(); ++i) tp.taps[i] = (float)h[i];
        {
            // the matched filter's constant operands, lane by lane (every carrier of every launch reads the same 4-6 KB)
            std::vector<uint32_t> ops(tetra_tap_operand_words(tp.ntaps));
            tetra_tap_operands(tp.taps, tp.ntaps, ops.data());
            HIP_TRY(hipMalloc((void **)&p->d_tapops, ops.size() * sizeof(uint32_t)));
            HIP_TRY(hipMemcpy(p->d_tapops, ops.data(), ops.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            tp.tap_ops = p->d_tapops;
        }
        {
            std::unique_ptr<Variant> v(new Variant);
            v->h.sample_rate = sample_rate;
            v->h.n = n_samples;
            v->h.n_dec = n_samples;
            v->h.rate_dec = sample_rate;
            v->h.sps = (int)sps;
            v->h.max_soft = tp.max_soft;
            v->h.lpf = true;
            p->cur = v.get();
            p->variants[n_samples] = std::move(v);
        }
        if (mode == TDM_MODE_TETRA_GARDNER) {
            p->rows = n_carriers;
            p->device = device;
            // (8-bit input: the fused kernel is instantiated for 33 and 35 taps; other tap counts take the three launches, whose
            //  matched filter converts)
            p->gardner_fused_ok = (debug_value("gardner_fused") != 0 && tetra_gardner_fused_available(tp.ntaps, n_carriers, tetra_fmt8(in_fmt))) ? 1 : 0;
            p->gardner_ntaps_design = ntaps_design;
            {
                // (no memory for the pieces' temporaries: the plan is made all the same and walks whole chunks)
                const int rc = gardner_choose_pieces(p.get(), debug_value("gardner_segments"));
                if (rc != TDM_OK && rc != TDM_ERR_NOMEM) return rc;
            }
            // the matched-filter output of the three-launch path: [rows][pitch] cf32, rows 16-byte aligned; about 1 GB at 4096 x 32 768, so only a plan that ever takes the
            // three launches allocates it (the default fused kernel keeps the filter output in LDS)
            p->gy_pitch = (n_samples + 1) & ~(int64_t)1;
        }
        HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&p->ev0));
        HIP_TRY(hipEventCreate(&p->ev1));
        *out = p.release();
        return TDM_OK;
    }
    // (tdm_debug_set("no_raw", 1): experiments / tests keep cu8 plans on the kernel that holds its samples as doubles)
    p->allow_raw = debug_value("no_raw") != 1;
    {
        // blocks of the double-based decimator below which a batch stays on it: two wavefronts per SIMD of the device
        // (tdm_debug_set("raw_min_blocks", n) overrides; tests use 0 to put single carriers on the raw-integer kernel)
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        p->raw_min_blocks = (int64_t)prop.multiProcessorCount * 8;
        if (debug_value("raw_min_blocks") >= 0) p->raw_min_blocks = debug_value("raw_min_blocks");
    }
    HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&p->ev0));
    HIP_TRY(hipEventCreate(&p->ev1));
    p->sample_rate = sample_rate;
    if ((rc = plan_select(p.get(), sample_rate, n_samples))) return rc;
    *out = p.release();
    return TDM_OK;
}

// Serve another chunk length with the same plan (reference mode).  The first call for a length builds its small tables
// (a fraction of a millisecond, one allocation, one copy); a length seen before costs a map look-up.  Work buffers are
// shared between lengths and grow when a longer chunk arrives.  Not to be called while another thread is inside a
// process call of the same plan; kernels already enqueued on the plan's stream are unaffected.
int tdm_plan_resize(tdm_plan *plan, int64_t n_samples)
{
    if (!plan) return fail(TDM_ERR_INVALID, "null plan");
    if (plan->mode != TDM_MODE_REFERENCE) return fail(TDM_ERR_UNSUPPORTED, "TETRA-mode plans have one chunk length");
    if (n_samples < 1 || n_samples > (int64_t(1) << 31)) return fail(TDM_ERR_INVALID, "bad n_samples");
    if (plan->cur && plan->cur->h.n == n_samples) return TDM_OK;
    HIP_TRY(hipSetDevice(plan->device));
    return plan_select(plan, plan->sample_rate, n_samples);
}

int tdm_plan_option(tdm_plan *plan, const char *key, int64_t value)
{
    if (!plan || !key) return fail(TDM_ERR_INVALID, "null argument");
    if (std::strcmp(key, "fast_pre_shift") == 0) {
        if (plan->mode != TDM_MODE_REFERENCE) return fail(TDM_ERR_UNSUPPORTED, "fast_pre_shift is a reference-mode option");
        plan->fast_pre_shift = value ? 1 : 0;
        return TDM_OK;
    }
    if (std::strcmp(key, "rows_per_chunk") == 0) {
        if (plan->mode != TDM_MODE_REFERENCE) return fail(TDM_ERR_UNSUPPORTED, "rows_per_chunk is a reference-mode option");
        if (value < 1 || plan->rows % value != 0) return fail(TDM_ERR_INVALID, "rows_per_chunk: a divisor of the plan's carrier count (1 = every row its own input row)");
        plan->rows_per_chunk = (int32_t)value;
        return TDM_OK;
    }
    if (std::strcmp(key, "gardner_ff_start") == 0) {
        if (plan->mode != TDM_MODE_TETRA_GARDNER) return fail(TDM_ERR_UNSUPPORTED, "gardner_ff_start is an option of TDM_MODE_TETRA_GARDNER plans");
        if (value && plan->gardner_fused_ok != 1) return fail(TDM_ERR_UNSUPPORTED, "gardner_ff_start needs the fused Gardner kernel, which does not serve this plan");
        plan->gardner_ff_first = value ? 1 : 0;
        return TDM_OK;
    }
    if (std::strcmp(key, "gardner_segments") == 0) {
        if (plan->mode != TDM_MODE_TETRA_GARDNER) return fail(TDM_ERR_UNSUPPORTED, "gardner_segments is an option of TDM_MODE_TETRA_GARDNER plans");
        if (value < -1 || value > 8) return fail(TDM_ERR_INVALID, "gardner_segments: 0 (whole chunks), 1 (the default: from the chunk alone), the largest number of pieces allowed (2..8), or -1 (fitted to this plan's batch and device)");
        HIP_TRY(hipSetDevice(plan->device));
        // calls in flight use the temporaries -- on the plan's stream or on one the caller handed to tdm_process_device
        HIP_TRY(hipDeviceSynchronize());
        return gardner_choose_pieces(plan, value);
    }
    return fail(TDM_ERR_INVALID, std::string("tdm_plan_option: unknown option '") + key + "'");
}

int tdm_gardner_geometry(double sample_rate, int64_t n_samples, int32_t pieces, int32_t *out)
{
    if (!out) return fail(TDM_ERR_INVALID, "null output");
    const double sps = sample_rate / kSymbolRate;
    if (!(sps >= 2.0 && sps <= 8.0) || n_samples < 1) return fail(TDM_ERR_UNSUPPORTED, "TETRA mode needs 2..8 samples per symbol");
    GardnerGeom g;
    if (!gardner_geometry(n_samples, sps, (int)tetra_rrc_taps(sps).size(), pieces, &g))
        return fail(TDM_ERR_UNSUPPORTED, "the chunk is too short for this many pieces");
    out[0] = g.n_v; out[1] = g.step; out[2] = g.seam_in; out[3] = g.seam_out; out[4] = g.margin; out[5] = g.lead;
    return TDM_OK;
}

int tdm_plan_destroy(tdm_plan *plan)
{
    plan_free(plan);
    return TDM_OK;
}

int tdm_plan_get_info(const tdm_plan *plan, tdm_plan_info *info)
{
    if (!plan || !info) return fail(TDM_ERR_INVALID, "null argument");
    if (!plan->cur) return fail(TDM_ERR_INVALID, "plan has no current length");
    const RefPlanHost &h = plan->h();
    std::memset(info, 0, sizeof(*info));
    info->sample_rate = h.sample_rate;
    info->rate_dec = h.rate_dec;
    info->n_samples = h.n;
    info->n_dec = h.n_dec;
    info->n_carriers = plan->rows;
    info->q = h.decimated ? h.q : 1;
    info->sps = h.sps;
    info->phase_step = h.phase_step;
    info->max_soft = (int32_t)h.max_soft;
    info->lpf_applied = h.lpf ? 1 : 0;
    info->in_fmt = plan->fmt;
    info->mode = plan->mode;
    info->device = plan->device;
    // (pieces are the fused kernel's: with tdm_debug_set("gardner_fused", 0) flipped after the plan was made the three launches
    //  walk whole chunks)
    info->gardner_segments = plan->mode == TDM_MODE_TETRA_GARDNER ? ((debug_value("gardner_fused") != 0 && plan->gardner_fused_ok) ? plan->gardner_seg : 1) : 0;
    if (plan->mode == TDM_MODE_REFERENCE && h.decimated) {
        // (the rule of run_ref_fmt; a call with an input-rate pre-shift stays on the double-based kernel)
        const bool raw = h.raw_S > 0 && plan->fmt == TDM_CU8 && (int64_t)plan->rows * h.dec.p.nb >= h.raw_min_blocks;
        info->dec_engine = raw ? 3 : (h.pz_S ? 2 : 1);
    }
    return TDM_OK;
}

static int process_device_impl(tdm_plan *plan, const void *iq, int64_t carrier_stride_samples, const double *pre_shift_hz,
                               const double *freq_offset_hz, uint8_t *hard, double *soft, int32_t *n_soft,
                               int32_t *best_phase, double *min_margin, void *stream, const int32_t *row_list, const int32_t *n_rows)
{
    if (!plan || !iq || !hard || !soft || !n_soft) return fail(TDM_ERR_INVALID, "null argument");
    if (row_list && plan->mode != TDM_MODE_TETRA)
        return fail(TDM_ERR_UNSUPPORTED, "a row list is taken by TDM_MODE_TETRA plans (the feed-forward receiver: one workgroup per carrier)");
    if (carrier_stride_samples < 0) return fail(TDM_ERR_INVALID, "negative carrier stride");
    HIP_TRY(hipSetDevice(plan->device));
    HipBackend be;
    be.stream = stream ? (hipStream_t)stream : plan->stream;
    be.timer = &plan->timer;
    be.device = plan->device;
    if (plan->mode == TDM_MODE_TETRA || plan->mode == TDM_MODE_TETRA_GARDNER) {
        if (pre_shift_hz || freq_offset_hz)
            return fail(TDM_ERR_UNSUPPORTED, "TETRA mode: carrier offsets are estimated, not supplied");
        if (carrier_stride_samples < plan->tp.n)
            return fail(TDM_ERR_INVALID, "TETRA mode: carrier stride shorter than the chunk");
        const TetraParams &tp = plan->tp;
        if (plan->mode == TDM_MODE_TETRA_GARDNER) {
            // matched filter -> HBM -> Gardner loop, one lane per carrier -> decisions (tetra_gardner_kernels.hpp)
            // the matched filter and the loop in ONE kernel (the filter output stays in LDS); tdm_debug_set("gardner_fused", 0):
            // the three launches with the filter output in HBM
            constexpr int stages = 7;
            const bool fused = debug_value("gardner_fused") != 0;
            bool fused_done = false;
            // (whether the fused kernel serves this plan, and in how many pieces per chunk, was settled when the plan was made)
            if (fused && plan->gardner_fused_ok && plan->gardner_seg > 1) {
                const int R = plan->rows, K = plan->gardner_seg;
                GardnerSeg S = plan->gseg;
                S.soft_a = S.pitch_a ? (float2 *)soft : nullptr;
                S.ff_first = plan->gardner_ff_first;
                {
                    HipBackend::Scope s(be, ST_TETRA_LOOP);
                    fused_done = tetra_gardner_fused_launch(plan->gtp, K * R, iq, tetra_fmt8(plan->fmt), carrier_stride_samples, plan->d_gsoft,
                                                            plan->d_gint, plan->d_gint + (size_t)K * R, be.stream, &S);
                }
                if (fused_done) {
                    // decisions, with the pieces joined first (pieces 1..: behind piece 0's rows in the temporary unless those
                    // went straight to the caller's rows)
                    HipBackend::Scope s(be, ST_TETRA_DECIDE);
                    if (!S.soft_a)   // (carrier counts that are no multiple of sixteen: piece 0 comes out of the temporary too)
                        HIP_TRY(hipMemcpy2DAsync(soft, (size_t)tp.max_soft * sizeof(float2), plan->d_gsoft, (size_t)plan->gtp.max_soft * sizeof(float2),
                                                 (size_t)std::min(plan->gtp.max_soft, tp.max_soft) * sizeof(float2), R, hipMemcpyDeviceToDevice, be.stream));
                    tetra_decide_launch(tp, R, (float2 *)soft, n_soft, hard, min_margin, be.stream, &S,
                                        plan->d_gsoft + (S.soft_a ? 0 : (size_t)R * plan->gtp.max_soft), plan->gtp.max_soft, plan->d_gint,
                                        plan->d_gint + (size_t)K * R, best_phase);
                    if (be.err != hipSuccess) return fail(TDM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(be.err));
                    return TDM_OK;
                }
            } else if (fused && plan->gardner_fused_ok) {
                HipBackend::Scope s(be, ST_TETRA_LOOP);
                GardnerSeg S{};
                S.ff_first = plan->gardner_ff_first;
                fused_done = tetra_gardner_fused_launch(tp, plan->rows, iq, tetra_fmt8(plan->fmt), carrier_stride_samples, (float2 *)soft, n_soft, best_phase, be.stream,
                                                        S.ff_first ? &S : nullptr);
            }
            if (plan->gardner_ff_first && !fused_done)
                return fail(TDM_ERR_UNSUPPORTED, "gardner_ff_start needs the fused Gardner kernel (gardner_fused switched off, or no kernel for this tap count / batch)");
            const bool three = !fused_done;   // (gardner_fused = 0, no fused kernel for this tap count, or too many carriers for it)
            if (three && !plan->d_gy)
                HIP_TRY(hipMalloc((void **)&plan->d_gy, (size_t)plan->rows * plan->gy_pitch * sizeof(float2)));
            if (three && (stages & 1)) {
                HipBackend::Scope s(be, ST_TETRA_MF);
                if (!tetra_mf_launch(tp, plan->rows, iq, tetra_fmt8(plan->fmt), carrier_stride_samples, plan->d_gy, plan->gy_pitch, be.stream))
                    return fail(TDM_ERR_UNSUPPORTED, "no RRC kernel instantiated for this tap count");
            }
            if (three && (stages & 2)) {
                HipBackend::Scope s(be, ST_TETRA_LOOP);
                tetra_gardner_loop_launch(tp, plan->rows, plan->d_gy, plan->gy_pitch, (float2 *)soft, n_soft, best_phase, be.stream);
            }
            if (stages & 4) {
                HipBackend::Scope s(be, ST_TETRA_DECIDE);
                tetra_decide_launch(tp, plan->rows, (float2 *)soft, n_soft, hard, min_margin, be.stream);
            }
            if (be.err != hipSuccess) return fail(TDM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(be.err));
            return TDM_OK;
        }
        {
            // one kernel: matched filter, timing, Farrow, carrier-offset estimate and decisions; one workgroup per carrier
            HipBackend::Scope s(be, ST_TETRA);
            if (!tetra_launch(tp, plan->rows, iq, tetra_fmt8(plan->fmt), carrier_stride_samples, (float2 *)soft, hard, n_soft, best_phase,
                              min_margin, be.stream, row_list, n_rows))
                return fail(TDM_ERR_UNSUPPORTED, "no RRC kernel instantiated for this tap count");
        }
        if (be.err != hipSuccess) return fail(TDM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(be.err));
        return TDM_OK;
    }
    if (!plan->cur) return fail(TDM_ERR_INVALID, "plan has no current length");
    const Variant &v = *plan->cur;
    RefBuffers B;
    B.dec_params = v.dec;
    B.lpf_params = v.lpf;
    B.y = v.d_y;
    B.z = v.d_z;
    B.partials = v.d_partials;
    B.lp2 = v.lp2;
    B.dec_raw_params = v.dec_raw;
    B.lp2_raw = v.lp2_raw;
    RefIO io{iq, carrier_stride_samples, pre_shift_hz, freq_offset_hz, hard, soft, n_soft, best_phase, min_margin, plan->fast_pre_shift, plan->rows_per_chunk};
    run_ref(be, v.h, plan->rows, plan->fmt, B, io);
    if (be.err != hipSuccess) return fail(TDM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(be.err));
    return TDM_OK;
}

int tdm_process_device(tdm_plan *plan, const void *iq, int64_t carrier_stride_samples, const double *pre_shift_hz,
                       const double *freq_offset_hz, uint8_t *hard, double *soft, int32_t *n_soft,
                       int32_t *best_phase, double *min_margin, void *stream)
{
    return process_device_impl(plan, iq, carrier_stride_samples, pre_shift_hz, freq_offset_hz, hard, soft, n_soft, best_phase,
                               min_margin, stream, nullptr, nullptr);
}

int tdm_process_device_rows(tdm_plan *plan, const void *iq, int64_t carrier_stride_samples, const int32_t *row_list,
                            const int32_t *n_rows, uint8_t *hard, float *soft, int32_t *n_soft, int32_t *timing_milli,
                            double *min_margin, void *stream)
{
    if (!row_list || !n_rows) return fail(TDM_ERR_INVALID, "null row list");
    return process_device_impl(plan, iq, carrier_stride_samples, nullptr, nullptr, hard, (double *)soft, n_soft, timing_milli,
                               min_margin, stream, row_list, n_rows);
}

int tdm_plan_rrc_filter(tdm_plan *plan, const void *iq, int64_t carrier_stride_samples, float *y, int64_t y_pitch, void *stream)
{
    if (!plan || !iq || !y) return fail(TDM_ERR_INVALID, "null argument");
    if (plan->mode != TDM_MODE_TETRA && plan->mode != TDM_MODE_TETRA_GARDNER) return fail(TDM_ERR_UNSUPPORTED, "a TETRA-mode plan holds the RRC taps");
    if (carrier_stride_samples < plan->tp.n || y_pitch < plan->tp.n || (y_pitch & 1)) return fail(TDM_ERR_INVALID, "stride / pitch (even, >= chunk length)");
    HIP_TRY(hipSetDevice(plan->device));
    HipBackend be;
    be.stream = stream ? (hipStream_t)stream : plan->stream;
    be.timer = &plan->timer;
    {
        HipBackend::Scope s(be, ST_TETRA_MF);
        if (!tetra_mf_launch(plan->tp, plan->rows, iq, tetra_fmt8(plan->fmt), carrier_stride_samples, (float2 *)y, y_pitch, be.stream))
            return fail(TDM_ERR_UNSUPPORTED, "no RRC kernel instantiated for this tap count");
    }
    if (be.err != hipSuccess) return fail(TDM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(be.err));
    return TDM_OK;
}

int tdm_set_stream(void *stream)
{
    g_cur_stream = (hipStream_t)stream;
    return TDM_OK;
}

int tdm_plan_stream(tdm_plan *plan, void **stream)
{
    if (!plan || !stream) return fail(TDM_ERR_INVALID, "null argument");
    *stream = (void *)plan->stream;
    return TDM_OK;
}

int tdm_plan_wait_for(tdm_plan *plan, tdm_plan *other)
{
    if (!plan || !other) return fail(TDM_ERR_INVALID, "null plan");
    if (plan == other) return TDM_OK;
    if (plan->device != other->device) return fail(TDM_ERR_UNSUPPORTED, "tdm_plan_wait_for: the two plans are on different devices");
    HIP_TRY(hipSetDevice(plan->device));
    if (!other->ev_order) HIP_TRY(hipEventCreateWithFlags(&other->ev_order, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(other->ev_order, other->stream));
    HIP_TRY(hipStreamWaitEvent(plan->stream, other->ev_order, 0));
    return TDM_OK;
}

int tdm_plan_sync(tdm_plan *plan)
{
    if (!plan) return fail(TDM_ERR_INVALID, "null plan");
    HIP_TRY(hipSetDevice(plan->device));
    HIP_TRY(hipStreamSynchronize(plan->stream));
    return TDM_OK;
}

int tdm_process(tdm_plan *plan, const void *iq, int64_t carrier_stride_samples, const double *pre_shift_hz,
                const double *freq_offset_hz, uint8_t *hard, double *soft, int32_t *n_soft, int32_t *best_phase,
                double *min_margin)
{
    if (!plan || !iq || !hard || !soft || !n_soft) return fail(TDM_ERR_INVALID, "null argument");
    if (carrier_stride_samples < 0) return fail(TDM_ERR_INVALID, "negative carrier stride");
    HIP_TRY(hipSetDevice(plan->device));
    if (!plan->cur) return fail(TDM_ERR_INVALID, "plan has no current length");
    const RefPlanHost &h = plan->h();
    const int rows = plan->rows;
    // (input rows: one per plan row, or one per rows_per_chunk of them)
    const int in_rows = plan->mode == TDM_MODE_REFERENCE ? rows / plan->rows_per_chunk : rows;
    const size_t span = carrier_stride_samples == 0 ? (size_t)h.n
                                                    : (size_t)(in_rows - 1) * carrier_stride_samples + h.n;
    const size_t bytes = span * fmt_bytes(plan->fmt);
    if (plan->d_iq_bytes < bytes) {
        if (plan->d_iq) (void)hipFree(plan->d_iq);
        plan->d_iq = nullptr;
        plan->d_iq_bytes = 0;
        HIP_TRY(hipMalloc(&plan->d_iq, bytes));
        plan->d_iq_bytes = bytes;
    }
    if (plan->staging_ready && plan->staging_soft < (size_t)rows * h.max_soft) plan->staging_ready = false;   // a longer chunk than the buffers were made for
    if (!plan->staging_ready) {
        // all or nothing: a failed allocation leaves the flag clear, the next call starts over (plan_free releases
        // whatever a failed attempt left behind)
        void **slots[] = {(void **)&plan->d_pre, (void **)&plan->d_foff, (void **)&plan->d_soft, (void **)&plan->d_hard,
                          (void **)&plan->d_nsoft, (void **)&plan->d_bp, (void **)&plan->d_margin};
        const size_t sizes[] = {rows * sizeof(double), rows * sizeof(double),
                                (size_t)rows * h.max_soft * 2 * sizeof(double),   // cf32 in TETRA mode uses half
                                (size_t)rows * h.max_soft, rows * sizeof(int32_t), rows * sizeof(int32_t), rows * sizeof(double)};
        for (int i = 0; i < 7; ++i) {
            if (*slots[i]) { (void)hipFree(*slots[i]); *slots[i] = nullptr; }
            HIP_TRY(hipMalloc(slots[i], sizes[i]));
        }
        plan->staging_ready = true;
        plan->staging_soft = (size_t)rows * h.max_soft;
    }
    hipStream_t st = plan->stream;
    HIP_TRY(hipMemcpyAsync(plan->d_iq, iq, bytes, hipMemcpyHostToDevice, st));
    if (pre_shift_hz) HIP_TRY(hipMemcpyAsync(plan->d_pre, pre_shift_hz, rows * sizeof(double), hipMemcpyHostToDevice, st));
    if (freq_offset_hz) HIP_TRY(hipMemcpyAsync(plan->d_foff, freq_offset_hz, rows * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(plan->d_hard, 0, (size_t)rows * h.max_soft, st));
    int rc = tdm_process_device(plan, plan->d_iq, carrier_stride_samples, pre_shift_hz ? plan->d_pre : nullptr,
                                freq_offset_hz ? plan->d_foff : nullptr, plan->d_hard, plan->d_soft, plan->d_nsoft,
                                plan->d_bp, plan->d_margin, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(hard, plan->d_hard, (size_t)rows * h.max_soft, hipMemcpyDeviceToHost, st));
    const size_t soft_elem = plan->mode != TDM_MODE_REFERENCE ? 2 * sizeof(float) : 2 * sizeof(double);
    HIP_TRY(hipMemcpyAsync(soft, plan->d_soft, (size_t)rows * h.max_soft * soft_elem, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(n_soft, plan->d_nsoft, rows * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (best_phase) HIP_TRY(hipMemcpyAsync(best_phase, plan->d_bp, rows * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (min_margin) HIP_TRY(hipMemcpyAsync(min_margin, plan->d_margin, rows * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return TDM_OK;
}

// ---- host-fed pipeline: H2D of batch i+1 and D2H of batch i-1 overlap the kernels of batch i ----------
int tdm_process_pipelined(tdm_plan *plan, const void *iq, int64_t n_batches, const double *freq_offset_hz, uint8_t *hard,
                          void *soft, int32_t *n_soft, int32_t *best_phase, double *min_margin)
{
    if (!plan || !iq || !hard || !soft || !n_soft || n_batches < 1) return fail(TDM_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(plan->device));
    if (!plan->cur) return fail(TDM_ERR_INVALID, "plan has no current length");
    const RefPlanHost &h = plan->h();
    const int rows = plan->rows;
    // (a batch's input: one row per plan row, or one per rows_per_chunk of them -- the rows of a time-batched plan carry
    //  pre-shifts, which this entry point does not take: such a plan goes through tdm_process)
    if (plan->mode == TDM_MODE_REFERENCE && plan->rows_per_chunk > 1)
        return fail(TDM_ERR_UNSUPPORTED, "tdm_process_pipelined: the plan has rows_per_chunk > 1 (use tdm_process / tdm_process_device)");
    const size_t in_bytes = (size_t)rows * h.n * fmt_bytes(plan->fmt);
    const size_t soft_elem = plan->mode != TDM_MODE_REFERENCE ? 2 * sizeof(float) : 2 * sizeof(double);
    const size_t hard_bytes = (size_t)rows * h.max_soft, soft_bytes = (size_t)rows * h.max_soft * soft_elem;
    // pin the caller's buffers in place so the copies are truly asynchronous (best effort)
    const bool pin_in = hipHostRegister((void *)iq, in_bytes * n_batches, hipHostRegisterDefault) == hipSuccess;
    const bool pin_h = hipHostRegister(hard, hard_bytes * n_batches, hipHostRegisterDefault) == hipSuccess;
    const bool pin_s = hipHostRegister(soft, soft_bytes * n_batches, hipHostRegisterDefault) == hipSuccess;
    (void)hipGetLastError();
    struct Slot {
        void *iq = nullptr; uint8_t *hard = nullptr; void *soft = nullptr; int32_t *ns = nullptr, *bp = nullptr; double *mm = nullptr;
        hipEvent_t in_done{}, comp_done{}, out_done{};
    } sl[2];
    hipStream_t s_in = nullptr, s_out = nullptr;
    double *d_fo = nullptr;
    int rc = TDM_OK;
    auto cleanup = [&]() {
        // copies and kernels may still be in flight on the error path: drain before buffers and pins go away
        if (s_in) (void)hipStreamSynchronize(s_in);
        if (s_out) (void)hipStreamSynchronize(s_out);
        (void)hipStreamSynchronize(plan->stream);
        for (auto &x : sl) {
            void *ps[] = {x.iq, x.hard, x.soft, x.ns, x.bp, x.mm};
            for (void *q : ps) if (q) (void)hipFree(q);
            if (x.in_done) (void)hipEventDestroy(x.in_done);
            if (x.comp_done) (void)hipEventDestroy(x.comp_done);
            if (x.out_done) (void)hipEventDestroy(x.out_done);
        }
        if (d_fo) (void)hipFree(d_fo);
        if (s_in) (void)hipStreamDestroy(s_in);
        if (s_out) (void)hipStreamDestroy(s_out);
        if (pin_in) (void)hipHostUnregister((void *)iq);
        if (pin_h) (void)hipHostUnregister(hard);
        if (pin_s) (void)hipHostUnregister(soft);
    };
#define PIPE_TRY(expr)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            rc = fail(TDM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
            cleanup();                                                                     \
            return rc;                                                                     \
        }                                                                                  \
    } while (0)
    PIPE_TRY(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking));
    PIPE_TRY(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking));
    for (auto &x : sl) {
        PIPE_TRY(hipMalloc(&x.iq, in_bytes));
        PIPE_TRY(hipMalloc(&x.hard, hard_bytes));
        PIPE_TRY(hipMalloc(&x.soft, soft_bytes));
        PIPE_TRY(hipMalloc(&x.ns, rows * sizeof(int32_t)));
        PIPE_TRY(hipMalloc(&x.bp, rows * sizeof(int32_t)));
        PIPE_TRY(hipMalloc(&x.mm, rows * sizeof(double)));
        PIPE_TRY(hipEventCreateWithFlags(&x.in_done, hipEventDisableTiming));
        PIPE_TRY(hipEventCreateWithFlags(&x.comp_done, hipEventDisableTiming));
        PIPE_TRY(hipEventCreateWithFlags(&x.out_done, hipEventDisableTiming));
    }
    if (freq_offset_hz) {
        PIPE_TRY(hipMalloc(&d_fo, rows * sizeof(double)));
        PIPE_TRY(hipMemcpy(d_fo, freq_offset_hz, rows * sizeof(double), hipMemcpyHostToDevice));
    }
    const char *src = (const char *)iq;
    for (int64_t b = 0; b < n_batches; ++b) {
        Slot &x = sl[b & 1];
        if (b >= 2) {
            PIPE_TRY(hipStreamWaitEvent(s_in, x.comp_done, 0));        // slot's input consumed
            PIPE_TRY(hipStreamWaitEvent(plan->stream, x.out_done, 0)); // slot's outputs copied out
        }
        PIPE_TRY(hipMemcpyAsync(x.iq, src + (size_t)b * in_bytes, in_bytes, hipMemcpyHostToDevice, s_in));
        PIPE_TRY(hipEventRecord(x.in_done, s_in));
        PIPE_TRY(hipStreamWaitEvent(plan->stream, x.in_done, 0));
        PIPE_TRY(hipMemsetAsync(x.hard, 0, hard_bytes, plan->stream));
        rc = tdm_process_device(plan, x.iq, h.n, nullptr, d_fo, x.hard, (double *)x.soft, x.ns, x.bp, x.mm, nullptr);
        if (rc) { cleanup(); return rc; }
        PIPE_TRY(hipEventRecord(x.comp_done, plan->stream));
        PIPE_TRY(hipStreamWaitEvent(s_out, x.comp_done, 0));
        PIPE_TRY(hipMemcpyAsync(hard + (size_t)b * hard_bytes, x.hard, hard_bytes, hipMemcpyDeviceToHost, s_out));
        PIPE_TRY(hipMemcpyAsync((char *)soft + (size_t)b * soft_bytes, x.soft, soft_bytes, hipMemcpyDeviceToHost, s_out));
        PIPE_TRY(hipMemcpyAsync(n_soft + (size_t)b * rows, x.ns, rows * sizeof(int32_t), hipMemcpyDeviceToHost, s_out));
        if (best_phase) PIPE_TRY(hipMemcpyAsync(best_phase + (size_t)b * rows, x.bp, rows * sizeof(int32_t), hipMemcpyDeviceToHost, s_out));
        if (min_margin) PIPE_TRY(hipMemcpyAsync(min_margin + (size_t)b * rows, x.mm, rows * sizeof(double), hipMemcpyDeviceToHost, s_out));
        PIPE_TRY(hipEventRecord(x.out_done, s_out));
    }
    PIPE_TRY(hipStreamSynchronize(s_out));
    PIPE_TRY(hipStreamSynchronize(plan->stream));
#undef PIPE_TRY
    cleanup();
    return TDM_OK;
}

// ---- timing -----------------------------------------------------------------------------------
int tdm_plan_time_begin(tdm_plan *plan)
{
    if (!plan) return fail(TDM_ERR_INVALID, "null plan");
    HIP_TRY(hipSetDevice(plan->device));
    plan->timer.release_all();
    plan->timer.on = true;
    HIP_TRY(hipEventRecord(plan->ev0, plan->stream));
    return TDM_OK;
}

int tdm_plan_time_begin_total(tdm_plan *plan)
{
    if (!plan) return fail(TDM_ERR_INVALID, "null plan");
    HIP_TRY(hipSetDevice(plan->device));
    plan->timer.release_all();
    plan->timer.on = false;
    HIP_TRY(hipEventRecord(plan->ev0, plan->stream));
    return TDM_OK;
}

int tdm_plan_time_end(tdm_plan *plan, float *elapsed_ms)
{
    if (!plan) return fail(TDM_ERR_INVALID, "null plan");
    HIP_TRY(hipSetDevice(plan->device));
    HIP_TRY(hipEventRecord(plan->ev1, plan->stream));
    HIP_TRY(hipEventSynchronize(plan->ev1));
    plan->timer.on = false;
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, plan->ev0, plan->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    return TDM_OK;
}

int tdm_plan_stage_times(tdm_plan *plan, int32_t max_stages, const char **names, float *ms, int32_t *n_stages)
{
    if (!plan || !names || !ms || !n_stages) return fail(TDM_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(plan->device));
    float acc[ST_COUNT] = {0};
    int cnt[ST_COUNT] = {0};
    for (auto &r : plan->timer.recs) {
        float t = 0;
        HIP_TRY(hipEventSynchronize(r.b));
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        acc[r.stage] += t;
        cnt[r.stage]++;
    }
    int n = 0;
    for (int s = 0; s < ST_COUNT && n < max_stages; ++s) {
        if (!cnt[s]) continue;
        names[n] = kStageNames[s];
        ms[n] = acc[s] / cnt[s];  // average per launch
        ++n;
    }
    *n_stages = n;
    return TDM_OK;
}

// ---- measured HBM ceilings (the roofline denominator is a copy-kernel figure from the box the bench
// runs on, next to the 8 TB/s of the data sheet) ---------------------------------------------------------------------
}  // extern "C"
namespace {
typedef float ceil_f4 __attribute__((ext_vector_type(4)));
// (four 16-byte accesses in flight per lane and trip; n16 is a multiple of 4 x the grid's stride for the sizes measured,
// the remainder loop covers any other)
template <bool NT>
__global__ __launch_bounds__(256) void k_ceiling_copy(const ceil_f4 *__restrict__ a, ceil_f4 *__restrict__ b, size_t n16)
{
    const size_t S = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * S < n16; i += 4 * S) {
        ceil_f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * S) : a[i + u * S];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], b + i + u * S);
            else b[i + u * S] = v[u];
        }
    }
    for (; i < n16; i += S) b[i] = a[i];
}
template <bool NT>
__global__ __launch_bounds__(256) void k_ceiling_read(const ceil_f4 *__restrict__ a, float *sink, size_t n16)
{
    ceil_f4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t S = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * S < n16; i += 4 * S) {
        ceil_f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * S) : a[i + u * S];
        acc += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (; i < n16; i += S)
        acc += a[i];
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123456.789f) sink[0] = s;   // (keeps the loads; never true for the zero-filled buffer)
}
// the flat forms: one 16-byte access per lane, one workgroup per 4 KB -- the form that reached the highest copy rate
// of all those tried on MI355X (6.29 TB/s; grid-stride forms 5.0-5.7, hipMemcpyDtoD 5.0: tools/harness/copy_bench.hip)
__global__ __launch_bounds__(256) void k_ceiling_copy_flat(const ceil_f4 *__restrict__ a, ceil_f4 *__restrict__ b, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_ceiling_read_flat(const ceil_f4 *__restrict__ a, float *sink, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) {
        const ceil_f4 v = a[i];
        if (v.x + v.y + v.z + v.w == 123456.789f) sink[0] = v.x;   // (never true for the zero-filled buffer)
    }
}
__global__ __launch_bounds__(256) void k_ceiling_write_flat(ceil_f4 *__restrict__ b, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) b[i] = ceil_f4{1.f, 2.f, 3.f, 4.f};
}
__global__ __launch_bounds__(256) void k_ceiling_write(ceil_f4 *__restrict__ b, size_t n16)
{
    const ceil_f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) b[i] = v;
}
}  // namespace
extern "C" {

int tdm_hbm_ceiling(int32_t device, size_t bytes, int32_t reps, double *gbs)
{
    if (!gbs || reps < 1 || bytes < (1u << 20)) return fail(TDM_ERR_INVALID, "tdm_hbm_ceiling: gbs[3], reps >= 1, bytes >= 1 MiB");
    int rc = use_device(device);
    if (rc) return rc;
    const size_t n16 = bytes / 16;
    ceil_f4 *a = nullptr, *b = nullptr;
    HIP_TRY(hipMalloc((void **)&a, n16 * 16));
    if (hipMalloc((void **)&b, n16 * 16) != hipSuccess) {
        (void)hipFree(a);
        return fail(TDM_ERR_NOMEM, "tdm_hbm_ceiling: hipMalloc");
    }
    hipStream_t st = g_cur_stream;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipMemsetAsync(a, 0, n16 * 16, st);
    (void)hipMemsetAsync(b, 0, n16 * 16, st);
    gbs[0] = gbs[1] = gbs[2] = 0.0;
    hipError_t err = hipSuccess;
    // grid-stride kernels, 16 bytes per lane, 8 workgroups of 256 threads per compute unit (2048) and twice that; the best of
    // the plain / non-temporal forms and of the two grids is the ceiling of each direction
    for (int variant = 0; variant < 5 && err == hipSuccess; ++variant) {   // plain / non-temporal x 2048 / 8192 workgroups, then the flat forms
        const bool nt = variant & 1, flat = variant == 4;
        const int grid = flat ? (int)((n16 + 255) / 256) : ((variant & 2) ? 8192 : 2048);
        for (int what = 0; what < 3; ++what) {
            if (what == 2 && nt) continue;
            auto launch = [&]() {
                if (flat) {
                    if (what == 0) hipLaunchKernelGGL(k_ceiling_copy_flat, dim3(grid), dim3(256), 0, st, a, b, n16);
                    else if (what == 1) hipLaunchKernelGGL(k_ceiling_read_flat, dim3(grid), dim3(256), 0, st, a, (float *)b, n16);
                    else hipLaunchKernelGGL(k_ceiling_write_flat, dim3(grid), dim3(256), 0, st, b, n16);
                    return;
                }
                if (what == 0) { if (nt) hipLaunchKernelGGL(k_ceiling_copy<true>, dim3(grid), dim3(256), 0, st, a, b, n16); else hipLaunchKernelGGL(k_ceiling_copy<false>, dim3(grid), dim3(256), 0, st, a, b, n16); }
                else if (what == 1) { if (nt) hipLaunchKernelGGL(k_ceiling_read<true>, dim3(grid), dim3(256), 0, st, a, (float *)b, n16); else hipLaunchKernelGGL(k_ceiling_read<false>, dim3(grid), dim3(256), 0, st, a, (float *)b, n16); }
                else hipLaunchKernelGGL(k_ceiling_write, dim3(grid), dim3(256), 0, st, b, n16);
            };
            for (int i = 0; i < 3; ++i) launch();
            (void)hipEventRecord(e0, st);
            for (int i = 0; i < reps; ++i) launch();
            (void)hipEventRecord(e1, st);
            err = hipEventSynchronize(e1);
            if (err != hipSuccess) break;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double moved = (what == 0 ? 2.0 : 1.0) * (double)(n16 * 16) * reps;
            const double rate = moved / ((double)ms * 1e-3) / 1e9;
            if (rate > gbs[what]) gbs[what] = rate;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    if (err != hipSuccess) return fail(TDM_ERR_HIP, std::string("tdm_hbm_ceiling: ") + hipGetErrorString(err));
    return TDM_OK;
}

// ---- device memory helpers --------------------------------------------------------------------
int tdm_dev_alloc(int32_t device, size_t bytes, void **ptr)
{
    if (!ptr) return fail(TDM_ERR_INVALID, "null ptr");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMalloc(ptr, bytes ? bytes : 1));
    return TDM_OK;
}
int tdm_dev_free(int32_t device, void *ptr)
{
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipFree(ptr));
    return TDM_OK;
}
int tdm_dev_upload(int32_t device, void *dst_dev, const void *src_host, size_t bytes)
{
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return TDM_OK;
}
int tdm_dev_download(int32_t device, void *dst_host, const void *src_dev, size_t bytes)
{
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return TDM_OK;
}
int tdm_host_register(int32_t device, void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return fail(TDM_ERR_INVALID, "tdm_host_register: null buffer");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return TDM_OK;
}
int tdm_host_unregister(int32_t device, void *ptr)
{
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipHostUnregister(ptr));
    return TDM_OK;
}
int tdm_dev_sync(int32_t device)
{
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return TDM_OK;
}

}  // extern "C"

// ---- single-method entry points (host pointers, blocking) ---------------------------------------
namespace {
// Device scratch of the stand-alone entry points (filter_signal, decimate, extract_symbols, ... each a host array in, a
// host array out): buffers go back to a pool instead of hipFree, so a caller that uses these per read (scanner.py:42-147
// style loops) pays for hipMalloc once per size class, not per call.  Every entry point synchronises the device before
// it returns, so a pooled buffer is idle when it is handed out again.
struct ScratchPool {
    struct Ent { void *p; size_t bytes; int dev; };
    std::vector<Ent> idle;
    size_t held = 0;
    std::mutex mu;
    static constexpr size_t kMaxIdle = 24, kMaxHeld = size_t(1) << 30;
    void *take(size_t bytes, int dev, size_t *got)
    {
        std::lock_guard<std::mutex> lk(mu);
        int best = -1;
        for (int i = 0; i < (int)idle.size(); ++i)
            if (idle[i].dev == dev && idle[i].bytes >= bytes && idle[i].bytes <= 4 * bytes + 65536 &&
                (best < 0 || idle[i].bytes < idle[best].bytes))
                best = i;
        if (best < 0) return nullptr;
        void *q = idle[best].p;
        *got = idle[best].bytes;
        held -= idle[best].bytes;
        idle.erase(idle.begin() + best);
        return q;
    }
    void give(void *q, size_t bytes, int dev)
    {
        std::lock_guard<std::mutex> lk(mu);
        idle.push_back({q, bytes, dev});
        held += bytes;
        while (idle.size() > kMaxIdle || held > kMaxHeld) {   // drop the largest
            int big = 0;
            for (int i = 1; i < (int)idle.size(); ++i)
                if (idle[i].bytes > idle[big].bytes) big = i;
            int cur = 0;
            (void)hipGetDevice(&cur);
            (void)hipSetDevice(idle[big].dev);
            (void)hipFree(idle[big].p);
            (void)hipSetDevice(cur);
            held -= idle[big].bytes;
            idle.erase(idle.begin() + big);
        }
    }
};
static ScratchPool g_scratch;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int dev = 0;
    ~DevBuf() { if (p) g_scratch.give(p, cap, dev); }
    int alloc(size_t bytes)
    {
        if (!bytes) bytes = 16;
        HIP_TRY(hipGetDevice(&dev));
        if ((p = g_scratch.take(bytes, dev, &cap))) return TDM_OK;
        bytes = (bytes + 4095) & ~size_t(4095);
        HIP_TRY(hipMalloc(&p, bytes));
        cap = bytes;
        return TDM_OK;
    }
    template <class T> T *as() { return (T *)p; }
};

// device copy of one zero-phase stage (tables + work for one row) out of the scratch pool
struct DevZp {
    ZpParams params{};
    DevBuf blob, sblob, work;
    int init(const ZpHostTables &t)
    {
        int rc;
        params = t.p;
        if ((rc = blob.alloc(t.blob.size() * sizeof(double)))) return rc;
        HIP_TRY(hipMemcpy(blob.p, t.blob.data(), t.blob.size() * sizeof(double), hipMemcpyHostToDevice));
        if (t.shared) {
            if ((rc = sblob.alloc(t.shared->blob.size() * sizeof(double)))) return rc;
            HIP_TRY(hipMemcpy(sblob.p, t.shared->blob.data(), t.shared->blob.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        t.bind(params, blob.as<double>(), sblob.as<double>());
        if ((rc = work.alloc(zp_work_doubles(params, 1) * sizeof(double)))) return rc;
        zp_bind_work(params, 1, work.as<double>());
        return TDM_OK;
    }
};

// one zero-phase stage on a c128 host array
int run_zp_stage(const ZpHostTables &t, bool sos, const double *x, int64_t n, double *y, int64_t n_out, double fs)
{
    // scipy's zero-phase filters carry one non-finite sample over the whole output (forward pass to the end, backward pass
    // back to the start: every value NaN + NaN j; goldens tests/golden/nonfinite.npz).  The device evaluates the filter in
    // blocks whose carries are cut below 1e-30 and would keep the NaN local, so this host-buffer entry point looks at its
    // input first: nothing is computed for such a call, the answer is the fill.  (process() decides the same thing on the
    // device, from the phase powers: FinishArgs::smear.)
    for (int64_t i = 0; i < 2 * n; ++i)
        if (!std::isfinite(x[i])) {
            std::fill(y, y + 2 * n_out, std::numeric_limits<double>::quiet_NaN());
            return TDM_OK;
        }
    DevZp dz;
    DevBuf dx, dy;
    int rc;
    if ((rc = dz.init(t))) return rc;
    if ((rc = dx.alloc((size_t)n * 16))) return rc;
    if ((rc = dy.alloc((size_t)n_out * 16))) return rc;
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * 16, hipMemcpyHostToDevice));
    HipBackend be;
    RawLoader<FMT_CF64, false> ld{dx.p, n, nullptr, fs};
    StagedLoader<PlainC128Src> ls{{dx.as<double>(), n}};
    if (sos) {
        if (t.p.pform) {
            RefPlanHost h;
            h.q = t.p.out_stride;
            h.dec.p.nb = t.p.nb;
            RawLoaderRT<false> lr{dx.p, n, nullptr, fs, FMT_CF64, 0};
            run_pz_block(be, h, dz.params, lr, 1);
        } else {
            be.zp_block<2, 4, kLDec, kEdgeSos>(dz.params, ld, t.p.nb, 1);
        }
        be.zp_carry<2, 4>(dz.params, t.p.nb, 1);
        be.zp_fixup<8, kLDec>(dz.params, t.p.nb, 1, dy.as<double>(), n_out, nullptr, fs);
    } else {
        be.zp_block<2, 2, kLLpf, kEdgeTf>(dz.params, ls, t.p.nb, 1);
        be.zp_carry<2, 2>(dz.params, t.p.nb, 1);
        be.zp_fixup<4, kLLpf>(dz.params, t.p.nb, 1, dy.as<double>(), n_out, nullptr, fs);
    }
    if (be.err != hipSuccess) return fail(TDM_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(be.err));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(y, dy.p, (size_t)n_out * 16, hipMemcpyDeviceToHost));
    return TDM_OK;
}
}  // namespace

extern "C" {

int tdm_filter_signal(const double *x, int64_t n, double bandwidth, double fs, double *y, int32_t *applied,
                      int32_t device)
{
    if (!x || !y || n < 0 || !(fs > 0)) return fail(TDM_ERR_INVALID, "bad argument");
    int rc = use_device(device);
    if (rc) return rc;
    if (applied) *applied = 0;
    if (n <= kEdgeTf) {  // filtfilt raises; the reference returns the input (processor.py:81-83)
        std::memcpy(y, x, (size_t)n * 16);
        return TDM_OK;
    }
    Tf4 tf = design_butter4(butter_cutoff(bandwidth, fs));
    ZpHostTables t = build_zp_tables(desc_from_tf(tf), n, kEdgeTf, kLLpf, n, 1);
    rc = run_zp_stage(t, false, x, n, y, n, fs);
    if (rc == TDM_OK && applied) *applied = 1;
    return rc;
}

int tdm_decimate(const double *x, int64_t n, int32_t q, double *y, int64_t *n_out, int32_t device)
{
    if (!x || !y || q < 2 || q > 4096) return fail(TDM_ERR_INVALID, "bad argument");
    if (n <= kEdgeSos) return fail(TDM_ERR_INVALID, "The length of the input vector x must be greater than padlen, which is 27.");
    int rc = use_device(device);
    if (rc) return rc;
    Sos4 s = design_cheby1_8(0.05, 0.8 / q);
    const int64_t m = (n + q - 1) / q;
    const int S = pz_outputs_per_lane(q);
    ZpHostTables t = S ? build_pz_tables(s.sos, 4, n, kEdgeSos, q * S, S, m, q)
                       : build_zp_tables(desc_from_sos(s), n, kEdgeSos, kLDec, m, q);
    rc = run_zp_stage(t, true, x, n, y, m, 1.0);
    if (rc == TDM_OK && n_out) *n_out = m;
    return rc;
}

int tdm_frequency_shift(const double *x, int64_t n, double freq_offset, double fs, double *y, int32_t device)
{
    if ((!x || !y) && n > 0) return fail(TDM_ERR_INVALID, "null argument");
    int rc = use_device(device);
    if (rc) return rc;
    if (n <= 0) return TDM_OK;
    DevBuf dx, dy;
    if ((rc = dx.alloc((size_t)n * 16)) || (rc = dy.alloc((size_t)n * 16))) return rc;
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_shift, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx.as<double>(), dy.as<double>(), n,
                       freq_offset, fs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(y, dy.p, (size_t)n * 16, hipMemcpyDeviceToHost));
    return TDM_OK;
}

static int run_finish(const double *x, int64_t n, int sps, int do_extract, int do_demod, double *soft_out,
                      uint8_t *hard_out, int64_t *n_soft_out, int32_t *best_phase, double *min_margin)
{
    DevBuf dx, dsoft, dhard, dmeta;
    int rc;
    if ((rc = dx.alloc((size_t)n * 16)) || (rc = dsoft.alloc((size_t)n * 16)) || (rc = dhard.alloc((size_t)n)) ||
        (rc = dmeta.alloc(64)))
        return rc;
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dmeta.p, 0, 64));
    FinishArgs fa{};
    fa.z = dx.as<double>();
    fa.n = n;
    fa.row_stride = n;
    fa.sps = sps;
    fa.do_extract = do_extract;
    fa.do_demod = do_demod;
    fa.max_soft = (int32_t)n;
    fa.soft = dsoft.as<double>();
    fa.hard = dhard.as<uint8_t>();
    fa.n_soft = (int32_t *)dmeta.p;
    fa.best_phase = (int32_t *)dmeta.p + 1;
    fa.min_margin = (double *)dmeta.p + 1;
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(kFinishThreads), 0, 0, fa);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    struct { int32_t ns, bp; double mm; } meta;
    HIP_TRY(hipMemcpy(&meta, dmeta.p, sizeof(meta), hipMemcpyDeviceToHost));
    if (soft_out && meta.ns > 0) HIP_TRY(hipMemcpy(soft_out, dsoft.p, (size_t)meta.ns * 16, hipMemcpyDeviceToHost));
    if (hard_out && meta.ns > 1) HIP_TRY(hipMemcpy(hard_out, dhard.p, (size_t)(meta.ns - 1), hipMemcpyDeviceToHost));
    if (n_soft_out) *n_soft_out = meta.ns;
    if (best_phase) *best_phase = meta.bp;
    if (min_margin) *min_margin = meta.mm;
    return TDM_OK;
}

int tdm_extract_symbols(const double *x, int64_t n, double fs, double symbol_rate, double *y, int64_t *n_out,
                        int32_t *best_phase, int32_t device)
{
    if (!n_out || (n > 0 && (!x || !y)) || !(symbol_rate > 0)) return fail(TDM_ERR_INVALID, "bad argument");
    int rc = use_device(device);
    if (rc) return rc;
    *n_out = 0;
    if (best_phase) *best_phase = 0;
    if (n <= 0) return TDM_OK;
    return run_finish(x, n, (int)(fs / symbol_rate), 1, 0, y, nullptr, n_out, best_phase, nullptr);
}

int tdm_demodulate_dqpsk(const double *x, int64_t n, uint8_t *out, int64_t *n_out, double *min_margin, int32_t device)
{
    if (!n_out || (n > 1 && (!x || !out))) return fail(TDM_ERR_INVALID, "bad argument");
    int rc = use_device(device);
    if (rc) return rc;
    *n_out = 0;
    if (n < 2) return TDM_OK;
    int64_t ns = 0;
    rc = run_finish(x, n, 1, 0, 1, nullptr, out, &ns, nullptr, min_margin);
    if (rc == TDM_OK) *n_out = ns > 0 ? ns - 1 : 0;
    return rc;
}

// out = DFT_L(in) with kernel exp(sign 2 pi i nk / L), times scale; in / out: L complex doubles in device memory (may be the
// same buffer); w0, w1, w2: work buffers of fft_work_len(L) complex doubles each.  Launches on the null stream.
static int64_t fft_work_len(int64_t L)
{
    if ((L & (L - 1)) == 0) return L;
    int64_t M = 1;
    while (M < 2 * L - 1) M <<= 1;
    return M;
}
static int fft_pow2(f64c *&a, f64c *&tmp, int64_t M, double sign)   // result in `a` (the pointers are swapped as the passes go)
{
    const int64_t half = M / 2;
    for (int64_t p = 1; p < M; p <<= 1) {
        hipLaunchKernelGGL(k_fft2_pass, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, 0, a, tmp, half, p, sign);
        std::swap(a, tmp);
    }
    HIP_TRY(hipGetLastError());
    return TDM_OK;
}
static int dft_any(const f64c *in, f64c *out, int64_t L, double sign, double scale, f64c *w0, f64c *w1, f64c *w2)
{
    int rc;
    const int64_t M = fft_work_len(L);
    const unsigned gm = (unsigned)((M + 255) / 256), gl = (unsigned)((L + 255) / 256);
    if (M == L) {   // a power of two: the passes themselves
        if (L == 1) { hipLaunchKernelGGL(k_fft_scale_copy, dim3(1), dim3(256), 0, 0, in, out, L, scale); return TDM_OK; }
        HIP_TRY(hipMemcpyAsync(w0, in, (size_t)L * 16, hipMemcpyDeviceToDevice, 0));
        f64c *a = w0, *t = w1;
        if ((rc = fft_pow2(a, t, M, sign))) return rc;
        hipLaunchKernelGGL(k_fft_scale_copy, dim3(gl), dim3(256), 0, 0, a, out, L, scale);
        return TDM_OK;
    }
    hipLaunchKernelGGL(k_bluestein_pre, dim3(gm), dim3(256), 0, 0, in, w0, w1, L, M, sign);
    f64c *a = w0, *b = w1, *t = w2;
    if ((rc = fft_pow2(a, t, M, -1.0))) return rc;       // (a, t) now name two of the three buffers; b is the third
    f64c *t2 = t;
    if ((rc = fft_pow2(b, t2, M, -1.0))) return rc;
    hipLaunchKernelGGL(k_fft_cmul, dim3(gm), dim3(256), 0, 0, a, b, M);
    f64c *t3 = t2;
    if ((rc = fft_pow2(a, t3, M, 1.0))) return rc;
    hipLaunchKernelGGL(k_bluestein_post, dim3(gl), dim3(256), 0, 0, a, out, L, sign, scale / (double)M);
    HIP_TRY(hipGetLastError());
    return TDM_OK;
}

// scipy.signal.resample (FFT method) on long inputs: X = fft(x), the spectrum bookkeeping of resample_plan.hpp, y = ifft(Y) num / n
static int resample_fft(const double *x, int64_t n, int64_t num, double *y, const ResamplePlan &rp)
{
    const int64_t nb = (int64_t)rp.src_bins.size(), nt = (int64_t)rp.term_src.size();
    const int64_t W = std::max(fft_work_len(n), fft_work_len(num));
    DevBuf dx, dX, dY, dy, w0, w1, w2, dbins, dsrc, ddst, dw;
    int rc;
    if ((rc = dx.alloc((size_t)n * 16)) || (rc = dX.alloc((size_t)n * 16)) || (rc = dY.alloc((size_t)num * 16)) || (rc = dy.alloc((size_t)num * 16)) ||
        (rc = w0.alloc((size_t)W * 16)) || (rc = w1.alloc((size_t)W * 16)) || (rc = w2.alloc((size_t)W * 16)) ||
        (rc = dbins.alloc((size_t)nb * 8)) || (rc = dsrc.alloc((size_t)nt * 8)) || (rc = ddst.alloc((size_t)nt * 8)) || (rc = dw.alloc((size_t)nt * 8)))
        return rc;
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dbins.p, rp.src_bins.data(), (size_t)nb * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dsrc.p, rp.term_src.data(), (size_t)nt * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ddst.p, rp.term_dst.data(), (size_t)nt * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dw.p, rp.term_w.data(), (size_t)nt * 8, hipMemcpyHostToDevice));
    if ((rc = dft_any(dx.as<f64c>(), dX.as<f64c>(), n, -1.0, 1.0, w0.as<f64c>(), w1.as<f64c>(), w2.as<f64c>()))) return rc;
    HIP_TRY(hipMemsetAsync(dY.p, 0, (size_t)num * 16, 0));
    hipLaunchKernelGGL(k_resample_terms, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, 0, dX.as<f64c>(), dY.as<f64c>(), dbins.as<int64_t>(),
                       dsrc.as<int64_t>(), ddst.as<int64_t>(), dw.as<double>(), nt);
    // ifft's 1 / num times resample's num / n
    if ((rc = dft_any(dY.as<f64c>(), dy.as<f64c>(), num, 1.0, 1.0 / (double)n, w0.as<f64c>(), w1.as<f64c>(), w2.as<f64c>()))) return rc;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(y, dy.p, (size_t)num * 16, hipMemcpyDeviceToHost));
    return TDM_OK;
}

int tdm_resample(const double *x, int64_t n, int64_t num, double *y, int32_t device)
{
    if (n < 0 || num < 0 || (n > 0 && !x) || (num > 0 && !y)) return fail(TDM_ERR_INVALID, "bad argument");
    if (n >= (int64_t(1) << 31) || num >= (int64_t(1) << 31)) return fail(TDM_ERR_UNSUPPORTED, "resample: length >= 2^31");
    int rc = use_device(device);
    if (rc) return rc;
    if (num == 0) return TDM_OK;
    if (n == 0) { std::memset(y, 0, (size_t)num * 16); return TDM_OK; }
    ResamplePlan rp = build_resample_plan(n, num);
    const int64_t nb = (int64_t)rp.src_bins.size(), nt = (int64_t)rp.term_src.size();
    if ((double)n * (double)nb + (double)nt * (double)num >= 16777216.0 && n <= (int64_t(1) << 24) && num <= (int64_t(1) << 24))
        return resample_fft(x, n, num, y, rp);   // long inputs: fast transforms (fft_kernels.hpp) instead of direct sums
    DevBuf dx, dX, dy, dbins, dsrc, ddst, dw;
    if ((rc = dx.alloc((size_t)n * 16)) || (rc = dX.alloc((size_t)nb * 16)) || (rc = dy.alloc((size_t)num * 16)) ||
        (rc = dbins.alloc((size_t)nb * 8)) || (rc = dsrc.alloc((size_t)nt * 8)) || (rc = ddst.alloc((size_t)nt * 8)) ||
        (rc = dw.alloc((size_t)nt * 8)))
        return rc;
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n * 16, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dbins.p, rp.src_bins.data(), (size_t)nb * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dsrc.p, rp.term_src.data(), (size_t)nt * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ddst.p, rp.term_dst.data(), (size_t)nt * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dw.p, rp.term_w.data(), (size_t)nt * 8, hipMemcpyHostToDevice));
    // X[k] = sum_n x[n] exp(-2 pi i k n / N) for the needed bins k
    hipLaunchKernelGGL(k_dft_terms, dim3((unsigned)nb), dim3(kFinishThreads), 0, 0, dbins.as<int64_t>(),
                       dx.as<double>(), n, (const int64_t *)nullptr, (const int64_t *)nullptr, (const double *)nullptr,
                       n, -1.0, 1.0, dX.as<double>());
    HIP_TRY(hipGetLastError());
    // y[t] = (1/N) sum_terms w X[src] exp(+2 pi i dst t / num)      (ifft's 1/num times num/N)
    hipLaunchKernelGGL(k_dft_terms, dim3((unsigned)num), dim3(kFinishThreads), 0, 0, (const int64_t *)nullptr,
                       dX.as<double>(), nt, dsrc.as<int64_t>(), ddst.as<int64_t>(), dw.as<double>(), num, 1.0,
                       1.0 / (double)n, dy.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(y, dy.p, (size_t)num * 16, hipMemcpyDeviceToHost));
    return TDM_OK;
}

// ---- spectrum / AFC / signal gate in front of process() ----------------------------------
int tdm_spectrum_gate(const void *iq, int32_t in_fmt, int64_t row_stride, int64_t n_samples, int32_t rows,
                      double sample_rate, double *out, double *afc, int32_t device_pointers, int32_t device)
{
    if (!iq || !out || rows < 1 || n_samples < 0 || in_fmt < 0 || in_fmt > 3 || !(sample_rate > 0))
        return fail(TDM_ERR_INVALID, "bad argument");
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf din, dout, dafc;
    GateArgs A{};
    A.iq = iq;
    A.row_stride = row_stride;
    A.n = n_samples;
    A.fmt = in_fmt;
    A.fs = sample_rate;
    A.out = out;
    A.afc = afc;
    const int64_t used = n_samples < kGateFft ? n_samples : kGateFft;  // only the first 2048 samples of a row matter
    if (!device_pointers) {
        // upload just the leading samples of every row
        const size_t eb = fmt_bytes(in_fmt);
        if ((rc = din.alloc((size_t)rows * (used ? used : 1) * eb)) || (rc = dout.alloc((size_t)rows * kGateOut * 8)) ||
            (rc = dafc.alloc((size_t)rows * 8)))
            return rc;
        if (used)
            HIP_TRY(hipMemcpy2D(din.p, used * eb, iq, row_stride * eb, used * eb, rows, hipMemcpyHostToDevice));
        A.iq = din.p;
        A.row_stride = used;
        A.out = dout.as<double>();
        A.afc = dafc.as<double>();
    }
    hipLaunchKernelGGL(k_gate, dim3(rows), dim3(kFinishThreads), 0, device_pointers ? g_cur_stream : nullptr, A);
    HIP_TRY(hipGetLastError());
    if (!device_pointers) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, dout.p, (size_t)rows * kGateOut * 8, hipMemcpyDeviceToHost));
        if (afc) HIP_TRY(hipMemcpy(afc, dafc.p, (size_t)rows * 8, hipMemcpyDeviceToHost));
    }
    return TDM_OK;
}

// ---- scanner heuristics: TetraSignalDetector.calculate_power / detect_tetra_modulation /
// detect_sync_pattern (tetraear/signal/scanner.py:42-147) on rows of complex128 host samples
int tdm_detect(const double *x, int64_t n, int32_t rows, double sample_rate, double *out, int32_t device)
{
    if (!out || rows < 1 || n < 0 || (n > 0 && !x) || !(sample_rate > 0)) return fail(TDM_ERR_INVALID, "bad argument");
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf dx, dang, dbits, dout;
    const size_t nn = (size_t)rows * (n ? n : 1);
    if ((rc = dx.alloc(nn * 16)) || (rc = dang.alloc(nn * 8)) || (rc = dbits.alloc(nn)) || (rc = dout.alloc((size_t)rows * kDetectOut * 8)))
        return rc;
    if (n) HIP_TRY(hipMemcpy(dx.p, x, (size_t)rows * n * 16, hipMemcpyHostToDevice));
    DetectArgs A{dx.as<double>(), n, sample_rate, -85.0, dang.as<double>(), dbits.as<uint8_t>(), dout.as<double>()};
    hipLaunchKernelGGL(k_detect, dim3(rows), dim3(kFinishThreads), 0, 0, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)rows * kDetectOut * 8, hipMemcpyDeviceToHost));
    return TDM_OK;
}

// ---- burst sync: TetraDecoder.find_sync on the hard symbols, batched ---------------
namespace {
// Scratch of the device-pointer form, one buffer per (device, stream), process-wide: launches on one stream are ordered
// whichever thread issues them, so they can share it.  The mutex is held from the look-up to the end of the enqueue, so a
// buffer is never replaced between another caller's look-up and its launches; an entry goes away with its stream
// (plan_free -> sync_scratch_release), so a long-lived thread that creates and destroys plans does not pile them up.
struct SyncScratch { void *p = nullptr; size_t bytes = 0; };
struct SyncScratchMap {
    std::mutex mu;
    std::map<std::pair<int, hipStream_t>, SyncScratch> m;
    ~SyncScratchMap()   // process exit: give the buffers back (errors ignored: the runtime may already be gone)
    {
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) return;
        for (auto &kv : m)
            if (kv.second.p && hipSetDevice(kv.first.first) == hipSuccess) (void)hipFree(kv.second.p);
        (void)hipSetDevice(cur);
    }
};
SyncScratchMap g_sync_scratch;
}  // namespace

// (the caller has made `device` current and synchronised `st`)
static void sync_scratch_release(int device, hipStream_t st)
{
    std::lock_guard<std::mutex> lk(g_sync_scratch.mu);
    auto it = g_sync_scratch.m.find(std::make_pair(device, st));
    if (it == g_sync_scratch.m.end()) return;
    if (it->second.p) (void)hipFree(it->second.p);
    g_sync_scratch.m.erase(it);
}

int tdm_find_sync(const uint8_t *units, int64_t row_stride, const int32_t *n_units, int32_t rows, int32_t from_bits,
                  double threshold, int32_t max_pos, int32_t *positions, int32_t *n_pos, double *max_corr,
                  int32_t device_pointers, int32_t device)
{
    if (!units || !n_units || !positions || !n_pos || !max_corr || rows < 1 || max_pos < 1 || row_stride < 0)
        return fail(TDM_ERR_INVALID, "bad argument");
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf du, dn, dp, dnp, dmc, dcnt;
    const uint8_t *u = units;
    const int32_t *nu = n_units;
    int32_t *pp = positions, *np_ = n_pos;
    double *mc = max_corr;
    const int unit_bits = (from_bits & 1) ? 1 : 2;
    int64_t max_units = row_stride;   // device pointers: the counts stay on the device, rows are bounded by their stride
    if (!device_pointers) {
        max_units = 0;
        for (int r = 0; r < rows; ++r) {
            const int64_t nr = (from_bits & 2) ? (n_units[r] > 0 ? n_units[r] - 1 : 0) : n_units[r];
            if (nr < 0 || nr > row_stride) return fail(TDM_ERR_INVALID, "n_units out of range");
            if (nr > max_units) max_units = nr;
        }
    }
    const int64_t max_bits = unit_bits * max_units + 1;
    hipStream_t st = device_pointers ? g_cur_stream : nullptr;
    if (!device_pointers) {
        if ((rc = du.alloc((size_t)rows * row_stride + 1)) || (rc = dn.alloc(rows * 4)) ||
            (rc = dp.alloc((size_t)rows * max_pos * 4)) || (rc = dnp.alloc(rows * 4)) || (rc = dmc.alloc(rows * 8)))
            return rc;
        HIP_TRY(hipMemcpy(du.p, units, (size_t)rows * row_stride, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dn.p, n_units, rows * 4, hipMemcpyHostToDevice));
        u = du.as<uint8_t>(); nu = dn.as<int32_t>(); pp = dp.as<int32_t>(); np_ = dnp.as<int32_t>(); mc = dmc.as<double>();
    }
    // scratch for the per-position counts: for the device-pointer form (asynchronous: it must outlive the call) one
    // buffer per (device, stream) -- launches on one stream are ordered, so they can share it; another device or another
    // stream gets its own; it goes away with the plan that owns the stream -- released with the call otherwise
    uint16_t *cnt = nullptr;
    const size_t cnt_bytes = (size_t)rows * max_bits * 2;
    std::unique_lock<std::mutex> scratch_lock(g_sync_scratch.mu, std::defer_lock);
    if (device_pointers) {
        scratch_lock.lock();   // held until both launches are enqueued
        SyncScratch &sc = g_sync_scratch.m[std::make_pair((int)device, st)];
        if (sc.bytes < cnt_bytes) {
            if (sc.p) {
                HIP_TRY(hipStreamSynchronize(st));   // the only stream that ever used this buffer
                (void)hipFree(sc.p);
                sc.p = nullptr;
                sc.bytes = 0;
            }
            HIP_TRY(hipMalloc(&sc.p, cnt_bytes));
            sc.bytes = cnt_bytes;
        }
        cnt = (uint16_t *)sc.p;
    } else {
        if ((rc = dcnt.alloc(cnt_bytes))) return rc;
        cnt = dcnt.as<uint16_t>();
    }
    hipLaunchKernelGGL(k_sync_count, dim3((unsigned)((max_bits + 255) / 256), rows), dim3(256), 0, st, u, row_stride, nu,
                       from_bits, max_bits, cnt);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_sync_walk, dim3((rows + 63) / 64), dim3(64), 0, st, cnt, nu, from_bits, max_bits,
                       rows, threshold, pp, max_pos, np_, mc);
    HIP_TRY(hipGetLastError());
    if (!device_pointers) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(positions, dp.p, (size_t)rows * max_pos * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(n_pos, dnp.p, rows * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(max_corr, dmc.p, rows * 8, hipMemcpyDeviceToHost));
    }
    return TDM_OK;
}

}  // extern "C"

// ---- tetra-mode channeliser (oversampled polyphase DFT filter bank) ------------------------------------
namespace {
// Kaiser-windowed sinc prototype, cutoff at half the output rate, unit DC gain (oracle/pfb_np.py prototype)
static double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 64; ++k) {
        term *= (x / (2.0 * k)) * (x / (2.0 * k));
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}
static std::vector<float> pfb_prototype(int M, int D, int P)
{
    const int L = M * P;
    const double beta = 8.0, fc = 0.5 / D;
    std::vector<double> h(L);
    double sum = 0;
    for (int i = 0; i < L; ++i) {
        const double n = i - (L - 1) / 2.0;
        const double a = 2.0 * fc * n;
        const double sinc = std::fabs(a) < 1e-12 ? 1.0 : std::sin(M_PI * a) / (M_PI * a);
        const double r = 2.0 * i / (L - 1) - 1.0;
        const double w = bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / bessel_i0(beta);
        h[i] = 2 * fc * sinc * w;
        sum += h[i];
    }
    std::vector<float> out(L);
    for (int i = 0; i < L; ++i) out[i] = (float)(h[i] / sum);
    return out;
}
struct PfbTables {
    float *h = nullptr;
    float2 *tw = nullptr;
};
static std::mutex g_pfb_mu;
static std::map<std::tuple<int, int, int, int>, PfbTables> g_pfb_cache;  // (device, M1, M2, D) -> tables (kept)

template <int M1, int M2, int P, int TB, int WGS>
int launch_pfb(int device, const void *iq, int fmt, int64_t n_in, int D, float2 *out, int64_t n_out, int64_t pitch,
               int n_streams, hipStream_t st, bool sync)
{
    constexpr int M = M1 * M2, L = M * P;
    PfbTables tb;
    {
        std::lock_guard<std::mutex> lk(g_pfb_mu);
        auto key = std::make_tuple(device, M1, M2, D);
        auto it = g_pfb_cache.find(key);
        if (it == g_pfb_cache.end()) {
            std::vector<float> h = pfb_prototype(M, D, P);
            std::vector<float2> tw((size_t)M1 * M1 + M + (size_t)M2 * M2);
            for (int k = 0; k < M1; ++k)
                for (int n = 0; n < M1; ++n) {
                    const double a = 2.0 * M_PI * ((k * n) % M1) / M1;
                    tw[k * M1 + n] = make_float2((float)std::cos(a), (float)std::sin(a));
                }
            for (int k1 = 0; k1 < M1; ++k1)
                for (int n2 = 0; n2 < M2; ++n2) {
                    const double a = 2.0 * M_PI * ((k1 * n2) % M) / M;
                    tw[M1 * M1 + k1 * M2 + n2] = make_float2((float)std::cos(a), (float)std::sin(a));
                }
            for (int k = 0; k < M2; ++k)
                for (int n = 0; n < M2; ++n) {
                    const double a = 2.0 * M_PI * ((k * n) % M2) / M2;
                    tw[M1 * M1 + M + k * M2 + n] = make_float2((float)std::cos(a), (float)std::sin(a));
                }
            HIP_TRY(hipMalloc(&tb.h, h.size() * 4));
            HIP_TRY(hipMalloc(&tb.tw, tw.size() * 8));
            HIP_TRY(hipMemcpy(tb.h, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(tb.tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
            g_pfb_cache[key] = tb;
        } else {
            tb = it->second;
        }
    }
    PfbParams Q{};
    Q.D = D;
    Q.fmt = fmt;
    Q.n_in = n_in;
    Q.n_out = n_out;
    Q.h = tb.h;
    Q.W1 = tb.tw;
    Q.WM = Q.W1 + M1 * M1;
    Q.W2 = Q.WM + M;
    Q.in_stride = n_in * (int64_t)fmt_bytes(fmt);
    Q.out_batch = (int64_t)M * pitch;
    const bool force_direct = debug_value("pfb_direct") == 1;
    if (!force_direct && D <= 4 * M) {
        const int64_t rounds = (n_out + TB - 1) / TB;
        // rounds per workgroup: `per_cu` workgroups per compute unit are resident (one of k_pfb_fft's 146 KB at M = 400), so the launch runs in ceil(workgroups / slots) waves of G rounds each plus a start-up of about
        // a third of a round per workgroup; the G that minimises that (32 streams x 263 rounds, one per CU: G = 3 -> 2816
        // workgroups = exactly 11 waves, 0.235 ms; round 2's G = 4 -> 8.25 waves, 0.244 ms)
        static int cu_count[64] = {0};   // (per device, asked once)
        int &cus = cu_count[device & 63];
        if (cus == 0) {
            hipDeviceProp_t prop;
            cus = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }
        auto pick_rounds = [&](int per_cu) {
            double best = 1e300;
            int G = 1;
            const int64_t slots = (int64_t)cus * per_cu;
            for (int g = 1; g <= 8; ++g) {
                const int64_t wgs = ((rounds + g - 1) / g) * n_streams;
                const double cost = (double)((wgs + slots - 1) / slots) * (g + 0.3);
                if (cost < best - 1e-9) { best = cost; G = g; }
            }
            if (debug_value("pfb_rounds") > 0) G = (int)debug_value("pfb_rounds");   // experiments
            return G;
        };
        Q.G = pick_rounds(1);
        const size_t lds = pfb_fft_lds<M1, M2, P, TB>(D) * sizeof(float2);
        if (lds <= 160 * 1024) {
            void (*kern)(const void *, cf32v *, int64_t, const PfbParams) = nullptr;
            const int nu = ((TB - 1) * D + L + 3) / 4;
            const unsigned threads = TB * M2;
            {
                const bool one = nu <= TB * M2;   // one prefetched unit per thread covers the window
                switch (fmt) {
                case TDM_CU8: kern = one ? k_pfb_fft<M1, M2, P, TB, 0, 1, WGS> : k_pfb_fft<M1, M2, P, TB, 0, 2, WGS>; break;
                case TDM_CS8: kern = one ? k_pfb_fft<M1, M2, P, TB, 1, 1, WGS> : k_pfb_fft<M1, M2, P, TB, 1, 2, WGS>; break;
                default: kern = one ? k_pfb_fft<M1, M2, P, TB, 2, 1, WGS> : k_pfb_fft<M1, M2, P, TB, 2, 2, WGS>; break;
                }
            }
            HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const unsigned blocks = (unsigned)((rounds + Q.G - 1) / Q.G);
            hipLaunchKernelGGL(kern, dim3(blocks, n_streams), dim3(threads), lds, st, iq, (cf32v *)out, pitch, Q);
            HIP_TRY(hipGetLastError());
            if (sync) HIP_TRY(hipStreamSynchronize(st));
            return TDM_OK;
        }
    }
    Q.T = M <= 128 ? 32 : 8;
    const size_t lds = ((size_t)(Q.T - 1) * D + L + 2 * (size_t)Q.T * (M + 1) + M1 * M1 + M + M2 * M2) * sizeof(float2);
    if (lds > 160 * 1024) return fail(TDM_ERR_UNSUPPORTED, "channeliser tile does not fit LDS");
    HIP_TRY(hipFuncSetAttribute((const void *)k_pfb<M1, M2, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned blocks = (unsigned)((n_out + Q.T - 1) / Q.T);
    hipLaunchKernelGGL((k_pfb<M1, M2, P>), dim3(blocks, n_streams), dim3(kPfbThreads), lds, st, iq, out, pitch, Q);
    HIP_TRY(hipGetLastError());
    if (sync) HIP_TRY(hipStreamSynchronize(st));
    return TDM_OK;
}
}  // namespace

extern "C" {

int tdm_channelise_batch(const void *iq, int32_t in_fmt, int64_t n_in, int32_t n_streams, int32_t M, int32_t D,
                          float *out, int64_t out_pitch, int64_t *n_out, int32_t device_pointers, int32_t device)
{
    if (!iq || !out || !n_out || n_in < 1 || D < 1 || n_streams < 1 || n_streams > 65535 ||
        (out_pitch != 0 && out_pitch < (n_in + D - 1) / D) ||
        (in_fmt != TDM_CU8 && in_fmt != TDM_CS8 && in_fmt != TDM_CF32))
        return fail(TDM_ERR_INVALID, "bad argument");
    int rc = use_device(device);
    if (rc) return rc;
    const int64_t no = (n_in + D - 1) / D;
    *n_out = no;
    const int64_t pitch = out_pitch ? out_pitch : no;
    DevBuf din, dout;
    const void *src = iq;
    float2 *dst = (float2 *)out;
    const size_t ib = (size_t)n_streams * n_in * fmt_bytes(in_fmt), ob = (size_t)n_streams * M * pitch * sizeof(float2);
    if (!device_pointers) {
        if ((rc = din.alloc(ib)) || (rc = dout.alloc(ob))) return rc;
        HIP_TRY(hipMemcpy(din.p, iq, ib, hipMemcpyHostToDevice));
        src = din.p;
        dst = dout.as<float2>();
    }
    const bool sync = !device_pointers;
    switch (M) {
    // device pointers: enqueue on the default stream and return (tdm_dev_sync waits)
#define TDM_PFB96 16, 4
#define TDM_PFB72 48, 2
#define TDM_PFB128 16, 4
#define TDM_PFB80 32, 3
    case 96: rc = launch_pfb<8, 12, 3, TDM_PFB96>(device, src, in_fmt, n_in, D, dst, no, pitch, n_streams, device_pointers ? g_cur_stream : nullptr, sync); break;
    case 72: rc = launch_pfb<8, 9, 3, TDM_PFB72>(device, src, in_fmt, n_in, D, dst, no, pitch, n_streams, device_pointers ? g_cur_stream : nullptr, sync); break;
    case 80: rc = launch_pfb<8, 10, 3, TDM_PFB80>(device, src, in_fmt, n_in, D, dst, no, pitch, n_streams, device_pointers ? g_cur_stream : nullptr, sync); break;
    case 128: rc = launch_pfb<8, 16, 3, TDM_PFB128>(device, src, in_fmt, n_in, D, dst, no, pitch, n_streams, device_pointers ? g_cur_stream : nullptr, sync); break;
    case 400: rc = launch_pfb<20, 20, 3, 32, 1>(device, src, in_fmt, n_in, D, dst, no, pitch, n_streams, device_pointers ? g_cur_stream : nullptr, sync); break;
    default: return fail(TDM_ERR_UNSUPPORTED, "channeliser built for M in {72, 80, 96, 128, 400}");
    }
    if (rc) return rc;
    if (!device_pointers) HIP_TRY(hipMemcpy(out, dout.p, ob, hipMemcpyDeviceToHost));
    return TDM_OK;
}

int tdm_occupancy_gate(const float *chan, int64_t pitch, int32_t n_streams, int32_t M, int64_t n_out, double chan_rate,
                       double snr_db, double min_dbfs, float *stats, uint8_t *flags, int32_t *row_list, int32_t *n_rows,
                       int32_t *n_soft, int32_t device_pointers, int32_t device)
{
    if (!chan || !stats || !flags || !row_list || !n_rows || n_streams < 1 || M < 4 || M > kOccMaxM || !(chan_rate > 0))
        return fail(TDM_ERR_INVALID, "bad argument");
    if (n_out < kOccFft || pitch < kOccFft)
        return fail(TDM_ERR_UNSUPPORTED, "the occupancy gate looks at the first 256 samples of every channel row");
    int rc = use_device(device);
    if (rc) return rc;
    const int64_t rows = (int64_t)n_streams * M;
    if (rows > (int64_t(1) << 30)) return fail(TDM_ERR_INVALID, "too many rows");
    // the 25 kHz around the centre in bins of chan_rate / 256 (ui/modern.py:1948-1953)
    const int bandwidth_bins = (int)(25000.0 / (chan_rate / kOccFft));
    OccArgs A{};
    A.bin_lo = std::max(0, kOccFft / 2 - bandwidth_bins / 2);
    A.bin_hi = std::min(kOccFft, kOccFft / 2 + bandwidth_bins / 2);
    if (A.bin_hi <= A.bin_lo) return fail(TDM_ERR_INVALID, "channel rate too high for a 25 kHz band of 256-point bins");
    DevBuf dch, dst, dfl, dli, dn, dns;
    hipStream_t st = device_pointers ? g_cur_stream : nullptr;
    A.chan = (const float2 *)chan;
    A.pitch = pitch;
    A.rows = (int32_t)rows;
    A.stats = (float2 *)stats;
    OccListArgs L{};
    L.flags = flags;
    L.row_list = row_list;
    L.n_rows = n_rows;
    L.n_soft = n_soft;
    if (!device_pointers) {
        if ((rc = dch.alloc((size_t)rows * kOccFft * 8)) || (rc = dst.alloc((size_t)rows * 8)) || (rc = dfl.alloc((size_t)rows)) ||
            (rc = dli.alloc((size_t)rows * 4)) || (rc = dn.alloc(4)) || (n_soft && (rc = dns.alloc((size_t)rows * 4))))
            return rc;
        HIP_TRY(hipMemcpy2D(dch.p, kOccFft * 8, chan, pitch * 8, kOccFft * 8, rows, hipMemcpyHostToDevice));
        if (n_soft) HIP_TRY(hipMemcpy(dns.p, n_soft, (size_t)rows * 4, hipMemcpyHostToDevice));
        A.chan = dch.as<float2>();
        A.pitch = kOccFft;
        A.stats = dst.as<float2>();
        L.flags = dfl.as<uint8_t>();
        L.row_list = dli.as<int32_t>();
        L.n_rows = dn.as<int32_t>();
        L.n_soft = n_soft ? dns.as<int32_t>() : nullptr;
    }
    L.stats = A.stats;
    L.M = M;
    L.snr_db = (float)snr_db;
    L.min_dbfs = (float)min_dbfs;
    L.peak_db = 3.0f;
    A.n_rows = L.n_rows;
    hipLaunchKernelGGL(k_occ_spectrum, dim3((unsigned)((rows + kOccRowsPerWg - 1) / kOccRowsPerWg)), dim3(256), 0, st, A);
    hipLaunchKernelGGL(k_occ_list, dim3((unsigned)n_streams), dim3(512), 0, st, L);
    HIP_TRY(hipGetLastError());
    if (!device_pointers) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(stats, dst.p, (size_t)rows * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(flags, dfl.p, (size_t)rows, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(row_list, dli.p, (size_t)rows * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(n_rows, dn.p, 4, hipMemcpyDeviceToHost));
        if (n_soft) HIP_TRY(hipMemcpy(n_soft, dns.p, (size_t)rows * 4, hipMemcpyDeviceToHost));
    }
    return TDM_OK;
}

int tdm_channelise(const void *iq, int32_t in_fmt, int64_t n_in, int32_t M, int32_t D, float *out, int64_t *n_out,
                   int32_t device_pointers, int32_t device)
{
    return tdm_channelise_batch(iq, in_fmt, n_in, 1, M, D, out, 0, n_out, device_pointers, device);
}

// ---- introspection (no device needed) -----------------------------------------------------------
int tdm_design_dump(double sample_rate, int64_t n_samples, double *sos, double *soszi, double *b, double *a,
                    double *zi, int32_t *q, double *rate_dec)
{
    if (!(sample_rate > 0)) return fail(TDM_ERR_INVALID, "bad sample_rate");
    const int qq = decimation_factor(sample_rate);
    const bool dec = qq > 1 && n_samples > kEdgeSos;
    const double rate = dec ? sample_rate / qq : sample_rate;
    if (sos) std::memset(sos, 0, 24 * sizeof(double));
    if (soszi) std::memset(soszi, 0, 8 * sizeof(double));
    if (qq > 1) {
        Sos4 s = design_cheby1_8(0.05, 0.8 / qq);
        if (sos) std::memcpy(sos, s.sos, sizeof(s.sos));
        if (soszi) std::memcpy(soszi, s.zi, sizeof(s.zi));
    }
    Tf4 t = design_butter4(butter_cutoff(25000.0, rate));
    if (b) std::memcpy(b, t.b, sizeof(t.b));
    if (a) std::memcpy(a, t.a, sizeof(t.a));
    if (zi) std::memcpy(zi, t.zi, sizeof(t.zi));
    if (q) *q = dec ? qq : 1;
    if (rate_dec) *rate_dec = rate;
    return TDM_OK;
}

int tdm_design_butter(double bandwidth, double fs, double *b, double *a, double *zi)
{
    if (!(fs > 0) || !b || !a || !zi) return fail(TDM_ERR_INVALID, "bad argument");
    Tf4 t = design_butter4(butter_cutoff(bandwidth, fs));
    std::memcpy(b, t.b, sizeof(t.b));
    std::memcpy(a, t.a, sizeof(t.a));
    std::memcpy(zi, t.zi, sizeof(t.zi));
    return TDM_OK;
}

}  // extern "C"
