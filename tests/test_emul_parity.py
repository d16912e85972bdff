"""CPU: the HIP kernel BODIES (compiled for the host, lock-step emulated; tests/emul) against the
golden vectors of the imported reference.  This checks the device code's table and index logic
without a GPU; the same comparison runs on the real device in test_gpu_parity.py."""
import numpy as np
import pytest

from tests.emul import emul
from tests.golden_cases import CASES, TIMING_DEGENERATE, case_c128, case_cu8

SOFT_TOL = 1e-10   # relative to max|soft|; north-star tolerance is 1e-5
BIG = {"q41_10M_1M", "noise_2400_256k"}
# (the eight channelised carriers of the shared multicarrier input take 13 s each in lock-step emulation: three of them here --
#  both ends and the middle of the grid --, all eight in the oracle tier and on the device)
BIG |= {"mc8_k1", "mc8_k2", "mc8_k4", "mc8_k5", "mc8_k6"}


def check(name, hard, soft, n_soft, gold_process):
    g_hard = gold_process[name + "__hard"]
    g_soft = gold_process[name + "__soft"]
    ns = int(n_soft[0])
    h = hard[0, :max(ns - 1, 0)]
    s = soft[0, :ns]
    if name in TIMING_DEGENERATE:
        m = min(len(h), len(g_hard))
        np.testing.assert_array_equal(h[:m], g_hard[:m])
        return
    assert ns == len(g_soft)
    np.testing.assert_array_equal(h, g_hard)
    if ns:
        scale = np.max(np.abs(g_soft)) or 1.0
        assert np.max(np.abs(s - g_soft)) <= SOFT_TOL * scale


@pytest.mark.parametrize("name", sorted(n for n in CASES if n not in BIG))
def test_emul_process_cases(name, gold_process):
    c = CASES[name]
    if c["n"] == 0:
        pytest.skip("n == 0 is handled by the host shim (processor.py:239-241)")
    pre = None
    if c["kind"] == "c128":
        x = case_c128(c)
        hard, soft, n_soft, bp, mm = emul.process(c["fs"], x, "cf64", c["n"], freq_offset=[c["foff"]])
    else:
        u8 = case_cu8(c)
        if "pre_shift" in c:
            pre = [c["pre_shift"]]
        hard, soft, n_soft, bp, mm = emul.process(c["fs"], u8, "cu8", c["n"], pre_shift=pre,
                                                  freq_offset=[c["foff"]])
    check(name, hard, soft, n_soft, gold_process)


def test_emul_q41_long(gold_process):
    """q = 41: filter memory (8246 samples) spans several 2048-sample blocks, so the cross-block
    carry series needs more than one term."""
    c = CASES["q41_10M_1M"]
    n = 262144 + 4321
    u8 = case_cu8(c)[: 2 * n]
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    o = OracleSignalProcessor(c["fs"])
    ref_hard = o.process(synth.cu8_to_c128(u8), c["foff"])
    hard, soft, n_soft, bp, mm = emul.process(c["fs"], u8, "cu8", n, freq_offset=[c["foff"]])
    ns = int(n_soft[0])
    assert ns == len(o.symbols)
    np.testing.assert_array_equal(hard[0, :ns - 1], ref_hard)
    assert np.max(np.abs(soft[0, :ns] - o.symbols)) <= SOFT_TOL * np.max(np.abs(o.symbols))


def test_emul_multirow_shared_and_formats(gold_process):
    """rows>1, shared input stream with per-row pre-shift, and cs8/cf32/cf64 formats."""
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    n = 9000
    u8 = synth.noise_cu8(n, 4242)
    x = synth.cu8_to_c128(u8)
    shifts = [-25000.0, 0.0, 37500.0]
    foffs = [0.0, 1171.875, -500.0]
    hard, soft, n_soft, bp, mm = emul.process(2.4e6, u8, "cu8", n, rows=3, stride=0, pre_shift=shifts,
                                              freq_offset=foffs)
    for r in range(3):
        o = OracleSignalProcessor(2.4e6)
        ref = o.process(o.frequency_shift(x, shifts[r]), foffs[r])
        ns = int(n_soft[r])
        assert ns == len(o.symbols) and bp[r] == o.best_phase
        np.testing.assert_array_equal(hard[r, :ns - 1], ref)
        assert np.max(np.abs(soft[r, :ns] - o.symbols)) <= SOFT_TOL * np.max(np.abs(o.symbols))
    # separate rows, other formats
    s8 = (u8.astype(np.int16) - 128).astype(np.int8)
    xs = (s8[0::2].astype(np.float64) + 1j * s8[1::2].astype(np.float64)) / 128.0
    o = OracleSignalProcessor(1.8e6)
    ref = o.process(xs, 250.0)
    for fmt, arr in (("cs8", s8), ("cf32", xs.astype(np.complex64)), ("cf64", xs)):
        xin = np.concatenate([arr, arr])
        hard, soft, n_soft, bp, mm = emul.process(1.8e6, xin, fmt, n, rows=2, freq_offset=[250.0, 250.0])
        for r in range(2):
            ns = int(n_soft[r])
            np.testing.assert_array_equal(hard[r, :ns - 1], ref)


def test_emul_filter_stage_conditioning(gold_stages):
    """filter_signal at the full 2.4 MHz rate: a very narrow order-4 Butterworth.  The device
    evaluates it as two biquads because companion-form states have ~1e4 transient growth, which
    the blocked evaluation would square (2e-8 error); with biquads the result sits at the
    reference's own rounding-noise level (scipy tf-form vs exact: ~4e-11)."""
    from tetraear_amd import synth
    g = gold_stages
    x = synth.cu8_to_c128(synth.noise_cu8(4000, int(g["x4000_seed"][0])))
    for bw, fs, key, tol in [(25000, 2.4e6, "filter_default", 1e-9), (50000, 2.4e6, "filter_bw50k", 1e-9),
                             (25000, 240000.0, "filter_240k", 1e-12), (100.0, 2.4e6, "filter_clamp_lo", 1e-9),
                             (25000, 20000.0, "filter_clamp_hi", 1e-8)]:
        y = emul.zp_stage(1, x, bandwidth=bw, fs=fs)
        assert np.max(np.abs(y - g[key])) <= tol, key
    for q in (7, 10, 41):
        y = emul.zp_stage(0, x, q=q)
        assert np.max(np.abs(y - g[f"decimate_q{q}"])) <= 1e-12 * np.max(np.abs(g[f"decimate_q{q}"]))


def test_emul_resample(gold_stages):
    """SignalProcessor.resample (processor.py:35-49): direct-DFT kernels vs scipy goldens + scipy's
    bin bookkeeping on odd/even, up/down, tiny lengths (incl. the empty-slice quirk at N == 2)."""
    from oracle.oracle import resample_np
    from tetraear_amd import synth
    g = gold_stages
    x = synth.cu8_to_c128(synth.noise_cu8(4000, int(g["x4000_seed"][0])))
    assert np.max(np.abs(emul.resample(x[:1000], 500) - g["resample_1200k"])) < 1e-13
    assert np.max(np.abs(emul.resample(x[:301], int(301 * 3.0e6 / 2.4e6)) - g["resample_up"])) < 1e-13
    for n, num in [(16, 8), (16, 32), (15, 7), (15, 31), (7, 7), (2, 5), (3, 2), (4, 2), (2, 1), (1, 4), (64, 63),
                   (63, 64), (2, 2), (2, 4), (5, 2)]:
        ref = resample_np(x[:n], 1.0, num / n + 1e-12)
        assert len(ref) == num
        assert np.max(np.abs(emul.resample(x[:n], num) - ref)) < 1e-13, (n, num)


@pytest.mark.parametrize("n", [2, 3, 4, 5, 8, 9, 10, 12, 16, 20, 25])
def test_small_dft_templates_match_numpy(n):
    """Register-resident DFTs of the channeliser (small_dft.hpp: radix 2/3/4/5 butterflies, Cooley-Tukey
    and Good-Thomas composites, constexpr twiddles) against numpy, sign +, unnormalised."""
    rng = np.random.default_rng(n)
    for _ in range(4):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        ref = np.fft.ifft(x.astype(np.complex128)) * n
        assert np.max(np.abs(emul.small_dft(x) - ref)) < 4e-7 * np.max(np.abs(ref)) * np.sqrt(n)


def test_emul_random_lengths_and_rates_vs_oracle():
    """Block-boundary coverage: seeded random chunk lengths (incl. lengths around multiples of the block sizes
    and the fall-back thresholds 27 / 15), sample rates and AFC offsets; kernel bodies in CPU emulation against
    the C oracle.  (tools/sweep_emul.py / tools/sweep_gpu.py run the same sweep open-ended: 350 CPU and 19 291
    GPU cases without a mismatch in round 1.)"""
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    rng = np.random.default_rng(2026)
    rates = [2.4e6, 2.4e6, 1.8e6, 2.048e6, 960000.0, 480000.0, 240000.0, 72000.0]
    specials = [27, 28, 150, 160, 161, 5119, 5120, 5121, 5271, 10240, 20473, 20480, 20481, 2048, 4097, 6144]
    for it in range(36):
        fs = rates[rng.integers(len(rates))]
        n = int(specials[it % len(specials)]) if it % 2 == 0 else int(rng.integers(1, 12000))
        f = 0.0 if it % 3 == 0 else float(rng.uniform(-8000, 8000))
        u8 = synth.noise_cu8(n, 9000 + it)
        ref = OracleSignalProcessor(fs)
        r = ref.process(synth.cu8_to_c128(u8), f)
        hard, soft, n_soft, bp, mm = emul.process(fs, u8, "cu8", n, 1, freq_offset=np.array([f]))
        ns = int(n_soft[0])
        assert ns == len(ref.symbols), (fs, n, f)
        np.testing.assert_array_equal(hard[0, :max(ns - 1, 0)], r)
        if ns:
            assert np.max(np.abs(soft[0, :ns] - ref.symbols)) <= 1e-10 * (np.max(np.abs(ref.symbols)) or 1.0), (fs, n, f)



def test_emul_fast_pre_shift_against_exact_and_oracle():
    """Plan option "fast_pre_shift": the input-rate shift's phase as the ideal ramp from an exactly anchored sample per lane,
    against the path that reproduces the reference's rounding of theta sample by sample, and against the oracle's
    process(frequency_shift(x, f_k)): hard decisions equal, soft symbols within 1e-9 (exact path: 1e-10)."""
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    n = 60000
    u8 = synth.noise_cu8(n, 4243)
    x = synth.cu8_to_c128(u8)
    shifts = [-787500.0, 12500.0, 612500.0]
    foffs = [0.0, 1171.875, -500.0]
    a = emul.process(2.4e6, u8, "cu8", n, rows=3, stride=0, pre_shift=shifts, freq_offset=foffs)
    with emul.fast_pre_shift():
        b = emul.process(2.4e6, u8, "cu8", n, rows=3, stride=0, pre_shift=shifts, freq_offset=foffs)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_array_equal(a[3], b[3])
    for r in range(3):
        o = OracleSignalProcessor(2.4e6)
        ref = o.process(o.frequency_shift(x, shifts[r]), foffs[r])
        ns = int(b[2][r])
        np.testing.assert_array_equal(b[0][r, :ns - 1], ref)
        scale = np.max(np.abs(o.symbols))
        assert np.max(np.abs(b[1][r, :ns] - o.symbols)) <= 1e-9 * scale
        assert np.max(np.abs(a[1][r, :ns] - o.symbols)) <= 1e-10 * scale
        assert np.max(np.abs(a[1][r, :ns] - b[1][r, :ns])) > 0      # (it IS another phase)


def test_emul_time_batched_shared_stream():
    """Plan option "rows_per_chunk" (config 3 in time batches): T consecutive chunks of ONE stream x C carriers in one call --
    plan row r reads input row r // C with its own pre-shift and offset, its phase starting at its chunk's first sample;
    every row equals the oracle's p.process(p.frequency_shift(x_chunk, f), foff)."""
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    n, T, Cc = 7000, 3, 2
    u8 = synth.noise_cu8(n * T, 4343)
    x = synth.cu8_to_c128(u8)
    shifts = np.tile([-37500.0, 25000.0], T)
    foffs = np.array([0.0, 1171.875, -500.0, 0.0, 250.0, -1171.875])
    for fmt, arr in (("cu8", u8), ("cf64", x)):
        hard, soft, n_soft, bp, mm = emul.process(2.4e6, arr, fmt, n, rows=T * Cc, stride=n, pre_shift=shifts,
                                                  freq_offset=foffs, rows_per_chunk=Cc)
        for r in range(T * Cc):
            o = OracleSignalProcessor(2.4e6)
            ref = o.process(o.frequency_shift(x[(r // Cc) * n:(r // Cc + 1) * n], shifts[r]), foffs[r])
            ns = int(n_soft[r])
            assert ns == len(o.symbols) and bp[r] == o.best_phase, (fmt, r)
            np.testing.assert_array_equal(hard[r, :ns - 1], ref)
            assert np.max(np.abs(soft[r, :ns] - o.symbols)) <= SOFT_TOL * np.max(np.abs(o.symbols))
    emul.lib().emu_rows_per_chunk(1)
