"""TETRA mode (RRC matched filter + feed-forward timing + Farrow + differential quadrant slicer).

No reference oracle exists for this mode -- "parity unpinned".  Checks:
  * the fp64 numpy definition (oracle/tetra_np.py) recovers the transmitted dibits of the
    synthetic generator without error (CPU);
  * the fp32 HIP kernels give the same hard decisions as that definition and soft symbols within
    1e-4 (GPU), and zero symbol errors against the transmitted data."""
import numpy as np
import pytest

from oracle import tetra_np
from tetraear_amd import synth


def make_signal(n, fs, seed, toff=0.0, coff=0.0, snr_db=20.0):
    x, dib = synth.dqpsk_baseband(n, fs, seed, timing_offset=toff)
    rng = np.random.default_rng(seed + 100)
    sps = fs / 18000.0
    sigma2 = sps / 10 ** (snr_db / 10)
    x = x + np.sqrt(sigma2 / 2) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    x = x * np.exp(2j * np.pi * coff * np.arange(n) / fs)
    return x.astype(np.complex64), dib


def best_ber(hard, dib, edge=0):
    """symbol error rate at the best alignment; `edge` symbols at each end of the chunk are ignored
    (a chunk is processed statelessly, so filters are still filling there)."""
    best = (1.0, -1)
    for lag in range(0, 40):
        m = min(len(hard), len(dib) - lag)
        if m < 50:
            continue
        e = float(np.mean(hard[edge:m - edge] != dib[lag + edge:lag + m - edge]))
        if e < best[0]:
            best = (e, lag)
    return best


CASES = [(72000.0, 8192, 1, 0.0, 0.0, 20.0), (72000.0, 8192, 2, 0.3, 0.0, 20.0), (72000.0, 16384, 3, -0.41, 40.0, 20.0),
         (75000.0, 8192, 4, 0.1, -120.0, 15.0), (80000.0, 12000, 5, 0.25, 60.0, 20.0), (54000.0, 6000, 6, -0.2, 0.0, 20.0)]


@pytest.mark.parametrize("fs,n,seed,toff,coff,snr", CASES)
def test_definition_recovers_transmitted_symbols(fs, n, seed, toff, coff, snr):
    x, dib = make_signal(n, fs, seed, toff, coff, snr)
    hard, soft, info = tetra_np.demod(x.astype(np.complex128), fs)
    assert len(hard) > 0.9 * n / (fs / 18000.0) - 20
    ber, lag = best_ber(hard, dib)
    assert ber == 0.0, (ber, lag)
    assert abs(info["tau"][len(info["tau"]) // 2] + toff - round(info["tau"][len(info["tau"]) // 2] + toff)) < 0.05


@pytest.mark.gpu
def test_gpu_tetra_matches_definition_and_transmitted():
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    for fs, n, seed, toff, coff, snr in CASES:
        rows = 3
        xs, dibs = zip(*[make_signal(n, fs, seed * 10 + r, toff + 0.05 * r, coff, snr) for r in range(rows)])
        bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA)
        hards, softs, timing, margin = bd.process(np.concatenate(xs))
        for r in range(rows):
            ref_hard, ref_dd, info = tetra_np.demod(xs[r].astype(np.complex128), fs)
            assert len(softs[r]) == info["n_sym"], (fs, r)
            np.testing.assert_array_equal(hards[r], ref_hard)
            scale = np.max(np.abs(info["sym"]))
            # (matched filter in split-bf16 products with fp32 accumulation: at most 6.2e-6 of the largest symbol over 22 065 random carriers)
            assert np.max(np.abs(softs[r] - info["sym"])) < 1e-5 * scale
            # differential detection at Es/N0 = 15 dB has a raw symbol error rate of order 1e-3
            assert best_ber(hards[r], dibs[r])[0] <= (0.0 if snr >= 20.0 else 3e-3)
            assert abs(timing[r] / 1000.0 - info["tau"][len(info["tau"]) // 2]) < 2e-3
            assert abs(margin[r] - info["margin"]) < 1e-3
        bd.close()


@pytest.mark.gpu
def test_gpu_tetra_noise_and_silence_do_not_crash():
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    n, fs = 4096, 72000.0
    rng = np.random.default_rng(0)
    x = np.concatenate([np.zeros(n), rng.standard_normal(n) + 1j * rng.standard_normal(n)]).astype(np.complex64)
    bd = BatchDemodulator(fs, n, 2, "cf32", mode=MODE_TETRA)
    hards, softs, timing, margin = bd.process(x)
    assert len(hards[0]) > 900 and np.all(hards[0] <= 3) and np.all(hards[1] <= 3)
    bd.close()


@pytest.mark.gpu
def test_gpu_tetra_final_passes_edge_cases():
    """The final passes (differential products, 4th-power estimate, decisions, margin) work on chunks of 2048 symbols with
    a separate path for a carrier's last chunk: symbol counts just below, at and above a multiple of 2048, and a silent
    carrier (every product exactly zero: decisions 0 and margin 0 by the definition's atan2(0, 0)), against the definition."""
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    fs = 72000.0
    seen = set()
    for n in list(range(8186, 8214, 3)) + [16384, 16388, 24580]:
        x, dib = make_signal(n, fs, 100 + n, 0.2, 30.0, 25.0)
        bd = BatchDemodulator(fs, n, 2, "cf32", mode=MODE_TETRA)
        hards, softs, timing, margin = bd.process(np.concatenate([x, np.zeros(n, np.complex64)]))
        bd.close()
        ref_hard, _, info = tetra_np.demod(x.astype(np.complex128), fs)
        assert len(softs[0]) == info["n_sym"]
        np.testing.assert_array_equal(hards[0], ref_hard)
        assert abs(margin[0] - info["margin"]) < 1e-3
        seen.add(info["n_sym"] % 2048)
        zh, _, zinfo = tetra_np.demod(np.zeros(n, np.complex128), fs)
        assert len(softs[1]) == zinfo["n_sym"] and not hards[1].any() and not zh.any()
        assert margin[1] == 0.0 and zinfo["margin"] == 0.0
    assert min(seen) <= 2 and max(seen) >= 2040   # (both sides of a chunk boundary were exercised)


def _end_of_chunk_contract(hard_dev, ns_dev, x, fs):
    """The contract at a chunk's end: the device forms the symbol instants in fp32 from fp32 timing
    estimates, the definition in fp64, so an instant within rounding of the bound t <= n - 3 may be kept by one and dropped
    by the other.  Allowed: a symbol COUNT that differs by at most one, and then only with the deciding instant at the
    bound; every decision both sides made must be equal.  Returns the count difference (device - definition)."""
    n = len(x)
    ref_hard, _, info = tetra_np.demod(x.astype(np.complex128), fs)
    diff = int(ns_dev) - int(info["n_sym"])
    assert abs(diff) <= 1, (fs, n, ns_dev, info["n_sym"])
    m = min(len(hard_dev), len(ref_hard))
    np.testing.assert_array_equal(hard_dev[:m], ref_hard[:m])
    if diff:
        t = info["t"]
        # device dropped the definition's last symbol: that instant sits just under the bound; device kept one more: the
        # definition's next instant (one symbol period on) sits just over it
        t_edge = t[-1] if diff < 0 else t[-1] + (t[-1] - t[-2])
        assert abs(t_edge - (n - 3.0)) < 5e-3, (fs, n, diff, t_edge)
    return diff


@pytest.mark.gpu
def test_gpu_tetra_end_of_chunk_symbol_count_bound():
    """(a) the carrier of the round-3 sweep whose last instant lies 1.3e-6 samples under the bound (fixture
    tests/golden/tetra_edge.npz, made by tests/golden/make_golden_tetra_edge.py): the device returns the definition's
    count or one fewer, all common decisions equal.  (b) a seeded slice built to land on the bound: timing offsets that
    put the symbol instants on whole samples, chunk lengths through every residue of the symbol period."""
    import os
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tetra_edge.npz"))
    x, fs = g["x"], float(g["fs"])
    bd = BatchDemodulator(fs, len(x), 1, "cf32", mode=MODE_TETRA)
    hards, softs, _, _ = bd.process(x)
    bd.close()
    assert int(g["n_sym"]) == 2817 and abs(float(g["t_last"]) - (len(x) - 3.0)) < 1e-4
    d = _end_of_chunk_contract(hards[0], len(softs[0]), x, fs)
    assert d in (0, -1)
    m = min(len(hards[0]), len(g["hard"]))
    np.testing.assert_array_equal(hards[0][:m], g["hard"][:m])   # (the committed decisions, not only today's definition)
    seen = {0: 0, 1: 0, -1: 0}
    for fs, toffs in ((72000.0, (0.0, 0.25, 0.5)), (108000.0, (0.0, 1.0 / 6.0, 0.5)), (144000.0, (0.0, 0.125))):
        sps = int(fs / 18000.0)
        for ti, toff in enumerate(toffs):
            n0 = 2900 + 37 * ti
            xfull, _ = make_signal(n0 + 2 * sps + 1, fs, 900 + ti, toff, 15.0, 25.0)
            for n in range(n0, n0 + 2 * sps + 1):
                xr = np.ascontiguousarray(xfull[:n])
                bd = BatchDemodulator(fs, n, 1, "cf32", mode=MODE_TETRA)
                hards, softs, _, _ = bd.process(xr)
                bd.close()
                seen[_end_of_chunk_contract(hards[0], len(softs[0]), xr, fs)] += 1
    assert sum(seen.values()) >= 80, seen


@pytest.mark.gpu
def test_gpu_tetra_every_rate_in_the_contract_and_nonfinite_input():
    """Channel rates whose RRC length is not one the kernel is instantiated for (45 / 50 / 60 kHz: 21, 23, 27 taps) run
    with the taps centred in the next length up and still equal the fp64 definition; a NaN / Inf sample in one
    carrier spoils that carrier only and never makes the symbol stage index outside its row."""
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    for fs in (45000.0, 50000.0, 60000.0, 36000.0, 144000.0):
        n = 6000
        x, dib = make_signal(n, fs, 77, 0.2, 30.0, 20.0)
        bd = BatchDemodulator(fs, n, 1, "cf32", mode=MODE_TETRA)
        hards, softs, timing, margin = bd.process(x)
        ref_hard, _, info = tetra_np.demod(x.astype(np.complex128), fs)
        np.testing.assert_array_equal(hards[0], ref_hard)
        # (at exactly 2 samples per symbol the square-law timing statistic has no margin: agreement with the
        # definition is required there, error-free reception from 2.5 samples per symbol up)
        assert best_ber(hards[0], dib, edge=8)[0] <= (0.0 if fs >= 45000.0 else 5e-3), fs
        bd.close()
    n, fs = 8192, 72000.0
    good, dib = make_signal(n, fs, 5, 0.1, 0.0, 20.0)
    bad = good.copy()
    bad[1000] = np.nan
    bad[5000] = np.inf
    bd = BatchDemodulator(fs, n, 4, "cf32", mode=MODE_TETRA)
    hards, softs, timing, margin = bd.process(np.concatenate([bad, good, bad, np.full(n, np.nan, np.complex64)]))
    assert best_ber(hards[1], dib, edge=8)[0] == 0.0
    assert all(np.all(h <= 3) for h in hards)
    # a carrier without a single comparable decision (every symbol NaN) reports the "no decision" margin, not the neutral
    # ratio of the slots past its last symbol (atan(1) = 0.785 before round 4)
    assert margin[3] > 3.0e38, margin[3]
    assert 0.0 < margin[1] < 0.8
    bd.close()


@pytest.mark.gpu
def test_gpu_tetra_input_scale():
    """Power-of-two scalings of the input (int16-range IQ passed as floats, very small IQ) give the same decisions, the
    same timing and margin, and soft symbols that are the unscaled ones times the factor: the split-bf16 matched filter
    and the 4th-power carrier-offset estimate (an 8th power of the amplitude) are both kept scale-free."""
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    fs, n = 72000.0, 16384
    x, dib = make_signal(n, fs, 31, 0.3, 60.0, 20.0)
    bd = BatchDemodulator(fs, n, 3, "cf32", mode=MODE_TETRA)
    scales = (1.0, 2.0 ** 15, 2.0 ** -22)
    hards, softs, timing, margin = bd.process(np.concatenate([(x * s).astype(np.complex64) for s in scales]))
    bd.close()
    assert best_ber(hards[0], dib, edge=8)[0] == 0.0
    for r, s in enumerate(scales):
        np.testing.assert_array_equal(hards[r], hards[0])
        np.testing.assert_allclose(softs[r] / s, softs[0], rtol=0, atol=2e-6 * np.max(np.abs(softs[0])))
        assert timing[r] == timing[0] and abs(margin[r] - margin[0]) < 1e-6


@pytest.mark.gpu
def test_gpu_tetra_instants_beyond_the_ring():
    """Symbol instants that leave the matched-filter ring in LDS (48 symbols below, 80 above their nominal positions at
    8 samples/symbol) take the kernel's direct path (filter outputs recomputed from the input).
    (a) A sampling-clock offset of -0.4 % walks the instants 65 symbols early over a 131 072-sample chunk; the fp64
        definition tracks the drift through its unwrap and the device follows it symbol for symbol.
    (b) No trackable clock offset reaches 80 symbols late, so two tones whose beat sits just off the symbol rate turn
        the timing statistic by 0.3 cycles per sub-block: the estimate runs to +-150 symbols, deterministically and
        with wide decision margins, and device and definition must agree exactly."""
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    fs, n = 144000.0, 131072
    sps = fs / 18000.0
    x, dib = make_signal(n, fs * (1 - 0.004), 4321, 0.1, 20.0, 25.0)
    cases = [(x, 384, 1e-3, 1e-3)]
    t = np.arange(n)
    for eps in (-0.3 / 256, 0.3 / 256):
        fb = 1 / sps + eps
        tone = (np.exp(2j * np.pi * (fb / 2) * t) + 0.8 * np.exp(-2j * np.pi * (fb / 2) * t + 0.4j)).astype(np.complex64)
        cases.append((tone, 640, 2e-4, 0.0))
    for x, reach, soft_tol, hard_tol in cases:
        ref_hard, _, info = tetra_np.demod(x.astype(np.complex128), fs)
        assert abs(info["tau"][-1]) * sps > reach + 100, info["tau"][-1]     # the instants do leave the ring
        bd = BatchDemodulator(fs, n, 2, "cf32", mode=MODE_TETRA)
        hards, softs, timing, margin = bd.process(np.concatenate([x, x]))
        bd.close()
        for r in range(2):
            assert len(softs[r]) == info["n_sym"], (len(softs[r]), info["n_sym"])
            scale = np.max(np.abs(info["sym"]))
            assert np.max(np.abs(softs[r] - info["sym"])) < soft_tol * scale
            # ((a): interpolating under a moving clock leaves a raw error rate of a few 1e-3, decisions next to a
            # boundary may differ between fp32 and fp64)
            assert np.mean(hards[r] != ref_hard) <= hard_tol
        np.testing.assert_array_equal(hards[0], hards[1])


def _wideband(n, fs, ks, M, seed0=300, snr_db=25.0):
    """Sum of pi/4-DQPSK carriers on the channeliser grid (channel index k -> k*fs/M, k >= M/2 negative)."""
    return synth.grid_carriers(n, fs, ks, M, seed0=seed0, snr_db=snr_db)


@pytest.mark.gpu
def test_gpu_channeliser_matches_definition():
    from oracle import pfb_np
    from tetraear_amd.channeliser import channelise
    for M, D, fs, n in ((96, 32, 2.4e6, 6000), (400, 125, 10e6, 9000), (72, 24, 1.8e6, 3000)):
        ks = [1, M // 3, M - 2]
        x, _ = _wideband(n, fs, ks, M)
        x32 = (x / 4).astype(np.complex64)
        y = channelise(x32, "cf32", M, D)
        probe = [0, 1, M // 3, M // 2, M - 2, M - 1]
        ref = pfb_np.channelise(x32.astype(np.complex128), M, D, channels=probe)
        scale = np.max(np.abs(ref))
        for i, k in enumerate(probe):
            assert np.max(np.abs(y[k] - ref[i])) < 2e-5 * scale, (M, k)
        # cu8 wire format goes through the same kernel
        u8 = synth.quantise_cu8(x, scale=0.25)
        y8 = channelise(u8, "cu8", M, D)
        ref8 = pfb_np.channelise(synth.cu8_to_c128(u8), M, D, channels=[ks[1]])
        assert np.max(np.abs(y8[ks[1]] - ref8[0])) < 2e-5 * np.max(np.abs(ref8))


@pytest.mark.gpu
def test_gpu_channeliser_batch_and_general_decimation():
    """grid.y batching gives the single-stream result per stream; a decimation that does not satisfy
    M | TB*D (register-FFT kernel's condition) takes the direct-DFT kernel and still matches."""
    from oracle import pfb_np
    from tetraear_amd.channeliser import channelise, channelise_batch
    M, D, fs, n = 400, 125, 10e6, 5000
    xs = [(_wideband(n, fs, [3, 77, 391], M, seed0=400 + 10 * i)[0] / 4).astype(np.complex64) for i in range(3)]
    yb = channelise_batch(np.concatenate(xs), "cf32", 3, M, D)
    yp = channelise_batch(np.concatenate(xs), "cf32", 3, M, D, pitch=48)
    for i in range(3):
        np.testing.assert_array_equal(yb[i], channelise(xs[i], "cf32", M, D))
        np.testing.assert_array_equal(yp[i], yb[i])
    for M, D, n in ((80, 27, 4000), (128, 40, 4100), (80, 25, 3000), (128, 36, 3000)):
        x, _ = _wideband(n, 2e6, [2, M // 2 + 3], M)
        x32 = (x / 4).astype(np.complex64)
        y = channelise(x32, "cf32", M, D)
        probe = [0, 2, M // 2 + 3, M - 1]
        ref = pfb_np.channelise(x32.astype(np.complex128), M, D, channels=probe)
        for i, k in enumerate(probe):
            assert np.max(np.abs(y[k] - ref[i])) < 2e-5 * np.max(np.abs(ref)), (M, D, k)


@pytest.mark.gpu
def test_gpu_channeliser_direct_kernel_agrees():
    """the general fallback kernel (direct small DFTs; taken when a window exceeds LDS) against the
    register-FFT kernel on the same input"""
    from tetraear_amd._lib import debug_option
    from tetraear_amd.channeliser import channelise
    for M, D, fs, n in ((400, 125, 10e6, 7000), (96, 32, 2.4e6, 5000)):
        x, _ = _wideband(n, fs, [1, M // 3, M - 2], M, seed0=700)
        x32 = (x / 4).astype(np.complex64)
        y_fft = channelise(x32, "cf32", M, D)
        with debug_option("pfb_direct", 1):
            y_dir = channelise(x32, "cf32", M, D)
        assert np.max(np.abs(y_fft - y_dir)) < 2e-5 * np.max(np.abs(y_dir))


@pytest.mark.gpu
def test_gpu_wideband_to_symbols():
    """2.4 MS/s wideband -> 96-channel filter bank (75 kS/s per channel) -> TETRA-mode demodulation;
    every occupied channel must give back the transmitted dibits."""
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    from tetraear_amd.channeliser import channelise
    M, D, fs, n = 96, 32, 2.4e6, 131072
    ks = [0, 3, 17, 47, 49, 80, 95]
    x, dibs = _wideband(n, fs, ks, M)
    y = channelise((x / 6).astype(np.complex64), "cf32", M, D)          # [96][4096] at 75 kS/s
    n_c = y.shape[1]
    bd = BatchDemodulator(fs / D, n_c, len(ks), "cf32", mode=MODE_TETRA)
    hards, softs, timing, margin = bd.process(np.ascontiguousarray(y[ks]))
    for i, k in enumerate(ks):
        ber, lag = best_ber(hards[i], dibs[k], edge=8)
        assert len(hards[i]) > 900 and ber == 0.0, (k, ber, lag)
    bd.close()


@pytest.mark.gpu
def test_gpu_pitched_rows_and_odd_chunk_lengths():
    """TETRA-mode plans read carriers at any row stride >= n (the channeliser's pitched output) and
    handle odd chunk lengths; results equal the dense even-length call on the same samples."""
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    fs, rows = 80000.0, 3
    for n, pitch in ((8389, 8400), (8389, 8389), (8192, 8200), (4097, 4101)):
        xs = [make_signal(n, fs, 900 + r, 0.1 * r, 30.0 * r, 20.0)[0] for r in range(rows)]
        dense = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA)
        h0, s0, t0, m0 = dense.process(np.concatenate(xs))
        padded = np.zeros((rows, pitch), dtype=np.complex64)
        for r in range(rows):
            padded[r, :n] = xs[r]
            padded[r, n:] = 7.0   # must never be read
        hard = np.zeros((rows, dense.info.max_soft), dtype=np.uint8)
        soft = np.zeros((rows, dense.info.max_soft), dtype=np.complex64)
        n_soft = np.zeros(rows, dtype=np.int32)
        from tetraear_amd._lib import check, ptr
        check(dense.lib.tdm_process(dense.handle, ptr(padded), pitch, None, None, ptr(hard), ptr(soft), ptr(n_soft), None, None))
        for r in range(rows):
            np.testing.assert_array_equal(hard[r, :n_soft[r] - 1], h0[r])
            np.testing.assert_array_equal(soft[r, :n_soft[r]], s0[r])
            assert best_ber(h0[r], make_signal(n, fs, 900 + r, 0.1 * r, 30.0 * r, 20.0)[1], edge=8)[0] == 0.0
        dense.close()


@pytest.mark.gpu
def test_gpu_wideband_receiver_device_chain():
    """WidebandReceiver: channeliser output handed to the TETRA-mode plan on the device (pitched rows, two
    streams per launch); every occupied channel gives back its transmitted dibits, and the result equals
    the host-chained path (channelise -> BatchDemodulator)."""
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    from tetraear_amd.channeliser import channelise
    from tetraear_amd.wideband import WidebandReceiver
    M, D, fs, n = 96, 32, 2.4e6, 65536
    ks = [[0, 5, 47, 90], [3, 49, 95]]
    xs, dibs = [], []
    for si, kk in enumerate(ks):
        x, d = _wideband(n, fs, kk, M, seed0=500 + 20 * si)
        xs.append((x / 6).astype(np.complex64))
        dibs.append(d)
    rx = WidebandReceiver(fs, n, M, D, streams=2, fmt="cf32")
    hard, n_sym, timing, margin = rx.process(np.concatenate(xs))
    for si, kk in enumerate(ks):
        y = channelise(xs[si], "cf32", M, D)
        bd = BatchDemodulator(fs / D, y.shape[1], len(kk), "cf32", mode=MODE_TETRA)
        hards, _, _, _ = bd.process(np.ascontiguousarray(y[kk]))
        bd.close()
        for i, k in enumerate(kk):
            got = hard[si, k, :n_sym[si, k]]
            np.testing.assert_array_equal(got, hards[i])
            ber, lag = best_ber(got, dibs[si][k], edge=8)
            assert n_sym[si, k] > 400 and ber == 0.0, (si, k, ber, lag)
    assert rx.channel_frequency(95) == -fs / M
    rx.close()


@pytest.mark.gpu
def test_gpu_wideband_receiver_with_the_gardner_loop():
    """the chain `north_star` spells out, on the device end to end: wideband stream -> polyphase channeliser -> RRC matched
    filter -> Gardner timing-error detector + loop -> Farrow -> differential decisions (WidebandReceiver(mode=
    MODE_TETRA_GARDNER)); every occupied channel gives back its transmitted dibits after the loop's acquisition, and its
    decisions equal the fp64 definition's loop run on the definition's channeliser output"""
    from oracle import pfb_np, tetra_np
    from tetraear_amd._lib import MODE_TETRA_GARDNER
    from tetraear_amd.wideband import WidebandReceiver
    M, D, fs, n = 96, 32, 2.4e6, 131072
    ks = [0, 5, 47, 49, 90]
    x, dibs = _wideband(n, fs, ks, M, seed0=640)
    xin = (x / 6).astype(np.complex64)
    ref = dict(zip(ks, pfb_np.channelise(xin.astype(np.complex128), M, D, channels=ks)))
    for ff in (False, True):      # gardner_ff_start: the loop of every chunk started at the feed-forward timing estimate
        rx = WidebandReceiver(fs, n, M, D, streams=1, fmt="cf32", mode=MODE_TETRA_GARDNER, gardner_ff_start=ff)
        hard, n_sym, timing, margin = rx.process(xin)
        rx.close()
        skip = 60 if ff else 700      # (a loop of this bandwidth may need a few hundred symbols from a half-symbol offset; started at the estimate it needs none)
        for k in ks:
            got = hard[0, k, :n_sym[0, k]]
            assert n_sym[0, k] > 900
            m = len(got)
            errs = min(int(np.sum(got[skip:m - 8] != dibs[k][lag + skip:lag + m - 8])) for lag in range(40) if len(dibs[k]) - lag >= m)
            assert errs == 0, (ff, k, errs)
            ref_hard, _, info = tetra_np.demod_gardner(ref[k], fs / D, ff_first=ff)
            mm = min(m, len(ref_hard))
            assert abs(m + 1 - len(info["t"])) <= 1 and np.mean(got[:mm] != ref_hard[:mm]) <= 2e-3, (ff, k)


@pytest.mark.gpu
def test_gpu_channeliser_random_configurations():
    """seeded random (M, D, length, wire format, pitch): channeliser vs the fp64 definition on probe channels
    (tools/sweep_pfb.py runs the same sweep open-ended: 76 115 cases without a mismatch in round 1)"""
    from oracle import pfb_np
    from tetraear_amd.channeliser import channelise_batch
    rng = np.random.default_rng(11)
    for it in range(250):
        M = int(rng.choice([72, 80, 96, 128, 400]))
        D = int(rng.integers(max(2, M // 8), M + 1)) if it % 2 else int(rng.choice([M // 4, M // 3, M // 2]))
        n = int(rng.integers(1, 9000))
        fmt = ["cf32", "cu8", "cs8"][it % 3]
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.2
        if fmt == "cf32":
            raw = x.astype(np.complex64)
            xd = raw.astype(np.complex128)
        elif fmt == "cu8":
            raw = synth.quantise_cu8(x, scale=1.0)
            xd = synth.cu8_to_c128(raw)
        else:
            q = np.clip(np.round(np.stack([x.real, x.imag], -1) * 128), -128, 127).astype(np.int8)
            raw = q.reshape(-1)
            xd = (q[:, 0].astype(np.float64) + 1j * q[:, 1].astype(np.float64)) / 128.0
        pitch = 0 if it % 4 < 2 else ((n + D - 1) // D + 15) // 16 * 16
        y = channelise_batch(raw, fmt, 1, M, D, pitch=pitch)[0]
        probe = sorted(set(int(k) for k in rng.integers(0, M, 4)) | {0, M - 1})
        ref = pfb_np.channelise(xd, M, D, channels=probe)
        sc = max(np.max(np.abs(ref)), 1e-30)
        for i, k in enumerate(probe):
            assert np.max(np.abs(y[k] - ref[i])) < 2e-5 * sc, (M, D, n, fmt, pitch, k)


@pytest.mark.gpu
def test_gpu_c5_full_size_wideband_chain():
    """BASELINE config 5 at its full size: one 10 MS/s cu8 stream of 1 048 576 samples -> 400-channel filter bank
    (D = 125, 80 kS/s per channel, 8389 samples each) -> TETRA-mode demodulation of every channel on the device.
    Nine occupied channels (band edges, both sides of DC, neighbours) give back their transmitted dibits without
    error, the device-chained result equals the host-chained one, the channeliser equals its fp64 definition on
    the probe channels over the whole length, and the channels away from every carrier stay at the noise floor."""
    from oracle import pfb_np
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    from tetraear_amd.channeliser import channelise
    from tetraear_amd.wideband import WidebandReceiver
    M, D, fs, n = 400, 125, 10e6, 1048576
    ks = [0, 1, 57, 133, 199, 201, 310, 398, 399]
    x, dibs = _wideband(n, fs, ks, M)
    u8 = synth.quantise_cu8(x, scale=1.0 / (4.0 * np.sqrt(len(ks) + 1.0)))
    rx = WidebandReceiver(fs, n, M, D, streams=1, fmt="cu8")
    hard, n_sym, timing, margin = rx.process(u8)
    assert hard.shape[:2] == (1, M) and rx.n_out == 8389
    y = channelise(u8, "cu8", M, D)                                   # [400][8389] cf32
    bd = BatchDemodulator(fs / D, y.shape[1], len(ks), "cf32", mode=MODE_TETRA)
    hards, _, _, _ = bd.process(np.ascontiguousarray(y[ks]))
    bd.close()
    for i, k in enumerate(ks):
        got = hard[0, k, :n_sym[0, k]]
        np.testing.assert_array_equal(got, hards[i])
        ber, lag = best_ber(got, dibs[k], edge=8)
        assert n_sym[0, k] > 1850 and ber == 0.0, (k, ber, lag)
    # channeliser against its definition at the full length (three probe channels: occupied, DC, empty)
    xd = synth.cu8_to_c128(u8)
    probe = [57, 0, 250]
    ref = pfb_np.channelise(xd, M, D, channels=probe)
    scale = np.max(np.abs(ref))
    for i, k in enumerate(probe):
        assert np.max(np.abs(y[k] - ref[i])) < 2e-5 * scale, k
    # channels at least three slots from every carrier hold only the noise floor (the bank is oversampled: 80 kS/s per
    # 25 kHz slot, so a carrier's energy also shows in its neighbours' transition bands)
    near = np.zeros(M, dtype=bool)
    for k in ks:
        near[[(k + d) % M for d in range(-2, 3)]] = True
    p_ch = np.mean(np.abs(y) ** 2, axis=1)
    assert np.max(p_ch[~near]) < 5e-2 * np.min(p_ch[ks])
    rx.close()


@pytest.mark.gpu
def test_gpu_rrc_matched_filter_alone_equals_definition():
    """tdm_plan_rrc_filter (the receiver's first stage as a stand-alone operator, the kernel the Gardner mode runs): against
    oracle/tetra_np.matched_filter with the plan's 16-bit taps, every tap count the kernel is instantiated for, chunk lengths
    that end inside a tile and inside a workgroup's run of tiles, rows at odd alignments"""
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    rng = np.random.default_rng(3)
    for fs, n in ((72000.0, 8192), (72000.0, 8193), (36000.0, 2047), (54000.0, 6001), (90000.0, 10240), (144000.0, 20000), (108000.0, 64)):
        rows = 3
        x = (rng.standard_normal((rows, n)) + 1j * rng.standard_normal((rows, n))).astype(np.complex64)
        bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA)
        y = bd.rrc_filter(x)
        bd.close()
        h = tetra_np.rrc_taps(fs / 18000.0)
        for r in range(rows):
            ref = tetra_np.matched_filter(x[r].astype(np.complex128), h)
            assert np.max(np.abs(y[r] - ref)) < 2e-6 * np.max(np.abs(ref)), (fs, n, r)


def test_occupancy_definition_picks_the_occupied_channels():
    """CPU: oracle/pfb_np.occupancy (the reference's gate rule per channel row, ui/modern.py:1921-2003, with the median of
    the channels' in-band power as noise floor) on the channeliser definition's output of a 96-channel stream: exactly the
    transmitted channels are flagged -- also where two of them are neighbours, whose energy the reference's own floor
    (the bins outside the centre channel) would have counted as noise."""
    from oracle import pfb_np
    M, D, fs = 96, 32, 2.4e6
    ks = [0, 5, 6, 47, 90, 95]
    n = D * (pfb_np.OCC_FFT + 8)
    x, _ = _wideband(n, fs, ks, M, seed0=610)
    y = pfb_np.channelise(x / 6, M, D)
    sig, peak, occ = pfb_np.occupancy(y, M, fs / D)
    assert sorted(np.where(occ)[0]) == ks
    assert np.min(sig[ks]) - np.median(sig) > 18 and np.max(np.delete(sig, ks)) - np.median(sig) < 8
    assert pfb_np.occupancy_bins(fs / D) == (128 - 42, 128 + 42)


@pytest.mark.gpu
def test_gpu_occupancy_gate_matches_definition_and_gated_receiver_equals_ungated_rows():
    """tdm_occupancy_gate on the channeliser's rows: statistics within 0.01 dB of the definition and the same flags; then
    WidebandReceiver(gated=True) -- channeliser -> gate -> receiver over the listed rows, all on the device -- gives, on every
    occupied row, bit for bit what the ungated chain gives there (hard decisions, soft symbols, timing, margin), reports no
    symbols on the others, and the occupied rows are the transmitted channels."""
    from oracle import pfb_np
    from tetraear_amd.channeliser import channelise
    from tetraear_amd.gate import occupancy_gate
    from tetraear_amd.wideband import WidebandReceiver
    for M, D, fs, n, ks in ((96, 32, 2.4e6, 65536, [[0, 5, 6, 47, 90, 95], [3, 49]]),
                            (400, 125, 10e6, 125 * 1500, [[0, 1, 57, 133, 199, 201, 310, 398, 399], [7, 200, 201, 202]])):
        xs = []
        for si, kk in enumerate(ks):
            x, _ = _wideband(n, fs, kk, M, seed0=640 + 20 * si)
            xs.append((x / (2.0 * np.sqrt(len(kk) + 1.0))).astype(np.complex64))
        y = np.concatenate([channelise(x, "cf32", M, D) for x in xs])          # [2 M][n_out]
        sig, peak, occ, rows = occupancy_gate(y, M, fs / D)
        rsig, rpeak, rocc = pfb_np.occupancy(y, M, fs / D)
        assert np.max(np.abs(sig - rsig)) < 0.01 and np.max(np.abs(peak - rpeak)) < 0.01
        np.testing.assert_array_equal(occ, rocc)
        want = sorted(si * M + k for si, kk in enumerate(ks) for k in kk)
        assert list(rows) == want and sorted(np.where(occ)[0]) == want
        # the chain on the device, gated against ungated
        both = []
        for gated in (False, True):
            rx = WidebandReceiver(fs, n, M, D, streams=2, fmt="cf32", gated=gated)
            rx.d_in.upload(np.concatenate(xs))
            rx.enqueue()
            rx.sync()
            both.append(rx.demod.download())
            if gated:
                gs, gp, go = rx.occupancy()
                np.testing.assert_array_equal(go.reshape(-1), occ)
            rx.close()
        (h0, s0, n0, t0, m0), (h1, s1, n1, t1, m1) = both
        for r in range(2 * M):
            if r in want:
                assert n1[r] == n0[r] and n0[r] > 100
                np.testing.assert_array_equal(h1[r, :n0[r] - 1], h0[r, :n0[r] - 1])
                np.testing.assert_array_equal(s1[r, :n0[r]], s0[r, :n0[r]])
                assert t1[r] == t0[r] and m1[r] == m0[r]
            else:
                assert n1[r] == 0


def _quantise8(x, fmt):
    """a complex baseband signal at a quarter of full scale as wire bytes, and the samples those bytes MEAN (what the
    definition is evaluated on): cu8 u / 127.5 - 1 (pyrtlsdr, the channeliser), cs8 s / 128"""
    x = x / (4.0 * np.max(np.abs(x)))
    if fmt == "cu8":
        raw = synth.quantise_cu8(x, scale=1.0)
        return raw, synth.cu8_to_c128(raw)
    raw = np.empty(2 * len(x), dtype=np.int8)
    raw[0::2] = np.clip(np.rint(128 * x.real), -128, 127)
    raw[1::2] = np.clip(np.rint(128 * x.imag), -128, 127)
    return raw, (raw[0::2].astype(np.float64) + 1j * raw[1::2].astype(np.float64)) / 128.0


@pytest.mark.gpu
def test_gpu_tetra_int8_input_matches_definition_on_the_dequantised_samples():
    """north_star: "coalesced complex-int8/float loads".  TDM_MODE_TETRA plans on cu8 / cs8 input (round 6: the bytes are
    converted where the kernels stage their window; in the fused receiver an 8-bit sample is ONE exact bf16 plane, two
    matrix-core products per step instead of four): hard decisions, symbol counts, timing and margin equal to the fp64
    definition evaluated on the samples the bytes mean, soft symbols within 1e-5 of it, no error against what was sent,
    and the same decisions as a cf32 plan fed with those samples; rows at odd byte offsets (a pitch that is no multiple of
    four bytes); the RRC stage alone (tdm_plan_rrc_filter) on 8-bit input against the definition's matched filter; the
    Gardner mode on 8-bit input (three launches: its matched filter converts)."""
    from tetraear_amd._lib import MODE_TETRA, MODE_TETRA_GARDNER, check, ptr
    from tetraear_amd.batch import BatchDemodulator
    for fmt in ("cu8", "cs8"):
        for fs, n, seed, toff, coff, snr in CASES[:5]:
            rows = 3
            sig = [make_signal(n, fs, seed * 10 + r, toff + 0.05 * r, coff, snr) for r in range(rows)]
            raws, xqs = zip(*[_quantise8(s[0].astype(np.complex128), fmt) for s in sig])
            bd = BatchDemodulator(fs, n, rows, fmt, mode=MODE_TETRA)
            hards, softs, timing, margin = bd.process(np.concatenate(raws))
            y8 = bd.rrc_filter(np.concatenate(raws))
            bd.close()
            bf = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA)
            hf, sf, tf, mf = bf.process(np.concatenate([q.astype(np.complex64) for q in xqs]))
            bf.close()
            for r in range(rows):
                ref_hard, _, info = tetra_np.demod(xqs[r], fs)
                assert len(softs[r]) == info["n_sym"], (fmt, fs, r)
                np.testing.assert_array_equal(hards[r], ref_hard)
                np.testing.assert_array_equal(hards[r], hf[r])
                scale = np.max(np.abs(info["sym"]))
                assert np.max(np.abs(softs[r] - info["sym"])) < 1e-5 * scale, (fmt, fs, r)
                assert best_ber(hards[r], sig[r][1])[0] <= (0.0 if snr >= 20.0 else 3e-3)
                assert abs(timing[r] / 1000.0 - info["tau"][len(info["tau"]) // 2]) < 2e-3
                assert abs(margin[r] - info["margin"]) < 1e-3
                ref_y = tetra_np.matched_filter(xqs[r], tetra_np.rrc_taps(fs / 18000.0))
                assert np.max(np.abs(y8[r] - ref_y)) < 2e-6 * np.max(np.abs(ref_y)), (fmt, fs, r)
    # rows with a pitch of an odd number of samples (2-byte aligned rows only), through the C-ABI
    fs, n, rows, pitch = 72000.0, 8191, 3, 8193
    sig = [make_signal(n, fs, 900 + r, 0.1 * r, 25.0, 22.0) for r in range(rows)]
    raws, xqs = zip(*[_quantise8(s[0].astype(np.complex128), "cu8") for s in sig])
    buf = np.full((rows, 2 * pitch), 77, dtype=np.uint8)
    for r in range(rows):
        buf[r, :2 * n] = raws[r]
    bd = BatchDemodulator(fs, n, rows, "cu8", mode=MODE_TETRA)
    ms = bd.info.max_soft
    hard = np.zeros((rows, ms), np.uint8); soft = np.zeros((rows, ms), np.complex64)
    ns = np.zeros(rows, np.int32); tm = np.zeros(rows, np.int32); mm = np.zeros(rows)
    check(bd.lib.tdm_process(bd.handle, ptr(buf), pitch, None, None, ptr(hard), ptr(soft), ptr(ns), ptr(tm), ptr(mm)))
    bd.close()
    for r in range(rows):
        ref_hard, _, info = tetra_np.demod(xqs[r], fs)
        assert ns[r] == info["n_sym"]
        np.testing.assert_array_equal(hard[r, :ns[r] - 1], ref_hard)
    # the Gardner mode on bytes: 33 taps -> the fused kernel (its producers convert), chunks in pieces as for cf32; 65 taps -> the
    # three launches (the stand-alone matched filter converts), whole chunks; decisions as the definition's loop evaluated the
    # same way
    for fs, n, want_pieces in ((72000.0, 12000, 4), (144000.0, 12000, 1)):
        rows = 4
        sig = [make_signal(n, fs, 950 + r, 0.07 * r - 0.1, 30.0, 22.0) for r in range(rows)]
        raws, xqs = zip(*[_quantise8(s[0].astype(np.complex128), "cu8") for s in sig])
        bd = BatchDemodulator(fs, n, rows, "cu8", mode=MODE_TETRA_GARDNER)
        K = int(bd.info.gardner_segments)
        assert K == want_pieces, (fs, K)
        hards, softs, timing, margin = bd.process(np.concatenate(raws))
        bd.close()
        bf = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA_GARDNER)
        assert int(bf.info.gardner_segments) == K
        hf, sf, tf, mf = bf.process(np.concatenate([q.astype(np.complex64) for q in xqs]))
        bf.close()
        for r in range(rows):
            ref_hard, _, info = tetra_np.demod_gardner(xqs[r], fs, segments=K)
            assert abs(len(softs[r]) - len(info["t"])) <= 1
            m = min(len(hards[r]), len(ref_hard))
            assert np.mean(hards[r][:m] != ref_hard[:m]) <= 1e-3, (fs, r)
            # the same samples as cf32 through the same kernels: the same decisions, soft symbols within fp32 rounding
            assert len(hf[r]) == len(hards[r]) and np.array_equal(hf[r], hards[r]), (fs, r)
            assert np.max(np.abs(sf[r] - softs[r])) <= 2e-5 * np.max(np.abs(sf[r])), (fs, r)
