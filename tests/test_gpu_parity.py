"""GPU: the HIP path (through the C-ABI / the drop-in SignalProcessor) against the golden vectors
of the imported reference and against the CPU oracle on seeded inputs.

Bars: hard symbols bit-exact; soft symbols within SOFT_TOL of max|soft| (north-star: 1e-5)."""
import ctypes as C

import numpy as np
import pytest

from tests.golden_cases import CASES, TIMING_DEGENERATE, case_c128, case_cu8

pytestmark = pytest.mark.gpu
SOFT_TOL = 1e-10


@pytest.fixture(scope="module")
def SP():
    from tetraear_amd.signal import SignalProcessor
    return SignalProcessor


def _check(name, hard, soft, gold_process):
    g_hard = gold_process[name + "__hard"]
    g_soft = gold_process[name + "__soft"]
    assert hard.dtype == np.uint8
    if name in TIMING_DEGENERATE:
        m = min(len(hard), len(g_hard))
        np.testing.assert_array_equal(hard[:m], g_hard[:m])
        return
    assert len(soft) == len(g_soft)
    np.testing.assert_array_equal(hard, g_hard)
    if len(g_soft):
        scale = np.max(np.abs(g_soft)) or 1.0
        err = np.max(np.abs(soft - g_soft)) / scale
        assert err <= SOFT_TOL, f"{name}: soft error {err:.3e}"


@pytest.mark.parametrize("name", sorted(CASES))
def test_process_c128_matches_reference(name, gold_process, SP):
    """Drop-in call exactly as the reference's callers make it: complex128 in."""
    c = CASES[name]
    x = case_c128(c)
    p = SP(c["fs"])
    if "pre_shift" in c:
        x = p.frequency_shift(x, c["pre_shift"])
    hard = p.process(x, c["foff"])
    _check(name, hard, p.symbols, gold_process)
    p.close()


@pytest.fixture(params=["auto", "raw"])
def cu8_engine(request, monkeypatch):
    """A few carriers run the decimator that holds its samples as doubles (shorter blocks fill the chip sooner), big
    batches the raw-integer one; "raw" puts single carriers on the raw-integer kernel too (tdm_debug_set raw_min_blocks 0)."""
    from tetraear_amd._lib import debug_option
    from tetraear_amd.signal import processor as P
    P.close_plans()
    if request.param == "raw":
        with debug_option("raw_min_blocks", 0):
            yield request.param
    else:
        yield request.param
    P.close_plans()


@pytest.mark.parametrize("name", sorted(n for n, c in CASES.items() if c["kind"] in ("noise", "dqpsk", "const")))
def test_process_cu8_matches_reference(name, gold_process, SP, cu8_engine):
    """Raw RTL-SDR bytes in (in-kernel u8 -> float conversion must equal pyrtlsdr's), both cu8 decimator kernels."""
    c = CASES[name]
    p = SP(c["fs"])
    hard = p.process_cu8(case_cu8(c), c["foff"])
    _check(name, hard, p.symbols, gold_process)
    # the same call again on the same plan and buffers
    for _ in range(2):
        again = p.process_cu8(case_cu8(c), c["foff"])
        np.testing.assert_array_equal(again, hard)
    p.close()


def test_channelised_shared_input(gold_process):
    """8 carriers in one wideband cu8 stream, per-carrier input-rate shift fused
    into the decimator load; oracle per carrier = p.process(p.frequency_shift(x, f_k))."""
    from tetraear_amd.batch import BatchDemodulator
    c0 = CASES["mc8_k0"]
    u8 = case_cu8(c0)
    offs = [CASES[f"mc8_k{k}"]["pre_shift"] for k in range(8)]
    bd = BatchDemodulator(2.4e6, c0["n"], 8, "cu8")
    hards, softs, bp, mm = bd.process(u8, freq_offsets=[0.0] * 8, pre_shifts=offs, shared_input=True)
    for k in range(8):
        _check(f"mc8_k{k}", hards[k], softs[k], gold_process)
    bd.close()


def test_stage_methods(gold_stages, SP):
    from tetraear_amd import synth
    g = gold_stages
    x = synth.cu8_to_c128(synth.noise_cu8(4000, int(g["x4000_seed"][0])))
    p = SP(2.4e6)

    def close(a, b, tol=1e-10):
        assert a.shape == b.shape
        if len(b):
            assert np.max(np.abs(a - b)) <= tol * max(1.0, np.max(np.abs(b)))

    # At fs = 2.4 MHz the 25 kHz Butterworth is very narrow (Wn = 0.0104) and scipy's own
    # transfer-function-form result is ~4e-11 away from exact arithmetic (measured against a
    # long-double run); the device runs the same filter as two biquads, so the two agree to the
    # reference's own rounding noise, not to 1e-12.
    close(p.filter_signal(x), g["filter_default"], 1e-9)
    close(p.filter_signal(x, bandwidth=50000), g["filter_bw50k"], 1e-9)
    close(p.filter_signal(x, 25000, 240000.0), g["filter_240k"])
    close(p.filter_signal(x, 25000, 20000.0), g["filter_clamp_hi"], 1e-8)
    close(p.filter_signal(x, 100.0, 2.4e6), g["filter_clamp_lo"], 1e-9)
    close(p.filter_signal(x[:15]), g["filter_short15"])
    close(p.filter_signal(x[:16], 25000, 240000.0), g["filter_short16"])
    close(p.frequency_shift(x, 1000), g["shift_1000"])
    close(p.frequency_shift(x, -3515.625, 240000.0), g["shift_m3515_240k"])
    close(p.frequency_shift(x, 0), g["shift_0"])
    np.testing.assert_array_equal(p.extract_symbols(x), g["extract_default"])
    np.testing.assert_array_equal(p.extract_symbols(x, 240000.0), g["extract_240k"])
    np.testing.assert_array_equal(p.extract_symbols(x, 300000.0), g["extract_300k"])
    np.testing.assert_array_equal(p.extract_symbols(x[:50], 18000.0), g["extract_18k"])
    assert len(p.extract_symbols(x[:5], 240000.0)) == 0
    np.testing.assert_array_equal(p.demodulate_dqpsk(x), g["demod_x"])
    np.testing.assert_array_equal(p.demodulate_dqpsk(x[:2]), g["demod_2"])
    np.testing.assert_array_equal(p.demodulate_dqpsk(np.zeros(10, dtype=complex)), g["demod_zeros"])
    close(p.resample(x[:1000], 1.2e6), g["resample_1200k"], 1e-12)
    close(p.resample(x[:301], 3.0e6), g["resample_up"], 1e-12)
    # 1-ulp threshold probes: decisions depend on the last bit of libm's atan2; require agreement
    # on every probe that is not within 4 ulp of a threshold
    seq = g["demod_probe_in"]
    probe = p.demodulate_dqpsk(seq)
    ref = g["demod_probe"]
    assert len(probe) == len(ref)
    ph = np.angle(seq[1:] * np.conj(seq[:-1]))
    thr = np.array([-5 * np.pi / 8, -3 * np.pi / 8, 3 * np.pi / 8, 5 * np.pi / 8])
    dist_ulp = np.min(np.abs(ph[:, None] - thr[None, :]) / np.spacing(np.abs(thr))[None, :], axis=1)
    differ = np.nonzero(probe != ref)[0]
    print("threshold probes that differ from the reference (index, phase, ulps from the threshold):",
          [(int(i), float(ph[i]), float(dist_ulp[i])) for i in differ])
    assert np.all(dist_ulp[differ] <= 4), "a decision more than 4 ulp from every threshold differs"
    assert np.array_equal(probe[dist_ulp > 4], ref[dist_ulp > 4])


def test_reference_style_contracts(SP):
    """The reference's own structural assertions (tests/unit/test_signal_processor.py:14-116,
    tests/integration/test_end_to_end.py:17-31) re-run against the GPU class."""
    rng = np.random.default_rng(0)
    t = np.arange(0, 0.01, 1 / 2.4e6)
    iq = np.exp(1j * 0 * t) + (rng.standard_normal(len(t)) + 1j * rng.standard_normal(len(t))) * 0.1
    p = SP()
    assert p.sample_rate == 2.4e6 and p.symbol_rate == 18000 and p.samples_per_symbol > 0
    assert SP(sample_rate=1.0e6).sample_rate == 1.0e6
    assert len(p.filter_signal(np.array([]))) == 0
    r = p.filter_signal(iq, bandwidth=25000)
    assert len(r) == len(iq) and isinstance(r, np.ndarray)
    assert len(p.filter_signal(iq, bandwidth=50000)) == len(iq)
    r = p.frequency_shift(iq, 1000)
    assert len(r) == len(iq) and np.iscomplexobj(r)
    assert len(p.frequency_shift(iq, 0)) == len(iq)
    assert len(p.demodulate_dqpsk(np.array([]))) == 0
    assert len(p.demodulate_dqpsk(np.array([1 + 1j]))) == 0
    r = p.demodulate_dqpsk(iq)
    assert r.dtype == np.uint8 and np.all(r <= 3) and len(r) > 0
    assert len(p.extract_symbols(np.array([]))) == 0
    r = p.extract_symbols(iq)
    assert len(r) > 0 and np.iscomplexobj(r)
    out = p.process(np.array([]))
    assert out.dtype == np.uint8 and len(out) == 0 and len(p.symbols) == 0
    # end-to-end chain of test_end_to_end.py: filter -> demod -> extract
    f = p.filter_signal(iq)
    d = p.demodulate_dqpsk(f)
    s = p.extract_symbols(f)
    assert len(d) > 0 and len(s) > 0
    r = p.resample(iq, 1.2e6)   # tests/unit/test_signal_processor.py:27-34
    assert len(r) == len(iq) // 2 and isinstance(r, np.ndarray) and np.iscomplexobj(r)


def test_batch_vs_oracle_many_rows():
    """64 independent carriers in one launch against the CPU oracle, ragged length."""
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    from tetraear_amd.batch import BatchDemodulator
    rows, n = 64, 40000 + 13
    u8 = np.concatenate([synth.noise_cu8(n, 9000 + r) for r in range(rows)])
    foffs = [(-1) ** r * 97.65625 * r for r in range(rows)]
    bd = BatchDemodulator(2.4e6, n, rows, "cu8")
    hards, softs, bp, mm = bd.process(u8, freq_offsets=foffs)
    worst = 0.0
    for r in range(rows):
        o = OracleSignalProcessor(2.4e6)
        ref = o.process(synth.cu8_to_c128(u8[2 * n * r: 2 * n * (r + 1)]), foffs[r])
        assert bp[r] == o.best_phase
        np.testing.assert_array_equal(hards[r], ref)
        worst = max(worst, np.max(np.abs(softs[r] - o.symbols)) / np.max(np.abs(o.symbols)))
        assert abs(mm[r] - o.min_margin) < 1e-9
    assert worst <= SOFT_TOL
    bd.close()


def test_full_size_properties():
    """BASELINE full size (262144-sample chunks): size-independent properties of the path --
    a pure phase rotation and a positive gain on the input leave every hard decision unchanged
    (differential detector, normalised slicer), and carriers in a batch do not interact."""
    from tetraear_amd import synth
    from tetraear_amd.batch import BatchDemodulator
    n, rows = 262144, 6
    base = synth.cu8_to_c128(synth.noise_cu8(n, 777)) * 0.5
    rot = base * np.exp(1j * 0.7)
    gain = base * 1.75
    other = synth.cu8_to_c128(synth.noise_cu8(n, 778))
    x = np.concatenate([base, rot, gain, other, base, other]).astype(np.complex128)
    bd = BatchDemodulator(2.4e6, n, rows, "cf64")
    hards, softs, bp, mm = bd.process(x, freq_offsets=[1171.875] * rows)
    assert len(hards[0]) >= 2013
    np.testing.assert_array_equal(hards[0], hards[1])
    np.testing.assert_array_equal(hards[0], hards[2])
    np.testing.assert_array_equal(hards[0], hards[4])
    np.testing.assert_array_equal(hards[3], hards[5])
    np.testing.assert_array_equal(softs[0], softs[4])            # deterministic, no cross-talk
    assert np.max(np.abs(softs[1] - softs[0] * np.exp(1j * 0.7))) < 1e-11 * np.max(np.abs(softs[0]))
    assert np.max(np.abs(softs[2] - softs[0] * 1.75)) < 1e-11 * np.max(np.abs(softs[0]))
    bd.close()


def test_abi_direct():
    """Straight through the C-ABI with ctypes (what INTEGRATION.md shows), cu8 in."""
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import _lib, synth
    L = _lib.load()
    assert L.tdm_device_count() >= 1
    n = 30000
    u8 = synth.noise_cu8(n, 31337)
    h = C.c_void_p()
    _lib.check(L.tdm_plan_create(2.4e6, n, 1, _lib.FMT_CU8, 0, 0, C.byref(h)))
    info = _lib.PlanInfo()
    _lib.check(L.tdm_plan_get_info(h, C.byref(info)))
    assert info.q == 10 and info.sps == 13 and info.n_dec == 3000
    hard = np.zeros(info.max_soft, np.uint8)
    soft = np.zeros(info.max_soft, np.complex128)
    ns = np.zeros(1, np.int32)
    fo = np.array([-250.0])
    _lib.check(L.tdm_process(h, _lib.ptr(u8), n, None, _lib.ptr(fo), _lib.ptr(hard), _lib.ptr(soft), _lib.ptr(ns),
                             None, None))
    o = OracleSignalProcessor(2.4e6)
    ref = o.process(synth.cu8_to_c128(u8), -250.0)
    np.testing.assert_array_equal(hard[:ns[0] - 1], ref)
    _lib.check(L.tdm_plan_destroy(h))
    # error convention: negative status + text, no exception across the ABI
    assert L.tdm_plan_create(-1.0, n, 1, 0, 0, 0, C.byref(h)) == -1
    assert "bad" in _lib.last_error()


def test_config3_64_carriers_shared_stream_vs_oracle():
    """BASELINE config 3 at full size: 64 carriers on a 25 kHz grid in ONE 2.4 MS/s cu8 stream of
    262144 samples, demodulated in one batch (per-carrier input-rate shift fused into the decimator
    load); oracle per carrier = p.process(p.frequency_shift(x, f_k)) (processor.py:85-100, :221-273), EVERY carrier."""
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    from tetraear_amd.batch import BatchDemodulator
    n = 262144
    offs = [(k - 31.5) * 25000.0 for k in range(64)]
    u8, _ = synth.multicarrier_cu8(n, 2.4e6, offs, seed0=100)
    bd = BatchDemodulator(2.4e6, n, 64, "cu8")
    hards, softs, bp, mm = bd.process(u8, pre_shifts=offs, shared_input=True)
    x = synth.cu8_to_c128(u8)
    worst = 0.0
    for k in range(64):
        o = OracleSignalProcessor(2.4e6)
        ref = o.process(o.frequency_shift(x, offs[k]))
        assert bp[k] == o.best_phase, k
        np.testing.assert_array_equal(hards[k], ref, err_msg=f"carrier {k}")
        worst = max(worst, float(np.max(np.abs(softs[k] - o.symbols)) / np.max(np.abs(o.symbols))))
    assert worst <= SOFT_TOL, worst
    assert all(len(h) >= 2013 for h in hards)
    bd.close()


def test_bench_shared_workload_every_carrier_vs_oracle_and_digest():
    """`bench.py --shared --carriers 64` (its own input and offsets): every carrier against the oracle, and the digest the
    bench asserts (tests/golden/bench_digest.json) equal to the one this run gives."""
    import bench
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    from tetraear_amd.batch import BatchDemodulator
    n, carriers = 262144, 64
    iq, _ = bench.make_batch(1, n, "cu8", 0)
    pre = bench.shared_offsets(carriers)
    bd = BatchDemodulator(2.4e6, n, carriers, "cu8")
    bd.alloc_device_io(shared_input=True)
    bd.upload(iq, freq_offsets=None, pre_shifts=pre)
    bd.enqueue()
    bd.sync()
    hard, soft, n_soft, bp, mm = bd.download()
    bd.close()
    x = synth.cu8_to_c128(iq)
    for k in range(carriers):
        o = OracleSignalProcessor(2.4e6)
        ref = o.process(o.frequency_shift(x, pre[k]))
        ns = int(n_soft[k])
        assert ns == len(o.symbols) and bp[k] == o.best_phase, k
        np.testing.assert_array_equal(hard[k, :ns - 1], ref, err_msg=f"carrier {k}")
        assert np.max(np.abs(soft[k, :ns] - o.symbols)) <= SOFT_TOL * np.max(np.abs(o.symbols))
    want = bench.expected_digest(bench.digest_key(carriers, n, "cu8", 2.4e6, 0, True))
    assert want is not None and bench.output_digest(hard, n_soft, bp) == want


def test_config3_in_time_batches_T4():
    """Config 3 as its callers run it -- chunk after chunk (ui/modern.py:1908-1912) -- four turns per call: plan option
    "rows_per_chunk" = 64 on a plan of 4 x 64 rows over FOUR CONSECUTIVE 262144-sample chunks of one multicarrier stream;
    every one of the 256 rows against the oracle's p.process(p.frequency_shift(x_chunk, f_k)), through the host-pointer
    call and through the device-resident one with bench.py --shared --chunks 4's own workload and its pinned digest."""
    import bench
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    from tetraear_amd.batch import BatchDemodulator
    n, C_, T = 262144, 64, 4
    offs = [(k - 31.5) * 25000.0 for k in range(C_)]
    u8, _ = synth.multicarrier_cu8(n * T, 2.4e6, offs, seed0=100)
    bd = BatchDemodulator(2.4e6, n, C_ * T, "cu8").set_rows_per_chunk(C_)
    hards, softs, bp, mm = bd.process(u8, pre_shifts=np.tile(offs, T))
    bd.close()
    x = synth.cu8_to_c128(u8)
    for r in range(0, C_ * T, 3):
        o = OracleSignalProcessor(2.4e6)
        ref = o.process(o.frequency_shift(x[(r // C_) * n:(r // C_ + 1) * n], offs[r % C_]))
        assert bp[r] == o.best_phase, r
        np.testing.assert_array_equal(hards[r], ref, err_msg=f"row {r}")
        assert np.max(np.abs(softs[r] - o.symbols)) <= SOFT_TOL * np.max(np.abs(o.symbols))
    # the bench's own workload, every row, and its digest
    iq, _ = bench.make_shared_stream(n, T, "cu8", 0)
    pre = np.tile(bench.shared_offsets(C_), T)
    bd = BatchDemodulator(2.4e6, n, C_ * T, "cu8").set_rows_per_chunk(C_)
    bd.alloc_device_io()
    bd.upload(iq, freq_offsets=None, pre_shifts=pre)
    bd.enqueue()
    hard, soft, n_soft, bp, mm = bd.download()
    bd.close()
    x = synth.cu8_to_c128(iq)
    for r in range(C_ * T):
        o = OracleSignalProcessor(2.4e6)
        ref = o.process(o.frequency_shift(x[(r // C_) * n:(r // C_ + 1) * n], pre[r]))
        ns = int(n_soft[r])
        assert ns == len(o.symbols) and bp[r] == o.best_phase, r
        np.testing.assert_array_equal(hard[r, :ns - 1], ref, err_msg=f"row {r}")
        assert np.max(np.abs(soft[r, :ns] - o.symbols)) <= SOFT_TOL * np.max(np.abs(o.symbols))
    want = bench.expected_digest(bench.digest_key(C_, n, "cu8", 2.4e6, 0, True) + ":chunks4")
    assert want is not None and bench.output_digest(hard, n_soft, bp) == want
    # a count that does not divide the rows is refused
    from tetraear_amd._lib import TetraHipError
    bd = BatchDemodulator(2.4e6, 4096, 6, "cu8")
    with pytest.raises(TetraHipError):
        bd.set_rows_per_chunk(4)
    bd.close()


def test_config5_10Msps_q41_all_400_grid_carriers():
    """BASELINE config 5 in reference mode at FULL size: one 10 MS/s cu8 stream of 1 048 576 samples (q = 41, filter memory
    8246 samples), ALL 400 carriers of the 25 kHz grid in ONE launch (grid.y = 400, per-carrier input-rate shift).
    Ten carriers against the oracle (p.process(p.frequency_shift(x, f_k), foff), processor.py:85-100, :221-273: band
    edges, both sides of DC, and a spread); every one of the 400 rows bit-identical to the same carrier demodulated in
    a small batch of 8 (the launch geometry of the full batch changes nothing), and duplicated offsets give duplicated
    rows (no cross-talk between rows)."""
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    from tetraear_amd.batch import BatchDemodulator
    n, M = 1048576, 400
    u8 = synth.noise_cu8(n, 4100)
    offs = np.array([(k - 199.5) * 25000.0 for k in range(M)])
    foffs = np.array([((k * 7) % 11 - 5) * 234.375 for k in range(M)])
    bd = BatchDemodulator(10e6, n, M, "cu8")
    bd.alloc_device_io(shared_input=True)
    bd.upload(u8, freq_offsets=foffs, pre_shifts=offs)
    bd.enqueue()
    bd.sync()
    hard, soft, n_soft, bp, mm = bd.download()
    bd.close()
    assert np.all(n_soft >= 1966) and np.all(n_soft <= 1967)
    x = synth.cu8_to_c128(u8)
    for k in (0, 1, 57, 150, 199, 200, 201, 310, 398, 399):
        o = OracleSignalProcessor(10e6)
        ref = o.process(o.frequency_shift(x, offs[k]), foffs[k])
        ns = int(n_soft[k])
        assert bp[k] == o.best_phase and ns == len(o.symbols), k
        np.testing.assert_array_equal(hard[k, :ns - 1], ref, err_msg=f"carrier {k}")
        assert np.max(np.abs(soft[k, :ns] - o.symbols)) <= SOFT_TOL * np.max(np.abs(o.symbols)), k
    small = BatchDemodulator(10e6, n, 8, "cu8")
    for k0 in range(0, M, 8):
        hs, ss, bps, _ = small.process(u8, pre_shifts=offs[k0:k0 + 8], freq_offsets=foffs[k0:k0 + 8], shared_input=True)
        for j in range(8):
            k = k0 + j
            ns = int(n_soft[k])
            assert bps[j] == bp[k] and len(ss[j]) == ns, k
            np.testing.assert_array_equal(hs[j], hard[k, :ns - 1], err_msg=f"carrier {k}")
            np.testing.assert_array_equal(ss[j], soft[k, :ns], err_msg=f"carrier {k}")
    # the same offsets in another row order: rows follow their offsets, nothing leaks between neighbours
    perm = np.array([399, 0, 399, 200, 1, 200, 57, 57])
    hs, ss, bps, _ = small.process(u8, pre_shifts=offs[perm], freq_offsets=foffs[perm], shared_input=True)
    for j, k in enumerate(perm):
        np.testing.assert_array_equal(hs[j], hard[k, :int(n_soft[k]) - 1])
    small.close()


def test_recorded_file_ingest_pipelined(tmp_path):
    """BASELINE config 1 shape: a cu8 'recording' cut into 262144-sample reads; every read must equal
    the oracle's process() of that read (chunks are stateless), through the copy/compute pipeline."""
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    from tetraear_amd.ingest import demodulate_recording
    chunk, n_chunks, tail = 65536, 11, 30011
    u8 = synth.noise_cu8(chunk * n_chunks + tail, 606)
    path = tmp_path / "capture.cu8"
    u8.tofile(path)
    outs = demodulate_recording(str(path), 2.4e6, chunk=chunk, freq_offset=1171.875, rows_per_batch=4)
    assert len(outs) == n_chunks + 1
    for i in (0, 3, 4, 7, 8, 10, 11):
        lo = 2 * chunk * i
        hi = lo + 2 * (chunk if i < n_chunks else tail)
        ref = OracleSignalProcessor(2.4e6).process(synth.cu8_to_c128(u8[lo:hi]), 1171.875)
        np.testing.assert_array_equal(outs[i], ref)


@pytest.mark.gpu
def test_random_lengths_and_rates_vs_oracle(cu8_engine):
    """seeded random chunk lengths (1 .. 300 000, plus lengths around block multiples and the fall-back
    thresholds), sample rates incl. 10 MS/s (q = 41) and AFC offsets: process_cu8 against the C oracle"""
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    from tetraear_amd.signal.processor import SignalProcessor
    rng = np.random.default_rng(77)
    rates = [2.4e6, 2.4e6, 1.8e6, 2.048e6, 960000.0, 480000.0, 240000.0, 72000.0, 3.2e6, 10e6]
    # the GUI's sample-rate control is continuous (ui/modern.py:3915-3917): seeded arbitrary rates, 0.5 .. 10.5 MS/s
    # (every decimation factor 2 .. 43, i.e. both decimator engines), beside the RTL-SDR ones
    rates += [float(np.round(r, 1)) for r in np.random.default_rng(78).uniform(0.5e6, 10.5e6, 14)]
    specials = [27, 28, 29, 150, 161, 5119, 5120, 5121, 20480, 20481, 6144, 131071, 131072, 262145]
    procs = {}
    for it in range(160):
        fs = rates[rng.integers(len(rates))]
        n = int(specials[it % len(specials)]) if it % 3 == 0 else int(rng.integers(1, 300000 if it % 10 == 1 else 40000))
        f = 0.0 if it % 4 == 0 else float(rng.uniform(-8000, 8000))
        u8 = synth.noise_cu8(n, 12000 + it)
        ref = OracleSignalProcessor(fs)
        r = ref.process(synth.cu8_to_c128(u8), f)
        p = procs.setdefault(fs, SignalProcessor(fs))
        h = p.process_cu8(u8, freq_offset=f)
        np.testing.assert_array_equal(h, r)
        assert len(p.symbols) == len(ref.symbols), (fs, n, f)
        if len(ref.symbols):
            assert np.max(np.abs(p.symbols - ref.symbols)) <= 1e-10 * (np.max(np.abs(ref.symbols)) or 1.0), (fs, n, f)


def test_c4_full_batch_every_carrier_vs_oracle():
    """BASELINE config 4 at its full size, the very batch bench.py times: 1024 carriers x 262144 cu8 samples through
    BatchDemodulator.enqueue -- 1024 DISTINCT streams (seeds 1000 + i); every carrier's hard symbols and
    timing phase equal the C oracle's (1024 oracle runs), soft <= 1e-10, the digest bench.py asserts after its timed
    region is this one, and the strong-scaling slices of the batch (8 GPUs x 128) have the digests pinned for them."""
    import bench
    from tetraear_amd.shard import carrier_range
    from tools.make_bench_digest import check_batch
    digest, n_oracle, worst, (hard, n_soft, bp) = check_batch(1024, 262144, 0, want_rows=True)
    assert n_oracle == 1024 and worst <= SOFT_TOL
    want = bench.expected_digest(bench.digest_key(1024, 262144, "cu8", bench.SAMPLE_RATE, 0, False))
    assert want is not None and digest == want
    for world in (2, 4, 8):
        for r in range(world):
            lo, hi = carrier_range(1024, r, world)
            key = bench.digest_key(hi - lo, 262144, "cu8", bench.SAMPLE_RATE, r, False) + f":strong{lo}-{hi}of1024"
            assert bench.expected_digest(key) == bench.output_digest(hard[lo:hi], n_soft[lo:hi], bp[lo:hi]), key


def test_decimate_entry_vs_goldens(gold_stages):
    """tdm_decimate (the stand-alone scipy.signal.decimate entry) against the goldens of the imported scipy, both
    engines: q = 7, 10, 41 run in parallel form, q = 23 on the cascade engine (checked against the oracle)."""
    import ctypes as C
    from oracle import oracle as orc
    from tetraear_amd import _lib, synth
    L = _lib.load()
    g = gold_stages
    x = synth.cu8_to_c128(synth.noise_cu8(4000, int(g["x4000_seed"][0])))

    def dec(q):
        y = np.zeros((len(x) + q - 1) // q, dtype=np.complex128)
        m = C.c_int64()
        _lib.check(L.tdm_decimate(_lib.ptr(x), len(x), q, _lib.ptr(y), C.byref(m), 0))
        assert m.value == len(y)
        return y
    for q in (7, 10, 41):
        ref = g[f"decimate_q{q}"]
        assert np.max(np.abs(dec(q) - ref)) <= 1e-12 * np.max(np.abs(ref)), q
    import scipy.signal as sg                     # (present on the GPU box as in this container; the goldens above do not need it)
    ref = sg.decimate(x, 23)
    assert np.max(np.abs(dec(23) - ref)) <= 1e-12 * np.max(np.abs(ref))


def test_resample_long_inputs_vs_scipy(SP):
    """SignalProcessor.resample (processor.py:35-49 -> scipy.signal.resample, FFT method) on capture-sized arrays: from 2^24
    terms on the transforms run as fast transforms (fft_kernels.hpp: Stockham radix-2 passes for powers of two, Bluestein's
    chirp-z form for every other length) instead of direct sums.  Against scipy itself (present on the GPU box as in the build
    container; the short goldens of test_stage_methods stay the pinned ones): powers of two, composite and PRIME lengths, down- and
    up-sampling, the folded / split Nyquist bin (even kept-bin counts), within 1e-11 of the largest output; a NaN input gives an
    all-NaN output as in the reference; and the time of one 131 072-sample call."""
    import time
    import scipy.signal as sg
    from tetraear_amd import synth
    p = SP(2.4e6)
    rng = np.random.default_rng(77)
    worst = 0.0
    for n, target in ((131072, 1.2e6), (262144, 240000.0), (100000, 1.0e6), (30011, 2.0e6), (65536, 3.1e6), (50000, 2.4e6 * 49999 / 50000),
                      (4100, 1.0e6), (4099, 2.4e6 * 5000 / 4099)):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        y = p.resample(x, target)
        ref = sg.resample(x, int(n * target / 2.4e6))
        assert y.shape == ref.shape and y.dtype == np.complex128, (n, target)
        err = float(np.max(np.abs(y - ref)) / np.max(np.abs(ref)))
        worst = max(worst, err)
        assert err <= 1e-11, (n, target, err)
    x = synth.cu8_to_c128(synth.noise_cu8(131072, 5))
    t0 = time.perf_counter()
    for _ in range(3):
        y = p.resample(x, 1.2e6)
    ms = (time.perf_counter() - t0) / 3 * 1e3
    x[7] = np.nan
    assert np.isnan(p.resample(x, 1.2e6)).all()
    print(f"resample on long inputs: worst error against scipy {worst:.2e} of the largest output; 131072 -> 65536 samples in {ms:.1f} ms per call (host arrays in and out)")
    p.close()


def test_one_plan_serves_ragged_lengths():
    """tdm_plan_resize: one 3-carrier plan walks through 48 chunk lengths (more than it keeps variants for), shorter and
    longer than the length it was made for (work buffers grow), revisits lengths, and every call equals the C oracle;
    fresh SignalProcessor objects (signal/scanner.py:164 makes one per call) share one plan."""
    import time
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    from tetraear_amd.batch import BatchDemodulator
    from tetraear_amd.signal import processor as P
    fs, rows = 2.4e6, 3
    rng = np.random.default_rng(5)
    lens = [40000, 131072, 28, 5000, 262144, 16, 131072, 40000, 300001] + [int(x) for x in rng.integers(100, 200000, 39)]
    bd = BatchDemodulator(fs, lens[0], rows, "cu8")
    ref = OracleSignalProcessor(fs)
    t_new, t_hit, seen = [], [], set()
    for it, n in enumerate(lens):
        t0 = time.perf_counter()
        bd.resize(n)
        dt = (time.perf_counter() - t0) * 1e3
        if n not in seen:
            t_new.append(dt)
        seen.add(n)
        assert bd.info.n_samples == n
        u8 = np.stack([synth.noise_cu8(n, 900 + 7 * it + r) for r in range(rows)])
        fo = [0.0, 1234.5, -3000.25]
        hards, softs, bp, mm = bd.process(u8, freq_offsets=fo)
        for r in range(rows):
            want = ref.process(synth.cu8_to_c128(u8[r]), fo[r])
            np.testing.assert_array_equal(hards[r], want)
            assert len(softs[r]) == len(ref.symbols)
            if len(ref.symbols):
                assert np.max(np.abs(softs[r] - ref.symbols)) <= 1e-10 * (np.max(np.abs(ref.symbols)) or 1.0), (n, r)
    t_hit = []
    for n in lens[-6:]:   # (no buffer growth since these were made: look-ups)
        t0 = time.perf_counter()
        bd.resize(n)
        t_hit.append((time.perf_counter() - t0) * 1e3)
    bd.close()
    print(f"resize: new length {np.median(t_new):.3f} ms (median of {len(t_new)}), seen length {np.median(t_hit):.4f} ms")
    assert np.median(t_new) < 5.0 and np.median(t_hit) < 0.2
    # instances share the plan of their (device, rate, format)
    P.close_plans()
    a, b = P.SignalProcessor(fs), P.SignalProcessor(fs)
    u8 = synth.noise_cu8(30000, 1)
    ha = a.process_cu8(u8)
    t0 = time.perf_counter()
    hb = b.process_cu8(u8)
    dt = (time.perf_counter() - t0) * 1e3
    np.testing.assert_array_equal(ha, hb)
    assert len(P._PLANS) == 1
    print(f"second instance, same length: {dt:.3f} ms per process_cu8 call (no plan built)")


@pytest.mark.gpu
def test_process_dtype_corners_match_reference(SP):
    """process() with complex64, float64 and float32 input against goldens from the imported reference
    (tests/golden/dtypes.npz): hard decisions identical; soft symbols within 1e-10 of the reference's result for the same
    samples as complex128 and within 5e-5 of its single-precision result; a real input without offset gives a real
    `symbols` array, as in the reference."""
    import os
    from tests.golden_cases import DTYPE_CASES, GOLDEN, dtype_case_input
    g = np.load(os.path.join(GOLDEN, "dtypes.npz"))
    for name, (fs, foff) in DTYPE_CASES.items():
        x = dtype_case_input(name)
        p = SP(fs)
        hard = p.process(x, foff)
        np.testing.assert_array_equal(hard, g[name + "__hard"], err_msg=name)
        soft, soft64 = g[name + "__soft"], g[name + "__soft64"]
        scale = np.max(np.abs(soft64))
        assert p.symbols.dtype == soft.dtype, (name, p.symbols.dtype, soft.dtype)
        assert np.max(np.abs(p.symbols - soft64)) <= 1e-10 * scale, name
        assert np.max(np.abs(p.symbols - soft)) <= 5e-5 * scale, name
        p.close()


@pytest.mark.gpu
def test_fast_pre_shift_config3_and_its_guard():
    """Plan option "fast_pre_shift" on BASELINE config 3 (64 carriers out of one 2.4 MS/s stream, shifts up to 787.5 kHz) and on
    bench.py --shared's own workload: hard decisions and timing phase equal to the oracle's p.process(p.frequency_shift(x,
    f_k)) on EVERY carrier, soft symbols within 1e-9 (the exact-phase path: 1e-10), no carrier's smallest margin below the
    1e-8 rad guard -- and the guard itself: with the threshold raised above every margin, process() re-runs the batch with
    the exact phase and returns that path's outputs bit for bit."""
    import bench
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    from tetraear_amd.batch import BatchDemodulator
    n = 262144
    offs = [(k - 31.5) * 25000.0 for k in range(64)]
    u8, _ = synth.multicarrier_cu8(n, 2.4e6, offs, seed0=100)
    exact = BatchDemodulator(2.4e6, n, 64, "cu8")
    h0, s0, bp0, mm0 = exact.process(u8, pre_shifts=offs, shared_input=True)
    exact.close()
    bd = BatchDemodulator(2.4e6, n, 64, "cu8").set_fast_pre_shift()
    hards, softs, bp, mm = bd.process(u8, pre_shifts=offs, shared_input=True)
    assert getattr(bd, "exact_reruns", 0) == 0 and float(np.min(mm)) >= 1e-8
    x = synth.cu8_to_c128(u8)
    worst = 0.0
    for k in range(64):
        o = OracleSignalProcessor(2.4e6)
        ref = o.process(o.frequency_shift(x, offs[k]))
        assert bp[k] == o.best_phase, k
        np.testing.assert_array_equal(hards[k], ref, err_msg=f"carrier {k}")
        worst = max(worst, float(np.max(np.abs(softs[k] - o.symbols)) / np.max(np.abs(o.symbols))))
    assert 1e-13 < worst <= 1e-9, worst
    # the guard: every carrier "too close" -> the exact path's outputs
    bd.FAST_SHIFT_MARGIN = 10.0
    h1, s1, bp1, mm1 = bd.process(u8, pre_shifts=offs, shared_input=True)
    assert bd.exact_reruns == 1
    for k in range(64):
        np.testing.assert_array_equal(h1[k], h0[k])
        np.testing.assert_array_equal(s1[k], s0[k])
    np.testing.assert_array_equal(mm1, mm0)
    bd.close()
    # bench.py --shared's workload through the device path with the option: the digest pinned to the oracle
    carriers = 64
    iq, _ = bench.make_batch(1, n, "cu8", 0)
    pre = bench.shared_offsets(carriers)
    bd = BatchDemodulator(2.4e6, n, carriers, "cu8").set_fast_pre_shift()
    bd.alloc_device_io(shared_input=True)
    bd.upload(iq, freq_offsets=None, pre_shifts=pre)
    bd.enqueue()
    bd.sync()
    hard, soft, n_soft, bpd, mmd = bd.download()
    bd.close()
    want = bench.expected_digest(bench.digest_key(carriers, n, "cu8", 2.4e6, 0, True))
    assert want is not None and bench.output_digest(hard, n_soft, bpd) == want
    assert float(np.min(mmd)) >= 1e-8
