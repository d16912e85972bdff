"""CPU: the oracle (oracle/) against golden vectors made by importing the reference."""
import numpy as np
import pytest

from oracle import design
from oracle.oracle import OracleSignalProcessor, resample_np
from tests.golden_cases import CASES, DTYPE_CASES, GOLDEN, TIMING_DEGENERATE, case_c128, dtype_case_input
from tetraear_amd import synth


def test_design_tables_match_scipy(gold_design):
    g = gold_design
    for k in g.files:
        if not k.startswith("meta_"):
            continue
        key = k[5:]
        fs, q, cur, cutoff, sps = g[k]
        if q > 1:
            sos = design.cheby1_lowpass_sos(8, 0.05, 0.8 / q)
            np.testing.assert_array_equal(sos, g["sos_" + key])
            np.testing.assert_array_equal(design.sosfilt_zi(sos), g["soszi_" + key])
        b, a = design.butter_lowpass_ba(4, cutoff)
        np.testing.assert_array_equal(b, g["b_" + key])
        np.testing.assert_array_equal(a, g["a_" + key])
        np.testing.assert_array_equal(design.lfilter_zi(b, a), g["zi_" + key])
        rp = design.RateParams(fs)
        assert rp.q == int(q) and rp.rate_dec == cur


@pytest.mark.parametrize("name", sorted(CASES))
def test_process_matches_reference(name, gold_process):
    c = CASES[name]
    x = case_c128(c)
    p = OracleSignalProcessor(c["fs"])
    if "pre_shift" in c:
        x = p.frequency_shift(x, c["pre_shift"])
    hard = p.process(x, c["foff"])
    g_hard = gold_process[name + "__hard"]
    g_soft = gold_process[name + "__soft"]
    assert hard.dtype == np.uint8
    if name in TIMING_DEGENERATE:
        assert abs(len(hard) - len(g_hard)) <= 1
        m = min(len(hard), len(g_hard))
        np.testing.assert_array_equal(hard[:m], g_hard[:m])
        return
    np.testing.assert_array_equal(hard, g_hard)          # bit-exact hard decisions
    assert p.symbols.shape == g_soft.shape
    if len(g_soft):
        scale = np.max(np.abs(g_soft)) or 1.0
        assert np.max(np.abs(p.symbols - g_soft)) <= 1e-12 * scale


def test_kat_anchor():
    """SURVEY.md section 8(c) known-answer test, recomputed from the seed."""
    u8 = synth.noise_cu8(131072, 20260929)
    x = synth.cu8_to_c128(u8)
    p = OracleSignalProcessor(2.4e6)
    out = p.process(x, 0)
    assert len(out) == 1007
    assert "".join(map(str, out[:32])) == "33301132333030133230003331013030"
    assert list(np.bincount(out, minlength=4)) == [323, 116, 140, 428]
    assert abs(p.symbols[0] - (-0.762974098502649 + 0.49451661786214185j)) < 1e-13
    out = p.process(x, 1171.875)
    assert "".join(map(str, out[:32])) == "33320033313030033312003310011000"


def test_stage_methods(gold_stages):
    g = gold_stages
    x = synth.cu8_to_c128(synth.noise_cu8(4000, int(g["x4000_seed"][0])))
    p = OracleSignalProcessor(2.4e6)

    def close(a, b, tol=1e-12):
        assert a.shape == b.shape
        if len(b):
            assert np.max(np.abs(a - b)) <= tol * max(1.0, np.max(np.abs(b)))

    close(p.filter_signal(x), g["filter_default"])
    close(p.filter_signal(x, bandwidth=50000), g["filter_bw50k"])
    close(p.filter_signal(x, 25000, 240000.0), g["filter_240k"])
    close(p.filter_signal(x, 25000, 20000.0), g["filter_clamp_hi"])
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
    assert len(p.extract_symbols(x[:5], 240000.0)) == len(g["extract_short"]) == 0
    np.testing.assert_array_equal(p.demodulate_dqpsk(x), g["demod_x"])
    np.testing.assert_array_equal(p.demodulate_dqpsk(x[:2]), g["demod_2"])
    np.testing.assert_array_equal(p.demodulate_dqpsk(np.zeros(10, dtype=complex)), g["demod_zeros"])
    np.testing.assert_array_equal(p.demodulate_dqpsk(g["demod_probe_in"]), g["demod_probe"])
    for q in (7, 10, 41):
        close(p.decimate(x, q), g[f"decimate_q{q}"])
    close(resample_np(x[:1000], 2.4e6, 1.2e6), g["resample_1200k"])
    close(resample_np(x[:301], 2.4e6, 3.0e6), g["resample_up"])


def test_reference_style_contracts():
    """The reference's own (structural) assertions, tests/unit/test_signal_processor.py:14-116."""
    p = OracleSignalProcessor()
    assert p.sample_rate == 2.4e6 and p.symbol_rate == 18000 and p.samples_per_symbol > 0
    assert len(p.filter_signal(np.array([]))) == 0
    assert len(p.demodulate_dqpsk(np.array([]))) == 0
    assert len(p.demodulate_dqpsk(np.array([1 + 1j]))) == 0
    assert len(p.extract_symbols(np.array([]))) == 0
    out = p.process(np.array([]))
    assert out.dtype == np.uint8 and len(out) == 0 and len(p.symbols) == 0


@pytest.mark.parametrize("name", sorted(DTYPE_CASES))
def test_process_dtype_corners_against_reference(name):
    """complex64 / float64 / float32 input (tests/golden/make_golden_dtypes.py: the reference follows its input's dtype
    through scipy.signal.decimate and filters complex64 / float32 input in single precision).  The fp64 chain on the same
    samples -- the oracle here, the device in test_gpu_parity -- gives the reference's complex128 result to 1e-12, its
    hard decisions for the input as handed over, and its single-precision soft symbols to 5e-5 (the reference's own
    float32 noise; north_star's 1e-5 applies to the complex128 input pyrtlsdr delivers)."""
    import os
    g = np.load(os.path.join(GOLDEN, "dtypes.npz"))
    fs, foff = DTYPE_CASES[name]
    x = dtype_case_input(name)
    p = OracleSignalProcessor(fs)
    hard = p.process(np.asarray(x).astype(np.complex128), foff)
    np.testing.assert_array_equal(hard, g[name + "__hard64"])
    np.testing.assert_array_equal(hard, g[name + "__hard"])
    soft64, soft = g[name + "__soft64"], g[name + "__soft"]
    scale = np.max(np.abs(soft64))
    assert np.max(np.abs(p.symbols - soft64)) <= 1e-12 * scale
    assert np.max(np.abs(p.symbols - soft)) <= 5e-5 * scale
    if not np.iscomplexobj(x) and foff == 0:
        assert soft.dtype == np.float64 and np.max(np.abs(p.symbols.imag)) == 0.0   # the reference's `symbols` is a real array
