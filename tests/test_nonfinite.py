"""Non-finite input samples (NaN / Inf): goldens made by importing the reference (tests/golden/make_golden_nonfinite.py).

The reference has no guard: `scipy.signal.decimate` (processor.py:254) and `filtfilt` (:79) carry ONE such sample over the whole
chunk, every slicer comparison is then false (:152-161 -> symbol 3), no timing phase beats `max_power = -1` (:196-210 -> phase
0); where no filter runs the NaN stays where it is and numpy's comparison / `np.max` semantics decide.  CPU tier: the oracle
and the kernel bodies in lock-step emulation; GPU tier: SignalProcessor and the C-ABI."""
import os
import warnings

import numpy as np
import pytest

from tests.golden_cases import GOLDEN, NONFINITE_CASES, nonfinite_case_input, nonfinite_stage_inputs


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "nonfinite.npz"))


def same(a, b, tol=1e-12):
    """equal shapes, the same samples non-finite (and of the same kind), the finite ones within tol of the largest"""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.dtype == np.uint8 or b.dtype == np.uint8:
        np.testing.assert_array_equal(a, b)
        return
    ac, bc = a.astype(complex), b.astype(complex)
    for part in (np.real, np.imag):
        pa, pb = part(ac), part(bc)
        np.testing.assert_array_equal(np.isnan(pa), np.isnan(pb))
        np.testing.assert_array_equal(np.isposinf(pa), np.isposinf(pb))
        np.testing.assert_array_equal(np.isneginf(pa), np.isneginf(pb))
    fin = np.isfinite(bc)
    if fin.any():
        assert np.max(np.abs(ac[fin] - bc[fin])) <= tol * max(1.0, np.max(np.abs(bc[fin])))


def check_methods(p, resample, decimate, gold, tol=1e-12):
    st = nonfinite_stage_inputs()
    for tag in ("nan", "inf"):
        x = st["x4000_" + tag]
        same(p.filter_signal(x), gold[f"st_filter_{tag}"], tol)
        same(p.filter_signal(x, 25000, 240000.0), gold[f"st_filter_240k_{tag}"], tol)
        same(p.filter_signal(x[1225:1240]), gold[f"st_filter_short15_{tag}"], tol)
        same(p.frequency_shift(x, 1000), gold[f"st_shift_{tag}"], tol)
        same(decimate(x, 10), gold[f"st_decimate_q10_{tag}"], tol)
        same(decimate(x, 7), gold[f"st_decimate_q7_{tag}"], tol)
        same(resample(x[:2000]), gold[f"st_resample_{tag}"], tol)
        same(p.extract_symbols(x, 240000.0), gold[f"st_extract_{tag}"], 0)
        same(p.demodulate_dqpsk(x), gold[f"st_demod_{tag}"])
    same(p.extract_symbols(st["x4000_nan_some_phases"], 240000.0), gold["st_extract_nan_some_phases"], 0)
    same(p.extract_symbols(st["x4000_nan_some_phases"], 300000.0), gold["st_extract_nan_some_phases_300k"], 0)
    for k in ("sym200_nan", "sym200_inf", "sym200_nan_ends"):
        same(p.demodulate_dqpsk(st[k]), gold["st_demod_" + k])


# ---------------------------------------------------------------------------------------------- CPU tier: the oracle
@pytest.mark.parametrize("name", sorted(NONFINITE_CASES))
def test_oracle_process_nonfinite(name, gold):
    from oracle.oracle import OracleSignalProcessor
    fs, foff = NONFINITE_CASES[name][:2]
    p = OracleSignalProcessor(fs)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hard = p.process(nonfinite_case_input(name), foff)
    same(hard, gold[name + "__hard"])
    g = gold[name + "__soft"]
    same(p.symbols.real if g.dtype == np.float64 else p.symbols, g)


def test_oracle_methods_nonfinite(gold):
    from oracle.oracle import OracleSignalProcessor, resample_np
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p = OracleSignalProcessor(2.4e6)
        check_methods(p, lambda x: resample_np(x, 2.4e6, 1.2e6), p.decimate, gold)


# ------------------------------------------------------- CPU tier: the device's kernel bodies in lock-step emulation
EMUL_CASES = sorted(n for n, c in NONFINITE_CASES.items() if c[2] <= 65536 and c[4] == "c128" and c[2] > 0)


@pytest.mark.parametrize("name", EMUL_CASES)
def test_emul_process_nonfinite(name, gold):
    from tests.emul import emul
    fs, foff, n = NONFINITE_CASES[name][:3]
    hard, soft, n_soft, bp, mm = emul.process(fs, nonfinite_case_input(name), "cf64", n, freq_offset=[foff])
    ns = int(n_soft[0])
    g_soft = gold[name + "__soft"]
    assert ns == len(g_soft)
    same(hard[0, :max(ns - 1, 0)], gold[name + "__hard"])
    same(soft[0, :ns], g_soft, 1e-10)
    if len(g_soft) and not np.isfinite(g_soft).any():
        assert int(bp[0]) == 0            # processor.py:196-210: no phase beats max_power = -1


# ----------------------------------------------------------------------------------------------------------- GPU tier
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(NONFINITE_CASES))
def test_gpu_process_nonfinite(name, gold):
    """SignalProcessor.process (complex128 / complex64 / float64 arrays, the kept API) on the device against the reference's
    goldens: all-NaN `symbols`, symbol 3 throughout, timing phase 0 wherever a zero-phase filter ran."""
    from tetraear_amd.signal import SignalProcessor
    fs, foff = NONFINITE_CASES[name][:2]
    p = SignalProcessor(fs)
    hard = p.process(nonfinite_case_input(name), foff)
    g_soft = gold[name + "__soft"]
    same(hard, gold[name + "__hard"])
    assert p.symbols.dtype == g_soft.dtype, (p.symbols.dtype, g_soft.dtype)
    same(p.symbols, g_soft, 1e-10)
    if len(g_soft) and not np.isfinite(g_soft).any():
        assert p.best_phase == 0
    p.close()


@pytest.mark.gpu
def test_gpu_methods_nonfinite(gold):
    from tetraear_amd.signal import SignalProcessor
    import ctypes as C
    from tetraear_amd import _lib
    p = SignalProcessor(2.4e6)

    def decimate(x, q):      # scipy.signal.decimate's stand-alone entry point
        y = np.zeros((len(x) + q - 1) // q, dtype=np.complex128)
        m = C.c_int64()
        _lib.check(_lib.load().tdm_decimate(_lib.ptr(x), len(x), q, _lib.ptr(y), C.byref(m), 0))
        return y[:m.value]
    check_methods(p, lambda x: p.resample(x, 1.2e6), decimate, gold, 1e-10)
    p.close()


@pytest.mark.gpu
def test_gpu_batch_one_poisoned_carrier_leaves_the_others_alone(gold):
    """A batch of 8 complex64 carriers, two of them with a NaN / an Inf: those two come out as the reference's all-NaN chunk,
    the other six bit-identical to a batch without the poison (through the C-ABI's device-resident path)."""
    from tetraear_amd import synth
    from tetraear_amd.batch import BatchDemodulator
    rows, n = 8, 65536
    x = np.stack([synth.cu8_to_c128(synth.noise_cu8(n, 6300 + r)) for r in range(rows)]).astype(np.complex64)
    foffs = np.linspace(-2000, 2000, rows)
    outs = []
    for poison in (False, True):
        xx = x.copy()
        if poison:
            xx[2, 40000] = np.nan
            xx[5, 7] = complex(1.0, np.inf)
        bd = BatchDemodulator(2.4e6, n, rows, "cf32")
        bd.alloc_device_io()
        bd.upload(xx.reshape(-1), freq_offsets=foffs)
        bd.enqueue()
        outs.append(bd.download())
        bd.enqueue()                       # (a second pass over the same plan: nothing sticks to it)
        again = bd.download()
        np.testing.assert_array_equal(outs[-1][0], again[0])
        bd.close()
    (h0, s0, n0, b0, m0), (h1, s1, n1, b1, m1) = outs
    for r in range(rows):
        ns = int(n1[r])
        if r in (2, 5):
            assert ns == (n // 10 + (n % 10 > 0)) // 13 and b1[r] == 0
            assert np.isnan(s1[r, :ns].real).all() and np.isnan(s1[r, :ns].imag).all()
            assert (h1[r, :ns - 1] == 3).all()
        else:
            assert ns == int(n0[r]) and b1[r] == b0[r] and m1[r] == m0[r]
            np.testing.assert_array_equal(h1[r, :ns - 1], h0[r, :ns - 1])
            np.testing.assert_array_equal(s1[r, :ns], s0[r, :ns])
