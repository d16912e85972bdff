"""TETRA mode: soft-symbol precision of the device against the UNQUANTISED fp64 definition.

The device's matched filter multiplies samples and coefficients as sums of two bfloat16 on the matrix cores; its
plan designs 16-bit coefficients (oracle/tetra_np.py coeff16) so that the split loses nothing on the coefficient side.
BASELINE.json's north_star asks for soft values within 1e-5 of the NumPy path: this file holds the device to that
bound against the float64 RRC WITHOUT the 16-bit rounding (rrc_taps(exact=True)), over >= 2000 seeded random carriers
(3..8 samples/symbol, 10..40 dB, random timing and carrier offsets, 2 000..20 000 samples), and prints the
distribution of the per-carrier maximum error, relative to the carrier's largest symbol.
"""
import numpy as np
import pytest

from oracle import tetra_np
from tetraear_amd import synth

N_CARRIERS = 2048
TOL = 1e-5


def _carrier(rng):
    fs = float(rng.choice([54000.0, 63000.0, 72000.0, 75000.0, 80000.0, 90000.0, 108000.0, 126000.0, 144000.0]))
    n = int(rng.integers(2000, 20000))
    snr_db = float(rng.uniform(10.0, 40.0))
    x, _ = synth.dqpsk_baseband(n, fs, int(rng.integers(1 << 30)), timing_offset=float(rng.uniform(-0.5, 0.5)))
    sps = fs / 18000.0
    x = x + np.sqrt(sps / 10 ** (snr_db / 10) / 2) * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    x = x * np.exp(2j * np.pi * float(rng.uniform(-150, 150)) * np.arange(n) / fs)
    return fs, n, (x * float(rng.uniform(0.05, 4.0))).astype(np.complex64)


def test_definition_16bit_coefficients_are_a_minus_96_dB_change():
    """the 16-bit coefficients differ from the float64 RRC by less than 2^-17 of the largest tap, and the two definitions'
    soft symbols by a few 1e-6 of the largest symbol (this is part of the device's error budget below)"""
    for sps in (3.0, 4.0, 4.1667, 5.0, 8.0):
        hq, he = tetra_np.rrc_taps(sps), tetra_np.rrc_taps(sps, exact=True)
        assert np.max(np.abs(hq - he)) < 2.0 ** -17 * np.max(np.abs(he))
    rng = np.random.default_rng(77)
    worst = 0.0
    for _ in range(12):
        fs, n, x = _carrier(rng)
        _, _, iq = tetra_np.demod(x.astype(np.complex128), fs)
        _, _, ie = tetra_np.demod(x.astype(np.complex128), fs, exact_taps=True)
        assert iq["n_sym"] == ie["n_sym"]
        worst = max(worst, float(np.max(np.abs(iq["sym"] - ie["sym"])) / np.max(np.abs(ie["sym"]))))
    assert worst < 5e-6, worst


@pytest.mark.gpu
def test_gpu_soft_symbols_within_1e5_of_the_unquantised_definition():
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    rng = np.random.default_rng(20260930)
    errs, errs_q, hard_diff = [], [], 0
    done = 0
    while done < N_CARRIERS:
        fs, n, x = _carrier(rng)
        bd = BatchDemodulator(fs, n, 1, "cf32", mode=MODE_TETRA)
        hards, softs, timing, margin = bd.process(x)
        bd.close()
        xd = x.astype(np.complex128)
        h_e, _, info_e = tetra_np.demod(xd, fs, exact_taps=True)
        assert len(softs[0]) == info_e["n_sym"], (fs, n)
        scale = float(np.max(np.abs(info_e["sym"])))
        errs.append(float(np.max(np.abs(softs[0] - info_e["sym"]))) / scale)
        hard_diff += int(np.sum(hards[0] != h_e))
        if done % 8 == 0:   # (the 16-bit definition on a subset: how much of the error is the coefficient rounding)
            _, _, info_q = tetra_np.demod(xd, fs)
            errs_q.append(float(np.max(np.abs(softs[0] - info_q["sym"]))) / scale)
        done += 1
    e = np.sort(np.array(errs))
    q = np.sort(np.array(errs_q))
    print(f"\nsoft error vs UNQUANTISED fp64 definition over {len(e)} carriers: median {np.median(e):.2e}  "
          f"p90 {e[int(0.9 * len(e))]:.2e}  p99 {e[int(0.99 * len(e))]:.2e}  max {e[-1]:.2e}   "
          f"(vs 16-bit-coefficient definition, {len(q)} carriers: median {np.median(q):.2e} max {q[-1]:.2e}); "
          f"hard decisions differing from the unquantised definition: {hard_diff}")
    assert e[-1] <= TOL, e[-5:]
