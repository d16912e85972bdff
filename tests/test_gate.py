"""Spectrum / AFC / signal gate: kernel body (CPU emulation) and GPU path against the
numpy restatement of tetraear/ui/modern.py:1921-2021 (oracle/gate_np.py; parity unpinned: the block is
inline GUI-thread code that cannot be imported)."""
import numpy as np
import pytest

from oracle import gate_np
from tetraear_amd import synth

FS = 2.4e6
SPECS = [(0.0, 30.0), (3000.0, 30.0), (-2500.0, 5.0), (9000.0, 30.0), (-11000.0, 25.0), (500.0, -5.0)]


def _rows(n):
    return [synth.dqpsk_cu8(n, FS, seed=60 + r, carrier_offset=co, esn0_db=snr)[0] for r, (co, snr) in enumerate(SPECS)]


def _check(out_row, afc, ref):
    for i, k in enumerate(("peak_freq_offset", "signal_power", "peak_power", "noise_floor", "snr")):
        assert abs(out_row[i] - ref[k]) <= 1e-9 * max(1.0, abs(ref[k])), k
    assert bool(out_row[5]) == ref["strong"]
    assert out_row[6] == ref["afc"] == afc


def test_emul_gate_matches_restatement():
    from tests.emul import emul
    n = 4096
    xs = _rows(n)
    out, afc = emul.gate(np.concatenate(xs), "cu8", n, len(xs), FS)
    strong = 0
    for r, u8 in enumerate(xs):
        ref = gate_np.gate(synth.cu8_to_c128(u8), FS)
        _check(out[r], afc[r], ref)
        strong += ref["strong"]
    assert 0 < strong < len(xs)                      # both branches of the rule are exercised
    out, afc = emul.gate(np.concatenate([x[:2 * 1000] for x in xs]), "cu8", 1000, len(xs), FS)
    assert np.all(out == 0) and np.all(afc == 0)     # len(samples) < n_fft: block skipped


@pytest.mark.gpu
def test_gpu_gate_matches_restatement_and_feeds_process():
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd.batch import BatchDemodulator
    from tetraear_amd.gate import spectrum_gate
    n = 131072
    xs = _rows(n)
    res, afc = spectrum_gate(np.concatenate(xs), "cu8", n, len(xs), FS)
    for r, u8 in enumerate(xs):
        ref = gate_np.gate(synth.cu8_to_c128(u8), FS)
        row = [res[r][k] for k in ("peak_freq_offset", "signal_power", "peak_power", "noise_floor", "snr")]
        row += [res[r]["strong"], res[r]["afc"]]
        _check(row, afc[r], ref)
    # the reference's loop: process(samples, freq_offset=afc_offset) for the rows that passed the gate
    bd = BatchDemodulator(FS, n, len(xs), "cu8")
    hards, softs, bp, mm = bd.process(np.concatenate(xs), freq_offsets=afc)
    for r, u8 in enumerate(xs):
        if res[r]["strong"]:
            ref = OracleSignalProcessor(FS).process(synth.cu8_to_c128(u8), afc[r])
            np.testing.assert_array_equal(hards[r], ref)
    bd.close()
    # complex128 rows, as the reference's caller holds them
    x = np.concatenate([synth.cu8_to_c128(u) for u in xs])
    res2, afc2 = spectrum_gate(x, "cf64", n, len(xs), FS)
    assert np.array_equal(afc, afc2)
