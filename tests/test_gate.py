"""Spectrum / AFC / signal gate: the numpy restatement (oracle/gate_np.py), the kernel
body (CPU emulation) and the GPU path against tests/golden/gate.npz -- outputs of the reference's own
statements (tetraear/ui/modern.py:1919-2021, :2032), executed from its AST by
tests/golden/make_golden_gate.py."""
import os

import numpy as np
import pytest

from oracle import gate_np
from tetraear_amd import synth

FS = 2.4e6
SPECS = [(0.0, 30.0), (3000.0, 30.0), (-2500.0, 5.0), (9000.0, 30.0), (-11000.0, 25.0), (500.0, -5.0)]


GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gate.npz"))
KEYS = ("peak_freq_offset", "signal_power", "peak_power", "noise_floor", "snr")


def test_oracle_matches_reference_block():
    """bit-identical: the restatement performs the reference's numpy operations in the reference's order"""
    n_strong = 0
    for name in GOLD["names"]:
        x = synth.cu8_to_c128(GOLD[f"in_{name}"])
        ref, fs = GOLD[f"out_{name}"], float(GOLD[f"fs_{name}"][0])
        got = gate_np.gate(x, fs)
        for i, k in enumerate(KEYS):
            assert got[k] == ref[i], (name, k)
        assert got["strong"] == bool(ref[5]) and got["afc"] == ref[6], name
        n_strong += int(ref[5])
    assert 0 < n_strong < len(GOLD["names"])


def _golden_groups():
    """golden cases grouped by (sample rate, length) so that one batched call covers a group"""
    groups = {}
    for name in GOLD["names"]:
        u8 = GOLD[f"in_{name}"]
        groups.setdefault((float(GOLD[f"fs_{name}"][0]), len(u8) // 2), []).append(str(name))
    return groups


def test_emul_gate_matches_reference_block():
    from tests.emul import emul
    for (fs, n), names in _golden_groups().items():
        out, afc = emul.gate(np.concatenate([GOLD[f"in_{m}"] for m in names]), "cu8", n, len(names), fs)
        for r, m in enumerate(names):
            ref = GOLD[f"out_{m}"]
            for i in range(5):
                assert abs(out[r][i] - ref[i]) <= 1e-9 * max(1.0, abs(ref[i])), (m, KEYS[i])
            assert bool(out[r][5]) == bool(ref[5]) and out[r][6] == ref[6] == afc[r], m


@pytest.mark.gpu
def test_gpu_gate_matches_reference_block():
    from tetraear_amd.gate import spectrum_gate
    for (fs, n), names in _golden_groups().items():
        res, afc = spectrum_gate(np.concatenate([GOLD[f"in_{m}"] for m in names]), "cu8", n, len(names), fs)
        for r, m in enumerate(names):
            ref = GOLD[f"out_{m}"]
            for i, k in enumerate(KEYS):
                assert abs(res[r][k] - ref[i]) <= 1e-9 * max(1.0, abs(ref[i])), (m, k)
            assert bool(res[r]["strong"]) == bool(ref[5]) and res[r]["afc"] == ref[6] == afc[r], m


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
