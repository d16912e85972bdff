"""Burst sync: device kernel bodies (CPU emulation) and the GPU path against
golden vectors made by running the reference's own find_sync."""
import os

import numpy as np
import pytest

from tests.golden.make_golden_sync import make_case

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "sync.npz"))
TS1 = [1, 1, 0, 1, 0, 0, 0, 0, 1, 1, 1, 0, 1, 0, 0, 1, 1, 1, 0, 1, 0, 0]


def _cases():
    for seed in G["seeds"]:
        sym = make_case(int(seed))
        for ti, thr in enumerate(G["thresholds"]):
            yield int(seed), sym, float(thr), G[f"pos_{seed}_{ti}"], float(G[f"mc_{seed}_{ti}"][0])


def test_emul_find_sync_matches_reference():
    from tests.emul import emul
    n_found = 0
    for seed, sym, thr, g_pos, g_mc in _cases():
        pos, mc = emul.find_sync(sym, False, thr, max_pos=64)
        assert pos == list(g_pos), (seed, thr)
        assert mc == g_mc, (seed, thr)          # counts/22 in double: exact
        bits = np.empty(2 * len(sym), dtype=np.uint8)
        bits[0::2] = sym >> 1
        bits[1::2] = sym & 1
        pos_b, mc_b = emul.find_sync(bits, True, thr, max_pos=64)
        assert pos_b == pos and mc_b == mc
        n_found += len(pos)
    assert n_found > 500
    bits = np.zeros(100, dtype=np.uint8)
    bits[20:42] = TS1
    pos, mc = emul.find_sync(bits, True, 0.8)
    assert pos == list(G["unit_pos"]) and mc == float(G["unit_mc"][0])


@pytest.mark.gpu
def test_gpu_find_sync_matches_reference():
    from tetraear_amd import sync
    for seed, sym, thr, g_pos, g_mc in _cases():
        pos, mc = sync.find_sync_symbols(sym, thr, return_max_corr=True)
        assert pos == list(g_pos) and mc == g_mc, (seed, thr)
    # reference unit tests (tests/unit/test_tetra_decoder.py:43-66)
    assert sync.find_sync(np.array([0] * 10)) == []
    bits = np.array([0] * 100)
    bits[20:42] = TS1
    pos, mc = sync.find_sync(bits, threshold=0.8, return_max_corr=True)
    assert isinstance(pos, list) and isinstance(mc, float)
    assert pos == list(G["unit_pos"]) and mc == float(G["unit_mc"][0])
    rnd = np.random.default_rng(3).integers(0, 2, size=100)
    assert isinstance(sync.find_sync(rnd, threshold=0.9), list)


@pytest.mark.gpu
def test_gpu_sync_batched_on_demodulator_output():
    """process() -> find_sync chained: rows of hard symbols straight into the batched entry point.  (A plumbing test: the
    checker is the kernel body compiled for the CPU, which test_emul_find_sync_matches_reference pins to the 300 golden
    cases made by the reference's own find_sync; parity of the device kernel itself is test_gpu_find_sync_matches_reference.)"""
    import ctypes as C
    from tests.emul import emul
    from tetraear_amd import _lib, synth
    from tetraear_amd.batch import BatchDemodulator
    rows, n = 6, 131072
    u8 = np.concatenate([synth.dqpsk_cu8(n, 2.4e6, seed=40 + r)[0] for r in range(rows)])
    bd = BatchDemodulator(2.4e6, n, rows, "cu8")
    hards, softs, bp, mm = bd.process(u8)
    ms = max(len(h) for h in hards)
    units = np.zeros((rows, ms), dtype=np.uint8)
    nun = np.zeros(rows, dtype=np.int32)
    for r in range(rows):
        units[r, :len(hards[r])] = hards[r]
        nun[r] = len(hards[r])
    pos = np.zeros((rows, 32), dtype=np.int32)
    npos = np.zeros(rows, dtype=np.int32)
    mc = np.zeros(rows)
    _lib.check(_lib.load().tdm_find_sync(_lib.ptr(units), ms, _lib.ptr(nun), rows, 0, 0.8, 32, _lib.ptr(pos),
                                         _lib.ptr(npos), _lib.ptr(mc), 0, 0))
    for r in range(rows):
        e_pos, e_mc = emul.find_sync(hards[r], False, 0.8, max_pos=32)
        assert list(pos[r, :npos[r]]) == e_pos and mc[r] == e_mc
    bd.close()


@pytest.mark.gpu
def test_gpu_capture_chain_gate_process_sync():
    """gate -> process(afc) -> find_sync ladder chained on the device (tetraear_amd.pipeline.CaptureChain) against
    the composition of the pinned CPU pieces: oracle/gate_np.py, the C oracle of process(), and the
    find_sync kernel body in CPU emulation (golden-checked above)."""
    from oracle import gate_np
    from oracle.oracle import OracleSignalProcessor
    from tests.emul import emul
    from tetraear_amd import synth
    from tetraear_amd.pipeline import CaptureChain, LADDER
    fs, n = 2.4e6, 131072
    specs = [(0.0, 30.0), (3000.0, 30.0), (-2500.0, 5.0), (9000.0, 30.0), (500.0, -5.0), (-11000.0, 25.0)]
    xs = [synth.dqpsk_cu8(n, fs, seed=80 + r, carrier_offset=co, esn0_db=snr)[0] for r, (co, snr) in enumerate(specs)]
    # plant a training sequence in one strong row's data so that the ladder has something to find
    ch = CaptureChain(fs, n, len(xs), "cu8")
    res = ch.step(np.concatenate(xs))
    n_strong = 0
    for r, u8 in enumerate(xs):
        x = synth.cu8_to_c128(u8)
        g = gate_np.gate(x, fs)
        assert res[r]["strong"] == g["strong"] and res[r]["afc"] == g["afc"], r
        if not g["strong"]:
            assert res[r]["symbols"] is None and res[r]["sync_positions"] == []
            continue
        n_strong += 1
        sym = OracleSignalProcessor(fs).process(x, g["afc"])
        np.testing.assert_array_equal(res[r]["symbols"], sym)
        pos, mc = [], 0.0
        for thr in LADDER:
            pos, mc = emul.find_sync(sym, False, thr, max_pos=64)
            if pos:
                break
        if not pos and mc >= 0.75:
            pos, _ = emul.find_sync(sym, False, max(0.75, mc - 0.02), max_pos=64)
        assert res[r]["sync_positions"] == pos and res[r]["max_corr"] == mc, r
    assert 0 < n_strong < len(xs)
    ch.close()
