"""CPU tier: the matched filter's operand table a TETRA-mode plan uploads (csrc/tetra_taps.hpp, compiled into the test
emulation library) against the definition's own split of the taps (oracle/tetra_np.py: coefficients = sum of two
bfloat16, round to nearest even) and the lane layout documented in csrc/tetra_kernels.hpp."""
import ctypes as C

import numpy as np
import pytest

from oracle import tetra_np
from tests.emul import emul


def _table(taps):
    L = emul.lib()
    L.emu_tetra_tap_operands.restype = C.c_int64
    t = np.ascontiguousarray(taps, dtype=np.float32)
    n = L.emu_tetra_tap_operands(t.ctypes.data_as(C.c_void_p), len(t), None)
    out = np.zeros(n, dtype=np.uint32)
    L.emu_tetra_tap_operands(t.ctypes.data_as(C.c_void_p), len(t), out.ctypes.data_as(C.c_void_p))
    return out


def _bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


@pytest.mark.parametrize("fs,nt", [(36000.0, 17), (72000.0, 33), (75000.0, 35), (108000.0, 49), (144000.0, 65)])
def test_tap_operand_table_is_the_definitions_split(fs, nt):
    h = tetra_np.rrc_taps(fs / 18000.0)                      # the definition's 16-bit coefficients
    pad = (nt - len(h)) // 2
    assert pad >= 0
    taps = np.zeros(nt, dtype=np.float32)
    taps[pad:pad + len(h)] = h.astype(np.float32)
    tab = _table(taps)
    ks = (16 + nt - 1 + 31) // 32
    assert tab.size == ks * 2 * 64 * 4
    tab = tab.reshape(ks, 2, 64, 4)
    hi = tetra_np._bf16(taps)
    lo = tetra_np._bf16(taps - hi)
    assert np.array_equal(hi.astype(np.float64) + lo.astype(np.float64), taps.astype(np.float64))   # taps ARE such sums
    for s in range(ks):
        for lane in range(64):
            for j in range(4):
                for c in range(2):
                    t = 32 * s + 8 * (lane >> 4) + 2 * j + c - (lane & 15)
                    want_hi, want_lo = (hi[t], lo[t]) if 0 <= t < nt else (np.float32(0), np.float32(0))
                    got_hi = _bf16_to_f32(np.uint16((tab[s, 0, lane, j] >> (16 * c)) & 0xFFFF))
                    got_lo = _bf16_to_f32(np.uint16((tab[s, 1, lane, j] >> (16 * c)) & 0xFFFF))
                    assert got_hi == want_hi and got_lo == want_lo, (s, lane, j, c)


def test_bf16_rounding_matches_numpy_on_random_floats():
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(95) * np.exp(rng.uniform(-20, 20, 95))).astype(np.float32)
    x[:3] = [0.0, 1.0, -1.0]
    # a 95-tap "filter" of arbitrary floats: operand (s = 0, lane = 0, j, c) is tap 2 j + c
    tab = _table(x).reshape(-1, 2, 64, 4)
    hi = tetra_np._bf16(x)
    lo = tetra_np._bf16(x - hi)
    for j in range(4):
        for c in range(2):
            t = 2 * j + c
            assert _bf16_to_f32(np.uint16((tab[0, 0, 0, j] >> (16 * c)) & 0xFFFF)) == hi[t]
            assert _bf16_to_f32(np.uint16((tab[0, 1, 0, j] >> (16 * c)) & 0xFFFF)) == lo[t]
