"""Burst synchronisation on the GPU: drop-in for
`TetraDecoder.find_sync` (tetraear/core/decoder.py:171-295) and the threshold ladder of
`TetraDecoder.decode` (decoder.py:845-858), fed either with the bit stream the reference builds
or directly with the demodulator's hard symbols."""
import numpy as np

from tetraear_amd import _lib
from tetraear_amd._lib import check, ptr


def _run(units, from_bits, threshold, device=0):
    u = np.ascontiguousarray(units, dtype=np.uint8)
    n = np.array([len(u)], dtype=np.int32)
    max_pos = max(4, len(u) * (1 if from_bits else 2) // 250 + 4)
    pos = np.zeros(max_pos, dtype=np.int32)
    n_pos = np.zeros(1, dtype=np.int32)
    mc = np.zeros(1, dtype=np.float64)
    if len(u) == 0:
        return [], 0.0
    check(_lib.load().tdm_find_sync(ptr(u), len(u), ptr(n), 1, 1 if from_bits else 0, float(threshold), max_pos,
                                    ptr(pos), ptr(n_pos), ptr(mc), 0, device))
    return [int(p) for p in pos[:min(int(n_pos[0]), max_pos)]], float(mc[0])


def find_sync(bits, threshold=0.85, return_max_corr=False, device=0):
    """Same contract as TetraDecoder.find_sync(bits, threshold, return_max_corr)."""
    b = np.asarray(bits)
    if len(b) < 22:
        return ([], 0.0) if return_max_corr else []
    # values other than 0/1 can never equal a pattern element; keep them distinguishable in a byte
    u = np.where((b == 0) | (b == 1), b, 2).astype(np.uint8)
    pos, mc = _run(u, True, threshold, device)
    return (pos, mc) if return_max_corr else pos


def find_sync_symbols(symbols, threshold=0.85, return_max_corr=False, device=0):
    """find_sync(symbols_to_bits(symbols)[0], ...) without materialising the bits on the host."""
    s = np.asarray(symbols)
    if 2 * len(s) < 22:
        return ([], 0.0) if return_max_corr else []
    pos, mc = _run(s, False, threshold, device)
    return (pos, mc) if return_max_corr else pos


def sync_ladder(symbols, device=0):
    """The threshold sequence TetraDecoder.decode applies (decoder.py:845-858)."""
    pos, mc = find_sync_symbols(symbols, 0.90, True, device)
    if not pos:
        pos, mc = find_sync_symbols(symbols, 0.85, True, device)
        if not pos:
            pos, mc = find_sync_symbols(symbols, 0.80, True, device)
            if not pos and mc >= 0.75:
                pos, _ = find_sync_symbols(symbols, max(0.75, mc - 0.02), True, device)
    return pos, mc
