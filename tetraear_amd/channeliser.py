"""Tetra-mode channeliser (oversampled polyphase DFT filter bank) -- host face of tdm_channelise."""
import ctypes as C

import numpy as np

from tetraear_amd import _lib
from tetraear_amd._lib import FMT_BYTES, check, ptr

_FMT_OF = {"cu8": 0, "cs8": 1, "cf32": 2}


def channelise(iq, fmt, M, D, device=0):
    """One wideband stream -> complex64 [M][ceil(n/D)]; channel k is centred on k*fs/M."""
    f = _FMT_OF[fmt]
    iq = np.ascontiguousarray(iq)
    n_in = iq.nbytes // FMT_BYTES[f]
    n_out = (n_in + D - 1) // D
    out = np.zeros((M, n_out), dtype=np.complex64)
    no = C.c_int64()
    check(_lib.load().tdm_channelise(ptr(iq), f, n_in, M, D, ptr(out), C.byref(no), 0, device))
    assert no.value == n_out
    return out


def aligned_pitch(n_out):
    """Row pitch (complex samples) that keeps the kernel's 128-byte stores inside one cache line."""
    return (n_out + 15) // 16 * 16


def channelise_batch(iq, fmt, n_streams, M, D, device=0, pitch=0):
    """n_streams wideband streams back to back -> complex64 [n_streams][M][ceil(n/D)] in one launch
    (a view of a [n_streams][M][pitch] array when a row pitch is given)."""
    f = _FMT_OF[fmt]
    iq = np.ascontiguousarray(iq)
    n_in = iq.nbytes // FMT_BYTES[f] // n_streams
    n_out = (n_in + D - 1) // D
    out = np.zeros((n_streams, M, pitch or n_out), dtype=np.complex64)
    no = C.c_int64()
    check(_lib.load().tdm_channelise_batch(ptr(iq), f, n_in, n_streams, M, D, ptr(out), pitch, C.byref(no), 0, device))
    assert no.value == n_out
    return out[:, :, :n_out]
