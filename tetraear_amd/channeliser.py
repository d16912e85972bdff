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
