"""TEST INFRASTRUCTURE: ctypes face of tests/emul/libtdm_emul.so (CPU lock-step emulation of
the HIP kernel bodies).  Never imported by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
        _LIB = C.CDLL(os.path.join(_HERE, "libtdm_emul.so"))
    return _LIB


FMT = {"cu8": 0, "cs8": 1, "cf32": 2, "cf64": 3}


def process(sample_rate, iq, fmt, n, rows=1, stride=None, pre_shift=None, freq_offset=None, rows_per_chunk=1):
    L = lib()
    L.emu_rows_per_chunk(int(rows_per_chunk))
    ms = C.c_int32()
    L.emu_process(C.c_double(sample_rate), C.c_int64(n), rows, FMT[fmt], None, C.c_int64(0), None, None,
                  None, None, None, None, None, C.byref(ms))
    ms = ms.value
    hard = np.zeros((rows, ms), dtype=np.uint8)
    soft = np.zeros((rows, ms), dtype=np.complex128)
    n_soft = np.zeros(rows, dtype=np.int32)
    bp = np.zeros(rows, dtype=np.int32)
    mm = np.zeros(rows, dtype=np.float64)
    iq = np.ascontiguousarray(iq)
    ps = None if pre_shift is None else np.ascontiguousarray(pre_shift, dtype=np.float64)
    fo = None if freq_offset is None else np.ascontiguousarray(freq_offset, dtype=np.float64)
    if stride is None:
        stride = n
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    L.emu_process(C.c_double(sample_rate), C.c_int64(n), rows, FMT[fmt], vp(iq), C.c_int64(stride), vp(ps),
                  vp(fo), vp(hard), vp(soft), vp(n_soft), vp(bp), vp(mm), None)
    return hard, soft, n_soft, bp, mm


class fast_pre_shift:
    """`with emul.fast_pre_shift(): ...` -- the plan option of the same name (the input-rate shift's phase as the ideal ramp)"""

    def __enter__(self):
        lib().emu_fast_pre_shift(1)
        return self

    def __exit__(self, *exc):
        lib().emu_fast_pre_shift(0)
        return False


def zp_stage(kind, x, q=10, bandwidth=25000.0, fs=240000.0):
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.complex128)
    n = len(x)
    n_out = (n + q - 1) // q if kind == 0 else n
    y = np.zeros(n_out, dtype=np.complex128)
    rc = L.emu_zp_stage(kind, x.ctypes.data_as(C.c_void_p), C.c_int64(n), q, C.c_double(bandwidth),
                        C.c_double(fs), y.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("input too short")
    return y


def resample(x, num):
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.complex128)
    y = np.zeros(num, dtype=np.complex128)
    L.emu_resample(x.ctypes.data_as(C.c_void_p), C.c_int64(len(x)), C.c_int64(num), y.ctypes.data_as(C.c_void_p))
    return y


def find_sync(units, from_bits, threshold, max_pos=64):
    L = lib()
    u = np.ascontiguousarray(units, dtype=np.uint8)
    pos = np.zeros(max_pos, dtype=np.int32)
    n = C.c_int32()
    mc = C.c_double()
    L.emu_find_sync(u.ctypes.data_as(C.c_void_p), C.c_int64(len(u)), int(from_bits), C.c_double(threshold), max_pos,
                    pos.ctypes.data_as(C.c_void_p), C.byref(n), C.byref(mc))
    return list(pos[:min(n.value, max_pos)]), mc.value


def gate(iq, fmt, n, rows, fs):
    L = lib()
    iq = np.ascontiguousarray(iq)
    out = np.zeros((rows, 8))
    afc = np.zeros(rows)
    L.emu_gate(iq.ctypes.data_as(C.c_void_p), C.c_int64(n), rows, FMT[fmt], C.c_double(fs),
               out.ctypes.data_as(C.c_void_p), afc.ctypes.data_as(C.c_void_p))
    return out, afc


def detect(x, fs):
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.complex128)
    out = np.zeros(8)
    L.emu_detect(x.ctypes.data_as(C.c_void_p), C.c_int64(len(x)), C.c_double(fs), out.ctypes.data_as(C.c_void_p))
    return out


def small_dft(x):
    """SmallDft<N> of small_dft.hpp on the host (sign +, unnormalised)."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    out = np.empty_like(x)
    rc = lib().emu_small_dft(len(x), x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError(f"no SmallDft<{len(x)}>")
    return out
