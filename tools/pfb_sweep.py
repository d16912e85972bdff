"""Channeliser throughput over the built configurations (runs on the GPU box): python tools/pfb_sweep.py"""
import ctypes as C, time, sys
import numpy as np
sys.path.insert(0, '.')
from tetraear_amd import _lib
from tetraear_amd.batch import DeviceBuffer
L = _lib.load()
for (M, D, fmt, fb, n_in, streams) in ((96, 32, 0, 2, 1 << 20, 64), (96, 32, 2, 8, 1 << 20, 64), (128, 40, 0, 2, 1 << 20, 64), (80, 25, 0, 2, 1 << 20, 64), (72, 24, 0, 2, 1 << 20, 64), (400, 125, 2, 8, 1 << 20, 32), (400, 100, 0, 2, 1 << 20, 32)):
    n_out = (n_in + D - 1) // D
    pitch = (n_out + 15) // 16 * 16
    din = DeviceBuffer(0, streams * n_in * fb); dout = DeviceBuffer(0, streams * M * pitch * 8)
    din.upload(np.random.default_rng(0).integers(0, 255, streams * n_in * fb, dtype=np.uint8) if fmt != 2 else np.random.default_rng(0).standard_normal(streams * n_in * 2).astype(np.float32))
    no = C.c_int64()
    def step():
        _lib.check(L.tdm_channelise_batch(din.ptr, fmt, n_in, streams, M, D, dout.ptr, pitch, C.byref(no), 1, 0))
    for _ in range(3): step()
    _lib.check(L.tdm_dev_sync(0)); t0 = time.perf_counter()
    for _ in range(20): step()
    _lib.check(L.tdm_dev_sync(0)); dt = (time.perf_counter() - t0) / 20
    byts = streams * (n_in * fb + M * n_out * 8)
    print(f"M={M} D={D} fmt={fmt} streams={streams}: {dt*1e3:.3f} ms  {streams*n_in/dt/1e9:.1f} Gsample/s  {byts/dt/1e12:.2f} TB/s")
    din.free(); dout.free()
