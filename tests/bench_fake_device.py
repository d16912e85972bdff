"""TEST INFRASTRUCTURE (run by bench.py only when TDM_BENCH_TEST_HOOK names this file): replaces the device with a stand-in
that returns fixed outputs, and librccl's device buffer with host memory, so that bench.py's LAUNCH logic -- self-spawned
local ranks, rendezvous, reductions, the one JSON line -- runs as a plain command on a box without a GPU."""
import ctypes as C
import os

import numpy as np

import tetraear_amd.batch as batch
import tetraear_amd.rccl as rccl


class _Info:
    n_dec, dec_engine, max_soft = 410, 3, 34


class FakeBatchDemodulator:
    def __init__(self, rate, chunk, carriers, fmt, device=0, **kw):
        self.carriers, self.info = carriers, _Info()
        self.rank = int(os.environ.get("RANK", "0"))
        self.fmt, self.soft_dtype = 0, np.complex128

    def sync(self): pass
    def wait_for(self, other): pass
    def alloc_device_io(self, shared_input=False): pass
    def upload(self, iq, freq_offsets=None, pre_shifts=None): pass
    def enqueue(self): pass
    def time_begin(self, per_stage=True): pass
    def time_end(self): return 1.0 + self.rank
    def stage_times(self): return {"dec_block": 0.5, "dec_carry": 0.01, "lpf_block": 0.3, "finish": 0.03}
    def close(self): pass

    def download(self):
        rows, ms = self.carriers, self.info.max_soft
        hard = np.full((rows, ms), self.rank, dtype=np.uint8)
        n_soft = np.full(rows, 30 + self.rank, dtype=np.int32)
        return hard, np.zeros((rows, ms), np.complex128), n_soft, np.zeros(rows, np.int32), np.zeros(rows)


class _HostMem:
    def __init__(self, device=0):
        self.buf = (C.c_byte * 16)()
        self.ptr = C.cast(self.buf, C.c_void_p)

    def upload(self, v): C.memmove(self.buf, C.byref(v), 8)
    def download(self, v): C.memmove(C.byref(v), self.buf, 8)
    def free(self): pass


batch.BatchDemodulator = FakeBatchDemodulator
rccl.DeviceMemory = _HostMem
bench.make_batch = lambda carriers, chunk, fmt, first=0, workers=None: (np.zeros(2 * carriers * chunk, np.uint8), np.zeros(carriers))  # noqa: F821
