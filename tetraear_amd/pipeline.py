"""The reference's live-decode step for many carriers at once, chained on the device:

    spectrum / AFC / signal gate   tetraear/ui/modern.py:1921-2021   (tdm_spectrum_gate)
    process(samples, afc_offset)   tetraear/ui/modern.py:2030-2033   (tdm_process_device, offsets read on the device)
    symbols_to_bits + find_sync    tetraear/core/decoder.py:840-858  (tdm_find_sync, threshold ladder)

One upload of the IQ chunk, one download of the results; the gate's AFC offsets and the hard symbols
never visit the host between the stages.  What the reference does after find_sync (burst parsing, MAC,
crypto, voice) takes these positions and symbols unchanged.
"""
import ctypes as C

import numpy as np

from . import _lib, sync as _sync
from ._lib import check
from .batch import BatchDemodulator, DeviceBuffer
from .gate import FIELDS

LADDER = (0.90, 0.85, 0.80)   # decoder.py:845-853


class CaptureChain:
    def __init__(self, sample_rate, n_samples, n_carriers=1, fmt="cu8", device=0, max_pos=64):
        self.lib = _lib.load()
        self.device = device
        self.rows, self.n, self.fs = int(n_carriers), int(n_samples), float(sample_rate)
        self.max_pos = int(max_pos)
        self.bd = BatchDemodulator(sample_rate, n_samples, n_carriers, fmt, device=device)
        self.dev = self.bd.alloc_device_io()
        self.dev["use_foff"] = True              # process() reads the gate's AFC offsets from device memory
        rows = self.rows
        self.d_gate = DeviceBuffer(device, rows * 8 * 8)
        self.d_pos = [DeviceBuffer(device, rows * self.max_pos * 4) for _ in LADDER]
        self.d_npos = [DeviceBuffer(device, rows * 4) for _ in LADDER]
        self.d_mc = [DeviceBuffer(device, rows * 8) for _ in LADDER]

    def step(self, iq):
        """iq: the carriers' chunks back to back in the plan's wire format.  Returns one dict per carrier:
        the gate's figures, `symbols` (uint8, None when the gate did not pass), `sync_positions`, `max_corr`."""
        L, d, rows = self.lib, self.dev, self.rows
        d["iq"].upload(iq)
        # one stream, no host round trip: gate -> process (AFC offsets read on the device) -> sync ladder on the hard
        # symbols in place (symbol counts read on the device: n_soft as it is, from_bits = 2)
        self.bd.make_stream_current()
        try:
            check(L.tdm_spectrum_gate(d["iq"].ptr, self.bd.fmt, self.n, self.n, rows, self.fs, self.d_gate.ptr,
                                      d["foff"].ptr, 1, self.device))
            self.bd.enqueue()
            ms = self.bd.info.max_soft
            for i, thr in enumerate(LADDER):
                check(L.tdm_find_sync(d["hard"].ptr, ms, d["n_soft"].ptr, rows, 2, float(thr), self.max_pos,
                                      self.d_pos[i].ptr, self.d_npos[i].ptr, self.d_mc[i].ptr, 1, self.device))
        finally:
            self.bd.release_stream()
        self.bd.sync()
        n_soft = d["n_soft"].download(np.int32, rows)
        n_units = np.maximum(n_soft - 1, 0).astype(np.int32)
        gate = self.d_gate.download(np.float64, rows * 8).reshape(rows, 8)
        hard = d["hard"].download(np.uint8, rows * ms).reshape(rows, ms)
        pos = [b.download(np.int32, rows * self.max_pos).reshape(rows, self.max_pos) for b in self.d_pos]
        npos = [b.download(np.int32, rows) for b in self.d_npos]
        mc = [b.download(np.float64, rows) for b in self.d_mc]
        out = []
        for r in range(rows):
            res = {k: float(gate[r, i]) for i, k in enumerate(FIELDS)}
            res["strong"] = bool(gate[r, 5] != 0.0)
            res.update(symbols=None, sync_positions=[], max_corr=0.0)
            if res["strong"]:                                    # ui/modern.py:2022: process only when signal_present
                sym = hard[r, :n_units[r]].copy()
                res["symbols"] = sym
                p, m = [], 0.0
                for i in range(len(LADDER)):
                    p = [int(v) for v in pos[i][r, :min(int(npos[i][r]), self.max_pos)]]
                    m = float(mc[i][r])
                    if p:
                        break
                if not p and m >= 0.75:                          # decoder.py:854-857 (per-carrier threshold: rare)
                    p, _ = _sync.find_sync_symbols(sym, max(0.75, m - 0.02), True, self.device)
                res["sync_positions"], res["max_corr"] = p, m
            out.append(res)
        return out

    def close(self):
        self.bd.close()
        for b in [self.d_gate] + self.d_pos + self.d_npos + self.d_mc:
            b.free()
