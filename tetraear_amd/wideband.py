"""Wideband receiver: polyphase channeliser -> TETRA-mode demodulation of every channel, with the
channelised samples handed over on the device (pitched rows, no host round trip).

BASELINE config 5 (10 MS/s -> 400 x 25 kHz carriers) and config 3's tetra-mode counterpart
(2.4 MS/s -> 96 channels).  No counterpart in the reference: defined by
oracle/pfb_np.py and oracle/tetra_np.py.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FMT_BYTES, MODE_TETRA, check
from .batch import BatchDemodulator, DeviceBuffer
from .channeliser import aligned_pitch

_FMT_OF = {"cu8": 0, "cs8": 1, "cf32": 2}


class WidebandReceiver:
    """`streams` wideband streams of `n_in` samples at `sample_rate` -> M channels each, spaced
    sample_rate/M and decimated by D (channel rate sample_rate/D), all demodulated per call."""

    def __init__(self, sample_rate, n_in, M, D, streams=1, fmt="cu8", device=0, slots=1, group=None, mode=MODE_TETRA,
                 gated=False, snr_db=15.0, min_dbfs=-70.0, gardner_ff_start=False):
        """mode: MODE_TETRA (feed-forward timing, the default) or MODE_TETRA_GARDNER (Gardner detector + loop) for the channels'
        demodulation; gardner_ff_start: that loop started at a feed-forward timing estimate in every chunk (the plan option of
        the same name: no hang-up at a chunk's head).

        slots = 2: two independent (channel buffer, demodulator plan) pairs, each with its own stream.  Consecutive
        batches go to alternating slots (enqueue(slot=k % 2)), so the channeliser of batch k+1 -- bound by its output
        stores -- runs beside the demodulation of batch k -- bound by instruction issue -- instead of behind it.

        group = G (a divisor of `streams`): the batch is worked off in sub-batches of G streams (enqueue_grouped): the
        channel rows of a sub-batch (G x M x pitch x 8 bytes: 27 MB per 10 MS/s stream) are written into a buffer small
        enough to stay in the 256 MB Infinity Cache and read back from there by the sub-batch's demodulation, instead of
        860 MB per 32-stream batch going out to HBM and coming back; sub-batches alternate between the slots, outputs of
        all sub-batches land in ONE set of output buffers (`out`)."""
        self.lib = _lib.load()
        self.fmt = _FMT_OF[fmt]
        self.device = device
        # gated: the reference demodulates only when its gate sees a signal (ui/modern.py:1921-2022).  For a channeliser's
        # rows: tdm_occupancy_gate decides on the device which rows are occupied and the receiver is launched over those
        # rows only (tdm_process_device_rows); the other rows report no symbols.  (TDM_MODE_TETRA, group == streams.)
        self.gated, self.snr_db, self.min_dbfs = bool(gated), float(snr_db), float(min_dbfs)
        if self.gated and (mode != MODE_TETRA or group):
            raise ValueError("gated: the feed-forward receiver (MODE_TETRA) over the whole batch")
        self.sample_rate, self.n_in, self.M, self.D, self.streams = float(sample_rate), int(n_in), int(M), int(D), int(streams)
        self.n_out = (self.n_in + self.D - 1) // self.D
        self.pitch = aligned_pitch(self.n_out)
        self.d_in = DeviceBuffer(device, self.streams * self.n_in * FMT_BYTES[self.fmt])
        self.group = int(group) if group else self.streams
        if self.streams % self.group:
            raise ValueError("group must divide streams")
        self.slots = []
        for _ in range(int(slots)):
            d_ch = DeviceBuffer(device, self.group * self.M * self.pitch * 8)
            demod = BatchDemodulator(self.sample_rate / self.D, self.n_out, self.group * self.M, "cf32",
                                     device=device, mode=mode)
            if gardner_ff_start:
                demod.set_gardner_ff_start(True)
            if self.group == self.streams:
                demod.alloc_device_io()
            self.slots.append((d_ch, demod))
            if self.gated:
                rows = self.streams * self.M
                demod.occ = {"stats": DeviceBuffer(device, rows * 8), "flags": DeviceBuffer(device, rows),
                             "rows": DeviceBuffer(device, rows * 4), "n": DeviceBuffer(device, 4)}
        self.d_ch, self.demod = self.slots[0]
        self.out = None
        if self.group != self.streams:
            rows, ms = self.streams * self.M, self.demod.info.max_soft
            self.out = {"hard": DeviceBuffer(device, rows * ms), "soft": DeviceBuffer(device, rows * ms * 8),
                        "n_soft": DeviceBuffer(device, rows * 4), "bp": DeviceBuffer(device, rows * 4),
                        "mm": DeviceBuffer(device, rows * 8)}

    def process(self, iq):
        """iq: the streams back to back in the plan's wire format.  Returns (hard, n_sym, timing, margin):
        hard uint8 [streams][M][max_sym] with n_sym [streams][M] valid symbols per channel."""
        iq = np.ascontiguousarray(iq)
        if iq.nbytes != self.d_in.nbytes:
            raise ValueError(f"iq holds {iq.nbytes} bytes, receiver needs {self.d_in.nbytes}")
        self.d_in.upload(iq)
        if self.out is not None:      # (group < streams: the slots' channel buffers hold one sub-batch each)
            self.enqueue_grouped()
            self.sync()
            hard, soft, n_soft, timing, margin = self.download_grouped()
        else:
            self.enqueue()
            self.demod.sync()
            hard, soft, n_soft, timing, margin = self.demod.download()
        shape = (self.streams, self.M)
        n_sym = np.maximum(n_soft - 1, 0).reshape(shape)
        return hard.reshape(shape + (-1,)), n_sym, timing.reshape(shape), margin.reshape(shape)

    def enqueue(self, slot=0, d_in=None):
        """channeliser and demodulator back to back on the slot's demodulator plan's stream (no host synchronisation);
        `d_in`: another device input buffer than the receiver's own (a capture loop's second read buffer)"""
        if self.out is not None:
            raise ValueError("this receiver works in sub-batches of `group` streams (its channel buffers hold one sub-batch): "
                             "use enqueue_grouped() / download_grouped()")
        no = C.c_int64()
        d_ch, demod = self.slots[slot]
        demod.make_stream_current()
        try:
            check(self.lib.tdm_channelise_batch((d_in or self.d_in).ptr, self.fmt, self.n_in, self.streams, self.M, self.D,
                                                d_ch.ptr, self.pitch, C.byref(no), 1, self.device))
            if self.gated:
                o = demod.occ
                check(self.lib.tdm_occupancy_gate(d_ch.ptr, self.pitch, self.streams, self.M, self.n_out, self.sample_rate / self.D,
                                                  self.snr_db, self.min_dbfs, o["stats"].ptr, o["flags"].ptr, o["rows"].ptr,
                                                  o["n"].ptr, demod._dev["n_soft"].ptr, 1, self.device))
                demod.enqueue_rows(o["rows"].ptr, o["n"].ptr, iq_ptr=d_ch.ptr, stride=self.pitch)
            else:
                demod.enqueue(iq_ptr=d_ch.ptr, stride=self.pitch)
        finally:
            demod.release_stream()

    def enqueue_grouped(self, d_in=None):
        """the whole batch as streams / group sub-batches, alternating between the slots (see __init__)"""
        if self.out is None:
            raise ValueError("enqueue_grouped() needs a receiver made with group < streams; this one takes the whole batch "
                             "per call: use enqueue() / demod.download()")
        no = C.c_int64()
        rows_g, ms = self.group * self.M, self.demod.info.max_soft
        in_bytes = self.group * self.n_in * FMT_BYTES[self.fmt]
        base = (d_in or self.d_in).ptr.value
        o = self.out
        for g in range(self.streams // self.group):
            d_ch, demod = self.slots[g % len(self.slots)]
            demod.make_stream_current()
            try:
                check(self.lib.tdm_channelise_batch(C.c_void_p(base + g * in_bytes), self.fmt, self.n_in, self.group, self.M,
                                                    self.D, d_ch.ptr, self.pitch, C.byref(no), 1, self.device))
                r0 = g * rows_g
                check(self.lib.tdm_process_device(demod.handle, d_ch.ptr, self.pitch, None, None,
                                                  C.c_void_p(o["hard"].ptr.value + r0 * ms),
                                                  C.c_void_p(o["soft"].ptr.value + r0 * ms * 8),
                                                  C.c_void_p(o["n_soft"].ptr.value + r0 * 4),
                                                  C.c_void_p(o["bp"].ptr.value + r0 * 4),
                                                  C.c_void_p(o["mm"].ptr.value + r0 * 8), None))
            finally:
                demod.release_stream()

    def download_grouped(self):
        if self.out is None:
            raise ValueError("download_grouped() needs a receiver made with group < streams")
        rows, ms = self.streams * self.M, self.demod.info.max_soft
        o = self.out
        n_soft = o["n_soft"].download(np.int32, rows)
        hard = o["hard"].download(np.uint8, rows * ms).reshape(rows, ms)
        soft = o["soft"].download(np.complex64, rows * ms).reshape(rows, ms)
        return hard, soft, n_soft, o["bp"].download(np.int32, rows), o["mm"].download(np.float64, rows)

    def occupancy(self, slot=0):
        """(gated receivers, after a sync) the gate's outputs of the slot's last batch: signal_power and peak_power in dBFS
        [streams][M], occupied [streams][M] bool"""
        o = self.slots[slot][1].occ
        rows, shape = self.streams * self.M, (self.streams, self.M)
        st = o["stats"].download(np.float32, rows * 2).reshape(rows, 2)
        return st[:, 0].reshape(shape), st[:, 1].reshape(shape), o["flags"].download(np.uint8, rows).astype(bool).reshape(shape)

    def sync(self):
        for _, demod in self.slots:
            demod.sync()

    def channel_frequency(self, k):
        """centre frequency of channel k relative to the stream's centre [Hz]"""
        return (k if k < self.M // 2 else k - self.M) * self.sample_rate / self.M

    def close(self):
        for d_ch, demod in self.slots:
            for b in getattr(demod, "occ", {}).values():
                b.free()
            demod.close()
            d_ch.free()
        self.slots = []
        if self.out:
            for b in self.out.values():
                b.free()
            self.out = None
        self.d_in.free()
