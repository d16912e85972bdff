"""Batched, plan-based face of the C-ABI: many carriers per call, optional device-resident I/O.

`BatchDemodulator` is what a multi-carrier capture loop would hold: one plan per
(sample_rate, chunk length, carrier count, wire format), reused for every chunk.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FMT_BYTES, FMT_CF32, FMT_CF64, FMT_CS8, FMT_CU8, MODE_REFERENCE, PlanInfo, check, ptr

_FMT_OF = {"cu8": FMT_CU8, "cs8": FMT_CS8, "cf32": FMT_CF32, "cf64": FMT_CF64}


class DeviceBuffer:
    def __init__(self, device, nbytes):
        self.device = device
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(_lib.load().tdm_dev_alloc(device, self.nbytes, C.byref(p)))
        self.ptr = p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(_lib.load().tdm_dev_upload(self.device, self.ptr, ptr(arr), arr.nbytes))

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(_lib.load().tdm_dev_download(self.device, ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            _lib.load().tdm_dev_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class BatchDemodulator:
    """n_carriers independent streams of n_samples each per call (SignalProcessor.process batched)."""

    def __init__(self, sample_rate, n_samples, n_carriers=1, fmt="cu8", device=0, mode=MODE_REFERENCE):
        self.lib = _lib.load()
        self.fmt = _FMT_OF[fmt] if isinstance(fmt, str) else int(fmt)
        self.device = device
        self.handle = None
        self._dev = None
        h = C.c_void_p()
        check(self.lib.tdm_plan_create(float(sample_rate), int(n_samples), int(n_carriers), self.fmt, mode,
                                       device, C.byref(h)))
        self.handle = h
        self.info = PlanInfo()
        check(self.lib.tdm_plan_get_info(self.handle, C.byref(self.info)))
        self.n_carriers = int(n_carriers)
        self.n_samples = int(n_samples)
        self.mode = mode
        self.soft_dtype = np.complex64 if mode in (_lib.MODE_TETRA, _lib.MODE_TETRA_GARDNER) else np.complex128

    FAST_SHIFT_MARGIN = 1e-8   # rad: below it a decision made with the fast pre-shift is re-made with the exact one

    def set_fast_pre_shift(self, on=True):
        """tdm_plan_option "fast_pre_shift": the input-rate pre-shift's phase as the ideal ramp (a third less arithmetic in
        the decimator of a channelised call); `process` then re-runs, with the exact phase, a batch in which some carrier's
        smallest decision margin is below FAST_SHIFT_MARGIN, so its hard decisions are the reference's either way."""
        check(self.lib.tdm_plan_option(self.handle, b"fast_pre_shift", 1 if on else 0))
        self.fast_pre_shift = bool(on)
        return self

    def set_rows_per_chunk(self, c):
        """tdm_plan_option "rows_per_chunk": the plan's rows are T x c -- c carriers out of each of T consecutive chunks of one
        stream (rows r c ... r c + c - 1 read input row r); pre-shifts and offsets stay per plan row."""
        check(self.lib.tdm_plan_option(self.handle, b"rows_per_chunk", int(c)))
        self.rows_per_chunk = int(c)
        return self

    def set_gardner_ff_start(self, on=True):
        """tdm_plan_option "gardner_ff_start" (MODE_TETRA_GARDNER): the first loop of every chunk starts at a feed-forward timing
        estimate (no hang-up half a symbol off the eye at the start of a chunk); the later pieces of a chunk always do."""
        check(self.lib.tdm_plan_option(self.handle, b"gardner_ff_start", 1 if on else 0))
        return self

    def set_gardner_segments(self, pieces=1):
        """tdm_plan_option "gardner_segments" (MODE_TETRA_GARDNER): 0 whole chunks, 1 the default (from the chunk alone: the
        same symbols whatever the batch), K at most K independently started loops per chunk, -1 fitted to this plan's batch
        and device (fastest; soft symbols behind a seam then depend on the plan's size); `info.gardner_segments` then says
        how many are in force."""
        check(self.lib.tdm_plan_option(self.handle, b"gardner_segments", int(pieces)))
        check(self.lib.tdm_plan_get_info(self.handle, C.byref(self.info)))
        return self

    def resize(self, n_samples):
        """Serve another chunk length with this plan (tdm_plan_resize): tables of a new length are built once (a fraction of
        a millisecond), a length seen before is a look-up; `info` then describes the new length.  Device I/O buffers made
        by alloc_device_io keep their size: call it again if the new length needs more."""
        n_samples = int(n_samples)
        if n_samples != self.n_samples:
            check(self.lib.tdm_plan_resize(self.handle, n_samples))
            check(self.lib.tdm_plan_get_info(self.handle, C.byref(self.info)))
            self.n_samples = n_samples
        return self

    # ---- host-pointer path ------------------------------------------------------------------
    def process(self, iq, freq_offsets=None, pre_shifts=None, shared_input=False):
        """iq: array holding the carriers back to back (or one shared stream). Returns
        (hard_list, soft_list, best_phase, min_margin)."""
        rows, ms = self.n_carriers, self.info.max_soft
        iq = np.ascontiguousarray(iq)
        need = (1 if shared_input else rows // getattr(self, "rows_per_chunk", 1)) * self.n_samples * FMT_BYTES[self.fmt]
        if iq.nbytes < need:
            raise ValueError(f"iq holds {iq.nbytes} bytes, plan needs {need}")
        fo = None if freq_offsets is None else np.ascontiguousarray(freq_offsets, dtype=np.float64)
        ps = None if pre_shifts is None else np.ascontiguousarray(pre_shifts, dtype=np.float64)
        hard = np.zeros((rows, ms), dtype=np.uint8)
        soft = np.zeros((rows, ms), dtype=self.soft_dtype)
        n_soft = np.zeros(rows, dtype=np.int32)
        bp = np.zeros(rows, dtype=np.int32)
        mm = np.zeros(rows, dtype=np.float64)
        check(self.lib.tdm_process(self.handle, ptr(iq), 0 if shared_input else self.n_samples, ptr(ps), ptr(fo),
                                   ptr(hard), ptr(soft), ptr(n_soft), ptr(bp), ptr(mm)))
        if getattr(self, "fast_pre_shift", False) and ps is not None and np.any(mm < self.FAST_SHIFT_MARGIN):
            # a decision this close to a threshold is the reference's only with the reference's own phase rounding
            check(self.lib.tdm_plan_option(self.handle, b"fast_pre_shift", 0))
            try:
                check(self.lib.tdm_process(self.handle, ptr(iq), 0 if shared_input else self.n_samples, ptr(ps), ptr(fo),
                                           ptr(hard), ptr(soft), ptr(n_soft), ptr(bp), ptr(mm)))
            finally:
                check(self.lib.tdm_plan_option(self.handle, b"fast_pre_shift", 1))
            self.exact_reruns = getattr(self, "exact_reruns", 0) + 1
        hards = [hard[r, :max(int(n_soft[r]) - 1, 0)].copy() for r in range(rows)]
        softs = [soft[r, :int(n_soft[r])].copy() for r in range(rows)]
        return hards, softs, bp, mm

    def process_stream(self, iq, n_batches, freq_offsets=None):
        """n_batches batches back to back in host memory, copies overlapped with compute
        (tdm_process_pipelined).  Returns arrays shaped [n_batches][n_carriers]..."""
        rows, ms = self.n_carriers, self.info.max_soft
        iq = np.ascontiguousarray(iq)
        need = n_batches * rows * self.n_samples * FMT_BYTES[self.fmt]
        if iq.nbytes < need:
            raise ValueError(f"iq holds {iq.nbytes} bytes, {need} needed")
        fo = None if freq_offsets is None else np.ascontiguousarray(freq_offsets, dtype=np.float64)
        hard = np.zeros((n_batches, rows, ms), dtype=np.uint8)
        soft = np.zeros((n_batches, rows, ms), dtype=self.soft_dtype)
        n_soft = np.zeros((n_batches, rows), dtype=np.int32)
        bp = np.zeros((n_batches, rows), dtype=np.int32)
        mm = np.zeros((n_batches, rows), dtype=np.float64)
        check(self.lib.tdm_process_pipelined(self.handle, ptr(iq), int(n_batches), ptr(fo), ptr(hard), ptr(soft),
                                             ptr(n_soft), ptr(bp), ptr(mm)))
        return hard, soft, n_soft, bp, mm

    # ---- device-resident path (bench, streaming pipelines) -------------------------------------
    def alloc_device_io(self, shared_input=False):
        rows, ms = self.n_carriers, self.info.max_soft
        d = {}
        d["iq"] = DeviceBuffer(self.device, (1 if shared_input else rows // getattr(self, "rows_per_chunk", 1)) * self.n_samples * FMT_BYTES[self.fmt])
        d["foff"] = DeviceBuffer(self.device, rows * 8)
        d["pre"] = DeviceBuffer(self.device, rows * 8)
        d["hard"] = DeviceBuffer(self.device, rows * ms)
        d["soft"] = DeviceBuffer(self.device, rows * ms * 16)
        d["n_soft"] = DeviceBuffer(self.device, rows * 4)
        d["bp"] = DeviceBuffer(self.device, rows * 4)
        d["mm"] = DeviceBuffer(self.device, rows * 8)
        d["shared"] = shared_input
        d["use_foff"] = False
        d["use_pre"] = False
        self._dev = d
        return d

    def upload(self, iq, freq_offsets=None, pre_shifts=None):
        d = self._dev
        d["iq"].upload(iq)
        d["use_foff"] = freq_offsets is not None
        d["use_pre"] = pre_shifts is not None
        if freq_offsets is not None:
            d["foff"].upload(np.ascontiguousarray(freq_offsets, dtype=np.float64))
        if pre_shifts is not None:
            d["pre"].upload(np.ascontiguousarray(pre_shifts, dtype=np.float64))

    def enqueue(self, iq_ptr=None, stride=None):
        """One pass of the hot path over the resident batch (asynchronous).  `iq_ptr` / `stride`
        (samples between carriers) let the pass read another device buffer, e.g. the channeliser's
        pitched [channels][pitch] output."""
        d = self._dev
        check(self.lib.tdm_process_device(self.handle, iq_ptr if iq_ptr is not None else d["iq"].ptr,
                                          (0 if d["shared"] else self.n_samples) if stride is None else int(stride),
                                          d["pre"].ptr if d["use_pre"] else None,
                                          d["foff"].ptr if d["use_foff"] else None, d["hard"].ptr, d["soft"].ptr,
                                          d["n_soft"].ptr, d["bp"].ptr, d["mm"].ptr, None))

    def enqueue_rows(self, row_list_ptr, n_rows_ptr, iq_ptr=None, stride=None):
        """One pass over the LISTED rows only (TDM_MODE_TETRA; tdm_process_device_rows): row_list / n_rows are device
        buffers, e.g. the occupancy gate's outputs; rows that are not listed are not touched."""
        d = self._dev
        check(self.lib.tdm_process_device_rows(self.handle, iq_ptr if iq_ptr is not None else d["iq"].ptr,
                                               self.n_samples if stride is None else int(stride), row_list_ptr, n_rows_ptr,
                                               d["hard"].ptr, d["soft"].ptr, d["n_soft"].ptr, d["bp"].ptr, d["mm"].ptr, None))

    def enqueue_rrc_filter(self, y_buf, y_pitch, iq_ptr=None, stride=None):
        """TETRA-mode plans: the RRC matched filter alone over the resident batch (tdm_plan_rrc_filter; asynchronous):
        y_buf = DeviceBuffer of n_carriers x y_pitch complex64"""
        d = self._dev
        check(self.lib.tdm_plan_rrc_filter(self.handle, iq_ptr if iq_ptr is not None else d["iq"].ptr,
                                           self.n_samples if stride is None else int(stride), y_buf.ptr, int(y_pitch), None))

    def rrc_filter(self, iq):
        """host in, host out: [n_carriers][n_samples] in the plan's wire format (complex64; for cu8 / cs8 plans interleaved
        bytes) -> the matched filter's output as complex64, same shape"""
        if self.fmt in (FMT_CU8, FMT_CS8):
            iq = np.ascontiguousarray(iq).view(np.uint8).reshape(self.n_carriers, 2 * self.n_samples)
        else:
            iq = np.ascontiguousarray(iq, dtype=np.complex64).reshape(self.n_carriers, self.n_samples)
        pitch = (self.n_samples + 1) & ~1
        din, dout = DeviceBuffer(self.device, iq.nbytes), DeviceBuffer(self.device, self.n_carriers * pitch * 8)
        try:
            din.upload(iq)
            check(self.lib.tdm_plan_rrc_filter(self.handle, din.ptr, self.n_samples, dout.ptr, pitch, None))
            self.sync()
            return dout.download(np.complex64, self.n_carriers * pitch).reshape(self.n_carriers, pitch)[:, :self.n_samples].copy()
        finally:
            din.free()
            dout.free()

    def sync(self):
        check(self.lib.tdm_plan_sync(self.handle))

    def wait_for(self, other):
        """tdm_plan_wait_for: what is enqueued on this plan from now on starts after everything enqueued so far on `other`
        has finished (device-side ordering between two plans' streams)."""
        check(self.lib.tdm_plan_wait_for(self.handle, other.handle))

    def make_stream_current(self):
        """The stand-alone device-pointer entry points (gate, channeliser, find_sync) called from this thread now enqueue on
        this plan's stream, i.e. in order with `enqueue` and without a host synchronisation in between."""
        s = C.c_void_p()
        check(self.lib.tdm_plan_stream(self.handle, C.byref(s)))
        check(self.lib.tdm_set_stream(s))

    def release_stream(self):
        check(self.lib.tdm_set_stream(None))

    def download(self):
        self.sync()   # (the plan's stream does not block the copies below by itself)
        d = self._dev
        rows, ms = self.n_carriers, self.info.max_soft
        n_soft = d["n_soft"].download(np.int32, rows)
        hard = d["hard"].download(np.uint8, rows * ms).reshape(rows, ms)
        soft = d["soft"].download(self.soft_dtype, rows * ms).reshape(rows, ms)
        bp = d["bp"].download(np.int32, rows)
        mm = d["mm"].download(np.float64, rows)
        return hard, soft, n_soft, bp, mm

    def time_begin(self, per_stage=True):
        """HIP-event mark on the plan's stream; per_stage=False leaves the launches back to back as they are untimed."""
        check((self.lib.tdm_plan_time_begin if per_stage else self.lib.tdm_plan_time_begin_total)(self.handle))

    def time_end(self):
        ms = C.c_float()
        check(self.lib.tdm_plan_time_end(self.handle, C.byref(ms)))
        return ms.value

    def stage_times(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = C.c_int32()
        check(self.lib.tdm_plan_stage_times(self.handle, 16, names, ms, C.byref(n)))
        return {names[i].decode(): ms[i] for i in range(n.value)}

    def close(self):
        if getattr(self, "handle", None):
            if self._dev:
                for v in self._dev.values():
                    if isinstance(v, DeviceBuffer):
                        v.free()
                self._dev = None
            self.lib.tdm_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PipelinedBatchDemodulator:
    """`depth` plans of the SAME batch geometry, consecutive steps handed to them in turn: step k runs on plan k % depth, on
    that plan's own stream and work buffers, so up to `depth` steps are in flight (device-resident path).

    Why: behind the decimator (which keeps the fp64 pipes full) a step runs three launches that do not -- carries, low-rate
    stage, finish: 40 % of a step at a third of the issue rate, and launch-latency for small batches.  With the next step's
    decimator running beside them on another stream the device stays busy: measured on 128 / 512 / 1024 carriers x 262 144
    samples (tools/split_bench.py, round 6) one plan 0.163 / 0.459 / 0.836 ms per step, two plans in turn 0.123 / 0.397 /
    0.788, three 0.109 / 0.395 / 0.781; the same batch cut into two HALVES on two streams 0.152 / 0.425 / 0.802 (smaller
    launches lose more than the overlap wins), "whole dispatch rounds on the raw-byte kernel + the rest on the double-based
    one" slower than one plan.  A capture loop gets the same by giving chunk k to plan k % depth -- `upload(..., slot=k)`
    then `enqueue()`; each plan's outputs are those of a lone plan bit for bit (every plan's digest is checked by the
    bench).

    The methods are BatchDemodulator's device-resident ones.  `upload` without a slot fills every plan's input (the bench:
    one resident batch); `download` returns the outputs of the step enqueued last, `download_all` every plan's.
    Per-stage timing (`time_begin(per_stage=True)`) orders the plans ONE AFTER THE OTHER on the device (tdm_plan_wait_for),
    so that every launch is timed alone; `stage_times` then average over the plans.
    """

    def __init__(self, sample_rate, n_samples, n_carriers, fmt="cu8", device=0, depth=3):
        self.plans = []
        try:
            for _ in range(max(1, int(depth))):
                self.plans.append(BatchDemodulator(sample_rate, n_samples, n_carriers, fmt, device))
        except Exception:
            self.close()
            raise
        self.n_carriers, self.n_samples, self.device = int(n_carriers), int(n_samples), device
        self.info = self.plans[0].info
        self.soft_dtype = getattr(self.plans[0], "soft_dtype", np.complex128)
        self._serial = False
        self._turn = 0          # the plan the next step goes to
        self._last = 0          # the plan that ran the step enqueued last

    @property
    def depth(self):
        return len(self.plans)

    def set_fast_pre_shift(self, on=True):
        for p in self.plans:
            p.set_fast_pre_shift(on)
        return self

    def set_rows_per_chunk(self, c):
        for p in self.plans:
            p.set_rows_per_chunk(c)
        return self

    def alloc_device_io(self, shared_input=False):
        for p in self.plans:
            p.alloc_device_io(shared_input)

    def upload(self, iq, freq_offsets=None, pre_shifts=None, slot=None):
        """slot None: every plan's input buffer (one resident batch for all steps); slot k: the input of step k's plan"""
        for p in (self.plans if slot is None else [self.plans[int(slot) % self.depth]]):
            p.upload(iq, freq_offsets, pre_shifts)

    def enqueue(self):
        p = self.plans[self._turn]
        if self._serial and self.depth > 1:
            # per-stage timing: this step starts after the previous one has finished -- on the device, no host round trip
            p.wait_for(self.plans[self._last])
        p.enqueue()
        self._last = self._turn
        self._turn = (self._turn + 1) % self.depth

    def sync(self):
        for p in self.plans:
            p.sync()

    def download(self):
        self.sync()
        return self.plans[self._last].download()

    def download_all(self):
        self.sync()
        return [p.download() for p in self.plans]

    def time_begin(self, per_stage=True):
        self.sync()
        self._serial = bool(per_stage)
        for p in self.plans:
            p.time_begin(per_stage)

    def time_end(self):
        self._serial = False
        return max(p.time_end() for p in self.plans)

    def stage_times(self):
        ts = [t for t in (p.stage_times() for p in self.plans) if t]     # (a plan that took no step of the pass has no times)
        keys = [k for t in ts for k in t]
        return {k: sum(t[k] for t in ts if k in t) / sum(1 for t in ts if k in t) for k in dict.fromkeys(keys)}

    def close(self):
        for p in getattr(self, "plans", []):
            p.close()
        self.plans = []


def batch_demodulator(sample_rate, n_samples, n_carriers, fmt="cu8", device=0, depth="auto"):
    """The plan(s) for a device-resident reference-mode batch that is demodulated step after step: a
    PipelinedBatchDemodulator of three plans (steps in turn, up to three in flight) -- depth "auto" -- or of `depth` plans;
    depth 1 is a plain BatchDemodulator."""
    d = 3 if depth == "auto" else int(depth)
    if d <= 1:
        return BatchDemodulator(sample_rate, n_samples, n_carriers, fmt, device)
    return PipelinedBatchDemodulator(sample_rate, n_samples, n_carriers, fmt, device, depth=d)
