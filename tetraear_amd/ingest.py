"""Recorded-IQ ingest (BASELINE config 1 "2.4 MS/s recorded IQ file").

The reference has no file reader; the format defined here is rtl_sdr's raw output:
interleaved unsigned bytes I,Q ("cu8"), converted on the GPU exactly as pyrtlsdr converts them.
A capture loop in the reference reads fixed-size chunks and demodulates each one independently
(decrypt_capture.py:101-107: `samples = capture.read_samples(chunk_size); processor.process(samples)`;
ui/modern.py:1908-1912 reads 128 Ki samples per turn), so a recording is cut into the same reads.

`iter_recording` is that loop as a generator over a file, a pipe or an array: it never holds more than two batches of
reads in memory.  Two page-locked host buffers are filled in turn by a reader thread while the GPU works on the other
one (rows of a batch = consecutive reads); ONE plan serves the whole recording -- the remainder batch runs on the
same plan with its unused rows blank, the last, shorter read through tdm_plan_resize.
"""
import os
import threading

import numpy as np

from tetraear_amd import _lib
from tetraear_amd._lib import check, ptr
from tetraear_amd.batch import BatchDemodulator


def _open(source):
    """-> (readinto(buffer) -> bytes read, close())"""
    if isinstance(source, (str, bytes, os.PathLike)):
        f = open(os.fspath(source), "rb", buffering=0)
        return f.readinto, f.close
    if hasattr(source, "readinto"):          # file object, pipe (sys.stdin.buffer), socket file
        return source.readinto, (lambda: None)
    # an array of interleaved bytes: handed out through a view with an offset (io.BytesIO would copy the recording)
    src = memoryview(np.ascontiguousarray(source, dtype=np.uint8).reshape(-1))
    pos = [0]

    def readinto(view):
        k = min(len(view), len(src) - pos[0])
        view[:k] = src[pos[0]:pos[0] + k]
        pos[0] += k
        return k
    return readinto, (lambda: None)


def _fill(readinto, view):
    """read until `view` is full or the source ends (a pipe hands out short reads); returns the byte count"""
    got = 0
    while got < len(view):
        k = readinto(view[got:])
        if not k:
            break
        got += k
    return got


def iter_recording(source, sample_rate=2.4e6, chunk=256 * 1024, freq_offset=0.0, rows_per_batch=64, device=0, pre_shifts=None):
    """source: path of a cu8 file, an object with readinto() (open file, pipe), or a uint8 array of interleaved I,Q.
    Yields, in order, one uint8 symbol array per read of `chunk` samples -- what process() returned per read in the
    reference's loop -- and last the shorter final read, if the recording does not end on a read boundary.

    pre_shifts (a list of C input-rate offsets in Hz): the recording is a wideband stream with C carriers in it -- BASELINE
    config 3 read chunk after chunk.  Every read is then demodulated C times, once per carrier, as the reference's
    `p.process(p.frequency_shift(samples, f_k), freq_offset)` would, `rows_per_batch` reads x C carriers per call (plan option
    rows_per_chunk), and what is yielded per read is a LIST of C symbol arrays."""
    readinto, close = _open(source)
    lib = _lib.load()
    rows = int(rows_per_batch)
    ncar = 0 if pre_shifts is None else len(pre_shifts)
    batch_bytes = 2 * chunk * rows
    bufs = [np.zeros(batch_bytes, dtype=np.uint8) for _ in range(2)]
    pinned = []
    bd = None
    t = None     # the reader thread in flight, if any (joined before the buffers go away, also when the consumer stops early)
    try:
        for b in bufs:
            check(lib.tdm_host_register(device, ptr(b), b.nbytes))
            pinned.append(b)
        bd = BatchDemodulator(sample_rate, chunk, rows * max(ncar, 1), "cu8", device=device)
        if ncar:
            bd.set_rows_per_chunk(ncar)
        foffs = [float(freq_offset)] * (rows * max(ncar, 1))
        pre = None if not ncar else np.tile(np.asarray(pre_shifts, dtype=np.float64), rows)
        # reader thread: fills the buffer the GPU is not working on
        filled = [0, 0]
        state = {"err": None}

        def read_into(slot):
            try:
                filled[slot] = _fill(readinto, memoryview(bufs[slot]))
            except Exception as e:  # noqa: BLE001 -- re-raised by the consumer
                state["err"] = e
                filled[slot] = 0

        read_into(0)
        slot = 0
        while True:
            if state["err"]:
                raise state["err"]
            got = filled[slot]
            if got == 0:
                break
            t = None
            if got == batch_bytes:               # (a short batch means the source has ended)
                t = threading.Thread(target=read_into, args=(slot ^ 1,), daemon=True)
                t.start()
            n_reads, tail = divmod(got // 2, chunk)
            if n_reads:
                if n_reads < rows:
                    bufs[slot][2 * chunk * n_reads + 2 * tail:] = 128     # blank rows (mid-scale bytes); their output is dropped
                hards, _, _, _ = bd.resize(chunk).process(bufs[slot], freq_offsets=foffs, pre_shifts=pre)
                for r in range(n_reads):
                    yield hards[r] if not ncar else hards[r * ncar:(r + 1) * ncar]
            if tail:
                # the last, shorter read of the recording: same plan, another chunk length, row 0
                seg = bufs[slot][2 * chunk * n_reads: 2 * (chunk * n_reads + tail)].copy()
                bufs[slot][:] = 128
                bufs[slot][:2 * tail] = seg
                bd.resize(tail)
                hards, _, _, _ = bd.process(bufs[slot][:2 * tail * rows], freq_offsets=foffs, pre_shifts=pre)
                yield hards[0] if not ncar else hards[:ncar]
            if t is None:
                break
            t.join()
            t = None
            slot ^= 1
    finally:
        # the consumer stopped early (or an error): the reader thread may sit in a pipe read that never returns.  It is a
        # daemon; it gets a moment to finish, and if it does not, the buffer it writes into is left registered and alive
        # (held by the thread's closure) rather than unpinned under it.
        stuck = False
        if t is not None:
            t.join(2.0)
            stuck = t.is_alive()
        if bd is not None:
            bd.close()
        if not stuck:
            for b in pinned:
                lib.tdm_host_unregister(device, ptr(b))
            close()


def demodulate_recording(source, sample_rate=2.4e6, chunk=256 * 1024, freq_offset=0.0, rows_per_batch=64, device=0, pre_shifts=None):
    """the whole recording at once: a list with one uint8 symbol array (or, with pre_shifts, one list of them) per read (see
    iter_recording)"""
    return list(iter_recording(source, sample_rate, chunk, freq_offset, rows_per_batch, device, pre_shifts))
