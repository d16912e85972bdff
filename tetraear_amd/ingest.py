"""Recorded-IQ ingest (BASELINE config 1 "2.4 MS/s recorded IQ file").

The reference has no file reader; the format defined here is rtl_sdr's raw output:
interleaved unsigned bytes I,Q ("cu8"), converted on the GPU exactly as pyrtlsdr converts them.
A capture loop in the reference reads fixed-size chunks and demodulates each one independently
(decrypt_capture.py:101-107: read_samples(256*1024) -> process()); a recording is therefore cut
into the same chunks, which become the rows of pipelined GPU batches.
"""
import numpy as np

from tetraear_amd.batch import BatchDemodulator


def demodulate_recording(source, sample_rate=2.4e6, chunk=256 * 1024, freq_offset=0.0, rows_per_batch=64, device=0):
    """source: path of a cu8 file, or a uint8 array of interleaved I,Q.
    Returns a list with one uint8 symbol array per chunk (what process() returned per read)."""
    u8 = np.fromfile(source, dtype=np.uint8) if isinstance(source, (str, bytes)) else np.ascontiguousarray(source, np.uint8)
    n_chunks = (len(u8) // 2) // chunk
    out = []
    if n_chunks > 0:
        rows = min(rows_per_batch, n_chunks)
        n_batches = n_chunks // rows
        bd = BatchDemodulator(sample_rate, chunk, rows, "cu8", device=device)
        hard, soft, n_soft, bp, mm = bd.process_stream(u8[:2 * chunk * rows * n_batches], n_batches,
                                                       freq_offsets=[float(freq_offset)] * rows)
        for b in range(n_batches):
            for r in range(rows):
                out.append(hard[b, r, :max(int(n_soft[b, r]) - 1, 0)].copy())
        bd.close()
        done = rows * n_batches
        if done < n_chunks:   # remaining chunks: one smaller batch
            rem = n_chunks - done
            bd = BatchDemodulator(sample_rate, chunk, rem, "cu8", device=device)
            hards, _, _, _ = bd.process(u8[2 * chunk * done:2 * chunk * n_chunks], freq_offsets=[float(freq_offset)] * rem)
            out.extend(hards)
            bd.close()
    tail = (len(u8) // 2) - n_chunks * chunk
    if tail > 0:              # the last, shorter read
        bd = BatchDemodulator(sample_rate, tail, 1, "cu8", device=device)
        hards, _, _, _ = bd.process(u8[2 * chunk * n_chunks:2 * (chunk * n_chunks + tail)], freq_offsets=[float(freq_offset)])
        out.extend(hards)
        bd.close()
    return out
