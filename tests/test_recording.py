"""BASELINE config 1: the 10 s / 24 M-sample 2.4 MS/s cu8 recording, cut into the reference's 131 072-sample reads
(183 reads + one of 13 824 samples).  Golden = the imported reference run over every read
(tests/golden/make_golden_recording.py): hard symbols of the first four reads and the last, sha256 over all 184.

CPU tier: the C oracle against that golden.  GPU tier: the streaming reader (tetraear_amd/ingest.py iter_recording) over
the recording as a FILE and as a PIPE, one plan, two page-locked buffers, against the same golden.
"""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from tetraear_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "recording.npz"), allow_pickle=False)
N, CHUNK, SEED, FOFF = int(G["n"]), int(G["chunk"]), int(G["seed"]), float(G["foff"])


def _digest(outs):
    h = hashlib.sha256()
    for o in outs:
        h.update(np.int32(len(o)).tobytes())
        h.update(np.ascontiguousarray(o, dtype=np.uint8).tobytes())
    return h.hexdigest()


def _check(outs):
    assert len(outs) == int(G["n_outputs"]) == 184
    assert [len(o) for o in outs] == G["lengths"].tolist()
    for i in range(4):
        np.testing.assert_array_equal(outs[i], G[f"first{i}"], err_msg=f"read {i}")
    np.testing.assert_array_equal(outs[-1], G["last"])
    assert _digest(outs) == str(G["sha256_all"])


def test_oracle_reproduces_the_reference_over_the_whole_recording():
    from oracle.oracle import OracleSignalProcessor
    u8 = synth.noise_cu8(N, SEED)
    o = OracleSignalProcessor(2.4e6)
    outs = []
    for lo in range(0, 2 * N, 2 * CHUNK):
        outs.append(o.process(synth.cu8_to_c128(u8[lo:lo + 2 * CHUNK]), FOFF))
    _check(outs)


@pytest.mark.gpu
def test_gpu_streaming_reader_file(tmp_path):
    from tetraear_amd.ingest import iter_recording
    path = tmp_path / "capture_10s.cu8"
    synth.noise_cu8(N, SEED).tofile(path)
    outs = []
    for hard in iter_recording(str(path), 2.4e6, chunk=CHUNK, freq_offset=FOFF, rows_per_batch=32):
        outs.append(hard)          # (a consumer would hand each read's symbols to the burst synchroniser here)
    _check(outs)


@pytest.mark.gpu
def test_gpu_streaming_reader_pipe_and_array(tmp_path):
    """the same recording through a pipe (short reads) and as an in-memory array, other batch sizes: the rows of a batch
    are independent reads, so the batch geometry changes nothing"""
    from tetraear_amd.ingest import demodulate_recording, iter_recording
    u8 = synth.noise_cu8(N, SEED)
    path = tmp_path / "capture_10s.cu8"
    u8.tofile(path)
    proc = subprocess.Popen([sys.executable, "-c", f"import sys,shutil; shutil.copyfileobj(open({str(path)!r},'rb'), sys.stdout.buffer, 70001)"],
                            stdout=subprocess.PIPE)
    try:
        outs = list(iter_recording(proc.stdout, 2.4e6, chunk=CHUNK, freq_offset=FOFF, rows_per_batch=64))
    finally:
        proc.stdout.close()
        proc.wait()
    _check(outs)
    _check(demodulate_recording(u8, 2.4e6, chunk=CHUNK, freq_offset=FOFF, rows_per_batch=7))


@pytest.mark.gpu
def test_gpu_streaming_reader_many_carriers_out_of_one_recording():
    """BASELINE config 3 read chunk after chunk: a wideband recording with four carriers in it, `iter_recording(...,
    pre_shifts=[...])` -- every read demodulated once per carrier, `rows_per_batch` reads x 4 carriers per call (plan option
    rows_per_chunk) -- against the oracle's p.process(p.frequency_shift(read, f_k), freq_offset) for every read and carrier,
    including a last read that is shorter than the others, and a remainder batch."""
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd.ingest import iter_recording
    chunk, offs, foff = 32768, [-312500.0, -37500.0, 62500.0, 287500.0], 1171.875
    n = 7 * chunk + 9001                       # 2 batches of 3 reads, a remainder batch of 1 read, a short read
    u8, _ = synth.multicarrier_cu8(n, 2.4e6, offs, seed0=400)
    outs = list(iter_recording(u8, 2.4e6, chunk, foff, rows_per_batch=3, pre_shifts=offs))
    assert len(outs) == 8 and all(len(o) == 4 for o in outs)
    x = synth.cu8_to_c128(u8)
    for i, per_carrier in enumerate(outs):
        seg = x[i * chunk:(i + 1) * chunk]
        for k, f in enumerate(offs):
            o = OracleSignalProcessor(2.4e6)
            ref = o.process(o.frequency_shift(seg, f), foff)
            np.testing.assert_array_equal(per_carrier[k], ref, err_msg=f"read {i} carrier {k}")
    assert len(outs[-1][0]) == (9001 // 10 + 1) // 13 - 1
