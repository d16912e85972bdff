"""CPU tier: the source shims of the recording reader (tetraear_amd/ingest.py `_open`, `_fill`) -- no device needed."""
import pathlib

import numpy as np

from tetraear_amd import ingest


def test_array_source_is_read_in_place_in_order_and_ends():
    a = np.arange(1000, dtype=np.uint8)
    readinto, close = ingest._open(a)
    out = np.zeros(384, dtype=np.uint8)
    got = []
    while True:
        k = ingest._fill(readinto, memoryview(out))
        got.append(out[:k].copy())
        if k < len(out):
            break
    assert np.array_equal(np.concatenate(got), a)
    assert readinto(memoryview(out)) == 0      # exhausted: zero, not an error
    close()
    # in place: a later change of the caller's array is what the reader hands out (no private copy was taken)
    b = np.zeros(64, dtype=np.uint8)
    readinto, _ = ingest._open(b)
    b[:] = 7
    assert readinto(memoryview(out)) == 64 and np.all(out[:64] == 7)


def test_pathlike_and_str_sources_open_the_file(tmp_path):
    p = pathlib.Path(tmp_path) / "x.cu8"
    data = np.random.default_rng(1).integers(0, 256, 4096, dtype=np.uint8)
    data.tofile(p)
    for src in (p, str(p), str(p).encode()):
        readinto, close = ingest._open(src)
        out = np.zeros(5000, dtype=np.uint8)
        assert ingest._fill(readinto, memoryview(out)) == 4096
        assert np.array_equal(out[:4096], data)
        close()
