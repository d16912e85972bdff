#!/usr/bin/env python3
"""BASELINE config 1 golden, made by IMPORTING the reference (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_recording.py

The "recorded IQ file" of SURVEY.md 8(d) C1: 10 s at 2.4 MS/s = 24 000 000 cu8 samples, here the integer-only seeded
stream synth.noise_cu8(24e6, seed 1) (bit-reproducible on any host, so the 48 MB never need to be stored), cut into the
reference's reads of 131 072 samples (ui/modern.py:1912): 183 reads and a last one of 13 824.  Each read goes through
`tetraear.signal.processor.SignalProcessor(2.4e6).process(read, 1171.875)` exactly as the capture loops call it
(decrypt_capture.py:101-107).  Stored: the hard symbols of the first four reads and of the last, shorter one, and the
sha256 over all 184 outputs in order (tests/golden/recording.npz).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from tetraear.signal.processor import SignalProcessor  # noqa: E402  (the reference)
from tetraear_amd import synth  # noqa: E402

N, CHUNK, SEED, FOFF = 24_000_000, 131072, 1, 1171.875


def main():
    u8 = synth.noise_cu8(N, SEED)
    p = SignalProcessor(2.4e6)
    h = hashlib.sha256()
    outs = []
    n_reads, tail = divmod(N, CHUNK)
    for i in range(n_reads + (1 if tail else 0)):
        lo = 2 * CHUNK * i
        hi = min(lo + 2 * CHUNK, 2 * N)
        hard = p.process(synth.cu8_to_c128(u8[lo:hi]), FOFF)
        h.update(np.int32(len(hard)).tobytes())
        h.update(np.ascontiguousarray(hard, dtype=np.uint8).tobytes())
        outs.append(np.asarray(hard, dtype=np.uint8))
        if i % 20 == 0:
            print(f"read {i}: {len(hard)} symbols", flush=True)
    np.savez_compressed(os.path.join(HERE, "recording.npz"), n=np.int64(N), chunk=np.int64(CHUNK), seed=np.int64(SEED),
                        foff=np.float64(FOFF), n_outputs=np.int64(len(outs)), sha256_all=np.array(h.hexdigest()),
                        first0=outs[0], first1=outs[1], first2=outs[2], first3=outs[3], last=outs[-1],
                        lengths=np.array([len(o) for o in outs], dtype=np.int32))
    print("reads", len(outs), "tail symbols", len(outs[-1]), "sha256", h.hexdigest()[:16])


if __name__ == "__main__":
    main()
