#!/usr/bin/env python3
"""Golden vectors for the input-dtype corners of SignalProcessor.process (run in the build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dtypes.py

`process()` follows its input's dtype through `scipy.signal.decimate` (processor.py:254: the SOS is cast to x.dtype), so
  * a complex64 array is decimated in SINGLE precision (everything behind the decimator is complex128 again), and
  * a real float64 array stays real up to the frequency shift (and for freq_offset == 0 to the end: `symbols` is a real array).
The reference is imported read-only; inputs are seeded (tests/golden_cases.py), outputs stored.  Writes tests/golden/dtypes.npz:
  (inputs: tests/golden_cases.py dtype_case_input, seeded)
  <case>__hard    uint8 decisions of the reference
  <case>__soft    its `symbols` attribute (complex128, or float64 for the real case without offset)
  <case>__hard64 / __soft64   the same samples handed over as complex128 (what fp64 arithmetic gives for them)
"""
import os
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from tetraear.signal.processor import SignalProcessor  # noqa: E402  (the reference)
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("golden_cases", os.path.join(REPO, "tests", "golden_cases.py"))   # (the reference has a `tests` package of its own)
_gc = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_gc)
DTYPE_CASES, dtype_case_input = _gc.DTYPE_CASES, _gc.dtype_case_input   # seeded inputs, shared with the tests


def main():
    out = {"meta": np.array([f"numpy {np.__version__}", f"scipy {scipy.__version__}",
                             "syrex1013/TetraEar v2.2 tetraear/signal/processor.py"])}
    cases = [(name, fs, foff, dtype_case_input(name)) for name, (fs, foff) in DTYPE_CASES.items()]
    for name, fs, foff, x in cases:
        p = SignalProcessor(fs)
        hard = p.process(x.copy(), foff)
        p64 = SignalProcessor(fs)
        hard64 = p64.process(x.astype(np.complex128), foff)
        out[name + "__hard"] = hard
        out[name + "__soft"] = np.asarray(p.symbols)
        out[name + "__hard64"] = hard64
        out[name + "__soft64"] = np.asarray(p64.symbols)
        out[name + "__par"] = np.array([fs, foff])
        s, s64 = np.asarray(p.symbols), np.asarray(p64.symbols)
        rel = np.max(np.abs(s - s64)) / np.max(np.abs(s64)) if len(s) == len(s64) and len(s) else float("nan")
        print(f"{name:28s} {x.dtype!s:10s} symbols dtype {s.dtype!s:10s} n {len(hard):5d} hard == fp64-input hard: {np.array_equal(hard, hard64)}"
              f"  differing {int(np.sum(hard != hard64)) if len(hard) == len(hard64) else -1}  soft vs fp64-input {rel:.2e}")
    np.savez_compressed(os.path.join(HERE, "dtypes.npz"), **out)


if __name__ == "__main__":
    main()
