#!/usr/bin/env python3
"""Golden vectors for NON-FINITE input samples (run in the build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_nonfinite.py

The reference has no guard: `scipy.signal.decimate` (processor.py:254, sosfiltfilt) and `filtfilt` (:79) carry one NaN / Inf
over the whole chunk (the forward pass to the end, the backward pass -- started from the forward pass's last value -- back to
the start), every comparison of the slicer is then false (:152-161 -> symbol 3) and no timing phase's power beats
`max_power = -1` (:196-210 -> phase 0); where no filter runs (<= 15 samples at an undecimated rate) the NaN stays put.
The reference is imported read-only; inputs are seeded (tests/golden_cases.py), outputs stored.  Writes
tests/golden/nonfinite.npz:
  <case>__hard / __soft        process(): decisions and the `symbols` attribute
  st_<method>_<input>          per-method outputs (filter_signal, decimate, frequency_shift, extract_symbols,
                               demodulate_dqpsk, resample) on the inputs of golden_cases.nonfinite_stage_inputs()
"""
import logging
import os
import sys
import warnings

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from tetraear.signal.processor import SignalProcessor  # noqa: E402  (the reference)
from scipy import signal  # noqa: E402
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("golden_cases", os.path.join(REPO, "tests", "golden_cases.py"))   # (the reference has a `tests` package of its own)
_gc = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_gc)


def main():
    warnings.simplefilter("ignore")          # (numpy's invalid-value warnings: the point of these cases)
    logging.disable(logging.CRITICAL)
    out = {"meta": np.array([f"numpy {np.__version__}", f"scipy {scipy.__version__}",
                             "syrex1013/TetraEar v2.2 tetraear/signal/processor.py"])}
    for name, (fs, foff, n, seed, dt, inject) in _gc.NONFINITE_CASES.items():
        x = _gc.nonfinite_case_input(name)
        p = SignalProcessor(fs)
        hard = p.process(x.copy(), foff)
        soft = np.asarray(p.symbols)
        out[name + "__hard"], out[name + "__soft"] = hard, soft
        nn = int(np.sum(~np.isfinite(soft.real) | ~np.isfinite(np.imag(soft))))
        print(f"{name:36s} {x.dtype!s:10s} hard {len(hard):5d} values {np.unique(hard)} soft {soft.dtype!s:10s} {len(soft):5d} non-finite {nn}")
    st = _gc.nonfinite_stage_inputs()
    p = SignalProcessor(2.4e6)
    for tag in ("nan", "inf"):
        x = st["x4000_" + tag]
        out[f"st_filter_{tag}"] = p.filter_signal(x)
        out[f"st_filter_240k_{tag}"] = p.filter_signal(x, 25000, 240000.0)
        out[f"st_filter_short15_{tag}"] = p.filter_signal(x[1225:1240])          # <= 15 samples: returned unfiltered
        out[f"st_shift_{tag}"] = p.frequency_shift(x, 1000)
        out[f"st_decimate_q10_{tag}"] = signal.decimate(x, 10)
        out[f"st_decimate_q7_{tag}"] = signal.decimate(x, 7)
        out[f"st_resample_{tag}"] = p.resample(x[:2000], 1.2e6)
        out[f"st_extract_{tag}"] = p.extract_symbols(x, 240000.0)
        out[f"st_demod_{tag}"] = p.demodulate_dqpsk(x)
    out["st_extract_nan_some_phases"] = p.extract_symbols(st["x4000_nan_some_phases"], 240000.0)
    out["st_extract_nan_some_phases_300k"] = p.extract_symbols(st["x4000_nan_some_phases"], 300000.0)
    for k in ("sym200_nan", "sym200_inf", "sym200_nan_ends"):
        out["st_demod_" + k] = p.demodulate_dqpsk(st[k])
    for k in sorted(out):
        if k.startswith("st_"):
            a = out[k]
            print(f"{k:36s} {a.dtype!s:10s} {len(a):5d} non-finite {int(np.sum(~np.isfinite(a.astype(complex))))}"
                  + (f" values {np.bincount(a, minlength=4)}" if a.dtype == np.uint8 else ""))
    np.savez_compressed(os.path.join(HERE, "nonfinite.npz"), **out)


if __name__ == "__main__":
    main()
