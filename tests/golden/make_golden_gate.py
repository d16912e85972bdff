#!/usr/bin/env python3
"""Golden vectors for the spectrum / AFC / signal gate, made by RUNNING the
reference's own statements: tetraear/ui/modern.py:1919-2021 (spectrum + detection block of
`CaptureThread.run`) and the `afc_offset = ...` assignment at :2032.

`tetraear.ui.modern` cannot be imported here (PyQt6 is absent) and the block is inline code of a
thread loop, so this script reads the reference file where it lies, takes exactly those statements
out of its AST, wraps them in a function and executes them unchanged with a stand-in `self` (the
Qt signal `.emit()` calls become no-ops).  Nothing is copied into the repository: only inputs
(cu8 bytes) and the values the block computed are stored.

    python tests/golden/make_golden_gate.py   ->  tests/golden/gate.npz
"""
import ast
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tetraear/ui/modern.py"
KEYS = ("peak_freq_offset", "signal_power", "peak_power", "noise_floor", "snr")


def load_reference_block():
    tree = ast.parse(open(REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CaptureThread")
    run = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "run")

    def find_body(node):
        """the statement list that holds `n_fft = 2048`"""
        for field in ("body", "orelse", "finalbody", "handlers"):
            seq = getattr(node, field, None)
            if not isinstance(seq, list):
                continue
            for i, st in enumerate(seq):
                if (isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name) and st.targets[0].id == "n_fft"):
                    return seq, i
                if isinstance(st, ast.AST):
                    r = find_body(st)
                    if r:
                        return r
        return None

    seq, i0 = find_body(run)
    i1 = next(i for i in range(i0, len(seq)) if isinstance(seq[i], ast.If) and isinstance(seq[i].test, ast.Name)
              and seq[i].test.id == "signal_present")
    block = seq[i0:i1]
    afc = next(n for n in ast.walk(seq[i1]) if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Name)
               and n.targets[0].id == "afc_offset")
    body = list(block) + [ast.If(test=ast.Name(id="signal_present", ctx=ast.Load()), body=[afc], orelse=[]),
                          ast.Return(value=ast.Call(func=ast.Name(id="locals", ctx=ast.Load()), args=[], keywords=[]))]
    args = ast.arguments(posonlyargs=[], args=[ast.arg(arg=a) for a in
                                               ("self", "samples", "last_spectrum_update", "spectrum_update_interval",
                                                "last_status_update", "status_update_interval")],
                         kwonlyargs=[], kw_defaults=[], defaults=[])
    fn = ast.FunctionDef(name="ref_gate", args=args, body=body, decorator_list=[])
    mod = ast.Module(body=[fn], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"np": np, "time": time}
    exec(compile(mod, REF, "exec"), ns)
    return ns["ref_gate"]


def run_block(fn, samples, sample_rate):
    sig = types.SimpleNamespace(emit=lambda *a, **k: None)
    me = types.SimpleNamespace(sample_rate=sample_rate, frequency=392.5e6, spectrum_update=sig, signal_detected=sig,
                               signal_lost=sig, last_signal_time=0.0)
    loc = fn(me, samples, 0.0, 1e9, 0.0, 1e9)
    if "signal_power" not in loc:          # len(samples) < n_fft: the block did nothing
        return np.zeros(7)
    return np.array([float(loc[k]) for k in KEYS] + [float(bool(loc["signal_present"])), float(loc.get("afc_offset", 0))])


def cases():
    """(name, sample_rate, cu8 bytes): pi/4-DQPSK at several offsets and SNRs, noise, a tone, short input."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tetraear_amd import synth
    out = []
    specs = [(0.0, 30.0), (3000.0, 30.0), (-2500.0, 5.0), (9000.0, 30.0), (-11000.0, 25.0), (500.0, -5.0),
             (12400.0, 35.0), (-600.0, 12.0)]
    for fs in (2.4e6, 1.8e6, 2.048e6):
        for r, (co, snr) in enumerate(specs):
            u8 = synth.dqpsk_cu8(4096, fs, seed=60 + r, carrier_offset=co, esn0_db=snr)[0]
            out.append((f"dqpsk_{int(fs)}_{r}", fs, u8[:2 * 2304]))
    out.append(("noise", 2.4e6, synth.noise_cu8(2304, 9)))
    t = np.arange(2304)
    for k, f in enumerate((0.0, 1171.875, -4700.0, 30000.0)):
        x = 0.6 * np.exp(2j * np.pi * f * t / 2.4e6)
        out.append((f"tone_{k}", 2.4e6, synth.quantise_cu8(x, scale=1.0)))
    out.append(("short", 2.4e6, synth.noise_cu8(1000, 3)))
    return out


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tetraear_amd import synth
    fn = load_reference_block()
    out = {}
    names = []
    for name, fs, u8 in cases():
        x = synth.cu8_to_c128(u8)
        names.append(name)
        out[f"in_{name}"] = u8
        out[f"fs_{name}"] = np.array([fs])
        out[f"out_{name}"] = run_block(fn, x, fs)
    out["names"] = np.array(names)
    out["keys"] = np.array(list(KEYS) + ["signal_present", "afc_offset"])
    np.savez_compressed(os.path.join(HERE, "gate.npz"), **out)
    strong = sum(int(out[f"out_{n}"][5]) for n in names)
    print(f"{len(names)} cases, {strong} pass the gate")


if __name__ == "__main__":
    main()
