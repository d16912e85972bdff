#!/usr/bin/env python3
"""Golden vectors for burst sync, made by RUNNING the reference's own
`TetraDecoder.symbols_to_bits` / `TetraDecoder.find_sync` (tetraear/core/decoder.py:140-295).

`tetraear.core.decoder` cannot be imported in this container (its module imports `bitstring`,
which is not installed), and the two methods do not use it; so this script reads the reference
file where it lies, takes those two function definitions out of its AST and executes them
unchanged.  Nothing is copied into the repository: only inputs (as seeds) and outputs are stored.

    python tests/golden/make_golden_sync.py   ->  tests/golden/sync.npz
"""
import ast
import logging
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/tetraear/core/decoder.py"


def load_reference_methods():
    tree = ast.parse(open(REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TetraDecoder")
    keep = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("symbols_to_bits", "find_sync")]
    assert len(keep) == 2
    mod = ast.Module(body=[ast.ClassDef(name="RefSync", bases=[], keywords=[], body=keep, decorator_list=[])],
                     type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"np": np, "logger": logging.getLogger("ref")}
    exec(compile(mod, REF, "exec"), ns)
    return ns["RefSync"]()


TS1 = [1, 1, 0, 1, 0, 0, 0, 0, 1, 1, 1, 0, 1, 0, 0, 1, 1, 1, 0, 1, 0, 0]
TS2 = [0, 1, 1, 1, 1, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1, 1, 0, 1, 1, 1, 0, 0]


def make_case(seed):
    """Random dibit symbols with training sequences planted at random bit positions, some corrupted."""
    rng = np.random.default_rng(seed)
    n_sym = int(rng.integers(40, 2200))
    sym = rng.integers(0, 4, size=n_sym, dtype=np.uint8)
    bits = np.empty(2 * n_sym, dtype=np.uint8)
    bits[0::2] = sym >> 1
    bits[1::2] = sym & 1
    for _ in range(int(rng.integers(0, 7))):
        if len(bits) < 30:
            break
        pos = int(rng.integers(0, len(bits) - 22))
        pat = np.array(TS1 if rng.integers(0, 2) == 0 else TS2, dtype=np.uint8)
        nerr = int(rng.integers(0, 6))
        flip = rng.choice(22, size=nerr, replace=False)
        pat = pat.copy()
        pat[flip] ^= 1
        bits[pos:pos + 22] = pat
    sym = (bits[0::2] << 1) | bits[1::2]
    return sym.astype(np.uint8)


def main():
    ref = load_reference_methods()
    out = {}
    thresholds = [0.90, 0.85, 0.80, 0.77, 0.95]
    seeds = list(range(5000, 5060))
    out["seeds"] = np.array(seeds)
    out["thresholds"] = np.array(thresholds)
    for seed in seeds:
        sym = make_case(seed)
        bits, mapped = ref.symbols_to_bits(sym)
        assert np.array_equal(mapped, sym)
        out[f"bits_sha_{seed}"] = np.array([int(np.sum(bits * (np.arange(len(bits)) % 251 + 1)))])
        for ti, thr in enumerate(thresholds):
            pos, mc = ref.find_sync(bits, threshold=thr, return_max_corr=True)
            out[f"pos_{seed}_{ti}"] = np.array(pos, dtype=np.int32)
            out[f"mc_{seed}_{ti}"] = np.array([mc])
    # the reference's own unit-test inputs (tests/unit/test_tetra_decoder.py:50-66)
    bits = np.array([0] * 100)
    bits[20:42] = TS1
    pos, mc = ref.find_sync(bits, threshold=0.8, return_max_corr=True)
    out["unit_pos"] = np.array(pos, dtype=np.int32)
    out["unit_mc"] = np.array([mc])
    pos, mc = ref.find_sync(np.array([0] * 10), return_max_corr=True)
    out["short_pos"] = np.array(pos, dtype=np.int32)
    out["short_mc"] = np.array([mc])
    np.savez_compressed(os.path.join(HERE, "sync.npz"), **out)
    n = sum(len(out[f"pos_{s}_{t}"]) for s in seeds for t in range(len(thresholds)))
    print("cases", len(seeds) * len(thresholds), "total positions", n, os.path.getsize(os.path.join(HERE, "sync.npz")))


if __name__ == "__main__":
    main()
