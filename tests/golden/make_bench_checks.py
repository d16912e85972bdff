#!/usr/bin/env python3
"""Writes tests/golden/bench_checks.npz: what bench.py's north-star legs compare their output with.

Runs on the CPU (no GPU, no reference needed): the legs' synthetic workloads (bench.py's own generators) go through the
fp64 definitions of oracle/ -- tetra_np (receiver), pfb_np (channeliser) -- and the results are stored as data:

  tetra leg     per distinct carrier: sha256 of (symbol count, hard decisions) of the definition with the plan's 16-bit
                coefficients, and the UNQUANTISED definition's soft symbols at 256 seeded probe indices
  pfb leg       the definition's output of stream 0 on four probe channels, every 16th output time
  wideband leg  sha256 of (symbol count, hard decisions) of the nine occupied channels through the definition chain
                channeliser (fp64) -> cast to cf32 -> receiver

bench.py never imports oracle/ for these legs: it loads this file.   usage: python tests/golden/make_bench_checks.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import pfb_np, tetra_np  # noqa: E402
from tetraear_amd import synth  # noqa: E402


def row_digest(n_sym, hard):
    h = hashlib.sha256()
    h.update(np.int32(n_sym).tobytes())
    h.update(np.ascontiguousarray(hard, dtype=np.uint8).tobytes())
    return h.hexdigest()


def main():
    out = {}
    # ---- tetra leg
    rng = np.random.default_rng(20260930)
    digests, idxs, refs = [], [], []
    for r, x in enumerate(bench.tetra_rows()):
        xd = x.astype(np.complex128)
        hard, _, info = tetra_np.demod(xd, bench.TETRA_FS)
        hard_e, _, info_e = tetra_np.demod(xd, bench.TETRA_FS, exact_taps=True)
        assert info["n_sym"] == info_e["n_sym"] and np.array_equal(hard, hard_e), r
        digests.append(row_digest(info["n_sym"], hard))
        idx = np.sort(rng.choice(info["n_sym"], size=256, replace=False)).astype(np.int64)
        idxs.append(idx)
        refs.append(info_e["sym"][idx])
        print(f"tetra row {r}: {info['n_sym']} symbols, sha256 {digests[-1][:16]}")
    # ---- the same carriers through the Gardner receiver of the definition (TDM_MODE_TETRA_GARDNER's side leg): its decisions,
    # as arrays -- the device runs the loop in fp32, so the leg compares decision by decision (<= 1e-3 may differ, the
    # symbol count by one at the chunk's end) instead of by digest
    g_n, g_hard = [], []
    for r, x in enumerate(bench.tetra_rows()):
        hard, _, info = tetra_np.demod_gardner(x.astype(np.complex128), bench.TETRA_FS)
        g_n.append(len(info["t"]))
        g_hard.append(hard)
        print(f"tetra row {r} (Gardner): {len(info['t'])} symbols")
    width = max(len(h) for h in g_hard)
    out["gardner_n_sym"] = np.array(g_n, dtype=np.int32)
    out["gardner_hard"] = np.stack([np.pad(h, (0, width - len(h))) for h in g_hard]).astype(np.uint8)
    out["tetra_digests"] = np.array(digests)
    out["tetra_soft_idx"] = np.stack(idxs)
    out["tetra_soft_ref"] = np.stack(refs)
    # ---- pfb leg
    chans = np.array([0, 57, 250, 399], dtype=np.int64)
    stride = 16
    xd = synth.cu8_to_c128(bench.pfb_stream())
    ref = pfb_np.channelise(xd, bench.PFB_M, bench.PFB_D, channels=[int(k) for k in chans])
    out["pfb_channels"] = chans
    out["pfb_time_stride"] = np.int64(stride)
    out["pfb_scale"] = np.float64(np.max(np.abs(ref)))
    out["pfb_ref"] = ref[:, ::stride]
    print(f"pfb: {ref[:, ::stride].size} probes on channels {chans.tolist()}, scale {float(out['pfb_scale']):.4f}")
    # ---- wideband leg
    u8, dibs = bench.wideband_stream()
    ks = [int(k) for k in bench.WIDEBAND_CHANNELS]
    y = pfb_np.channelise(synth.cu8_to_c128(u8), bench.PFB_M, bench.PFB_D, channels=ks)
    h = hashlib.sha256()
    for i, k in enumerate(ks):
        hard, _, info = tetra_np.demod(y[i].astype(np.complex64).astype(np.complex128), bench.PFB_FS / bench.PFB_D)
        # the definition must give back what was sent (sanity of the fixture itself)
        sent = dibs[k]
        best = min(np.mean(hard[8:-8] != sent[lag + 8: lag + len(hard) - 8]) for lag in range(0, 24))
        assert best == 0.0, (k, best)
        h.update(np.int32(info["n_sym"]).tobytes())
        h.update(np.ascontiguousarray(hard).tobytes())
        print(f"wideband channel {k}: {info['n_sym']} symbols, error-free against the transmitted dibits")
    out["wideband_digest"] = np.array(h.hexdigest())
    np.savez_compressed(os.path.join(HERE, "bench_checks.npz"), **out)
    print("wrote", os.path.join(HERE, "bench_checks.npz"))


if __name__ == "__main__":
    main()
