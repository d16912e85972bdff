"""The FIR arithmetic of the TETRA-mode definitions against scipy.signal (fixtures: tests/golden/scipy_fir.npz, made by
tests/golden/make_golden_scipy_fir.py with scipy.signal.firwin / upfirdn / lfilter).  These stages have no counterpart
in the reference ("parity unpinned"); scipy is the reference's own arithmetic dependency, so the definitions -- and
through them the device -- are at least not checked against themselves only."""
import os

import numpy as np
import pytest

from oracle import pfb_np, tetra_np
from tetraear_amd import synth

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scipy_fir.npz"), allow_pickle=False)
N_PFB = sum(1 for k in G.files if k.endswith("_MD"))
N_RRC = sum(1 for k in G.files if k.endswith("_fs"))


@pytest.mark.parametrize("ci", range(N_PFB))
def test_channeliser_definition_equals_scipy_upfirdn(ci):
    M, D = (int(v) for v in G[f"pfb{ci}_MD"])
    assert np.max(np.abs(pfb_np.prototype(M, D) - G[f"pfb{ci}_h"])) < 1e-15      # Kaiser-windowed sinc == scipy.signal.firwin
    x = synth.cu8_to_c128(G[f"pfb{ci}_u8"])
    probe = [int(k) for k in G[f"pfb{ci}_probe"]]
    y = pfb_np.channelise(x, M, D, channels=probe)
    ref = G[f"pfb{ci}_y"]
    assert y.shape == ref.shape
    assert np.max(np.abs(y - ref)) < 1e-12 * np.max(np.abs(ref))


@pytest.mark.parametrize("ci", range(N_RRC))
def test_matched_filter_definition_equals_scipy_lfilter(ci):
    fs = float(G[f"rrc{ci}_fs"])
    x = G[f"rrc{ci}_x"].astype(np.complex128)
    h = tetra_np.rrc_taps(fs / 18000.0, exact=True)
    y = tetra_np.matched_filter(x, h)
    ref = G[f"rrc{ci}_y"]
    assert np.max(np.abs(y - ref)) < 1e-12 * np.max(np.abs(ref))
    _, _, info = tetra_np.demod(x, fs, exact_taps=True)
    assert info["n_sym"] == len(G[f"rrc{ci}_sym"])
    assert np.max(np.abs(info["sym"] - G[f"rrc{ci}_sym"])) < 1e-9 * np.max(np.abs(G[f"rrc{ci}_sym"]))


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(N_PFB))
def test_gpu_channeliser_vs_scipy_upfirdn(ci):
    from tetraear_amd.channeliser import channelise
    M, D = (int(v) for v in G[f"pfb{ci}_MD"])
    y = channelise(G[f"pfb{ci}_u8"], "cu8", M, D)
    ref = G[f"pfb{ci}_y"]
    scale = np.max(np.abs(ref))
    for i, k in enumerate(G[f"pfb{ci}_probe"]):
        assert np.max(np.abs(y[int(k)] - ref[i])) < 2e-5 * scale, (M, D, int(k))


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(N_RRC))
def test_gpu_receiver_soft_symbols_vs_scipy_filtered_definition(ci):
    """soft symbols of the device against the definition run on scipy.signal.lfilter's matched-filter output
    (unquantised taps): 1e-5 of the largest symbol"""
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    fs = float(G[f"rrc{ci}_fs"])
    x = G[f"rrc{ci}_x"]
    bd = BatchDemodulator(fs, len(x), 1, "cf32", mode=MODE_TETRA)
    hards, softs, timing, margin = bd.process(x)
    bd.close()
    ref = G[f"rrc{ci}_sym"]
    assert len(softs[0]) == len(ref)
    assert np.max(np.abs(softs[0] - ref)) < 1e-5 * np.max(np.abs(ref))
