import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Built artefacts are git-ignored: on a fresh checkout compile them once (hipcc cross-compiles
    # without a GPU; the oracle and the CPU emulation harness need only gcc/g++).
    import shutil
    import subprocess
    lib = os.path.join(REPO, "tetraear_amd", "libtetrahip.so")
    if not os.path.exists(lib) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.run(["make", "-C", os.path.join(REPO, "tetraear_amd", "csrc"), "-s"], check=False)
    if not os.path.exists(os.path.join(REPO, "oracle", "liboracle.so")):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "-s"], check=False)


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def gold_inputs():
    return np.load(os.path.join(GOLDEN, "inputs.npz"))


@pytest.fixture(scope="session")
def gold_process():
    return np.load(os.path.join(GOLDEN, "process.npz"))


@pytest.fixture(scope="session")
def gold_stages():
    return np.load(os.path.join(GOLDEN, "stages.npz"))


@pytest.fixture(scope="session")
def gold_design():
    return np.load(os.path.join(GOLDEN, "design.npz"))
