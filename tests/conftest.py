import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "midvision-probe_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no libmvp_hip.so (built artefacts are git-ignored): build it once (hipcc cross-compiles gfx950
    without a GPU) so that the ABI tests can load it.  An existing library is left alone."""
    so = os.path.join(PKG, "csrc", "libmvp_hip.so")
    if not os.path.exists(so):
        import subprocess

        subprocess.run(["make", "-C", os.path.join(PKG, "csrc")], check=False, stdout=subprocess.DEVNULL)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_l2(a, b):
    """Relative L2 error ||a-b|| / ||b|| in float64 (the "rel" of north_star's 1e-3)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def max_rel(a, b):
    """max |a-b| / max |b|."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
