import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gan-aug-pfa_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Lazily loaded committed reference outputs (tests/golden/*.npz, made by tests/make_golden.py)."""
    class _G:
        def __init__(self):
            self._z = {}

        def __call__(self, fname):
            if fname not in self._z:
                self._z[fname] = np.load(os.path.join(GOLDEN, fname + ".npz"))
            return self._z[fname]
    return _G()


def assert_close(got, want, atol, rtol, what=""):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: {bad.sum()}/{bad.size} outside atol={atol} rtol={rtol}; "
                             f"worst at {i}: got {got[i]} want {want[i]} (max abs err {err.max():.3e})")
