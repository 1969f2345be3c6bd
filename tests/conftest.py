import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The HIP library travels with the repo snapshot; if it is missing (fresh checkout) build it -- hipcc cross-compiles
    gfx950 without a GPU."""
    lib = os.path.join(ROOT, "crt1d_amd", "libcrt1d_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__

        __graft_entry__.build()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"{name}.npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import crt_oracle

    return crt_oracle


def rel_profile_err(got, ref):
    """max over everything of |got - ref| / (max |ref| over the level axis of that (column, band))."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    ax = got.ndim - 2  # level axis of (..., nz, nb)
    scale = np.abs(ref).max(axis=ax, keepdims=True)
    scale = np.where(scale == 0, 1.0, scale)
    return float(np.max(np.abs(got - ref) / scale))


def rel_elem_err(got, ref, floor_frac=1e-9):
    """ELEMENTWISE relative error, max over everything of |got - ref| / max(|ref|, floor), the form north_star words
    ("within 1e-6 relative").  floor = floor_frac x (max |ref| over the level axis of that (column, band)): an element more than
    nine orders of magnitude below its own profile's maximum is compared against that floor instead of against itself (the
    upward flux of bl is identically zero; tails of deep canopies underflow towards it)."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    ax = got.ndim - 2
    scale = np.abs(ref).max(axis=ax, keepdims=True)
    den = np.maximum(np.abs(ref), floor_frac * np.where(scale == 0, 1.0, scale))
    return float(np.max(np.abs(got - ref) / den))
