import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def flow2d():
    """The product package (ctypes plumbing over the C-ABI).  The native libraries are built in-tree by
    __graft_entry__.build(); a fresh checkout builds them here once (make is a no-op when up to date)."""
    mod = importlib.import_module("cuda-flow2d_amd")
    mod.build()
    return mod


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle -- the checker, never the thing under test."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture()
def ctx(flow2d):
    if flow2d.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu-marked tests need the MI355X box (there is no CPU fallback)")
    c = flow2d.Context(0)
    yield c
    c.close()


def level_fields(oracle, w, h, seed=0, flow_scale=1.0):
    """Seeded level-sized inputs of the solver kernels (frames, flow, increments)."""
    rng = np.random.default_rng(seed)
    f0, f1 = oracle.synthetic_pair(w, h, 1.5, -0.75)
    f1 = (f1 + rng.uniform(-1, 1, f1.shape)).astype(np.float32)
    u = (rng.normal(0, 1, (h, w)) * flow_scale).astype(np.float32)
    v = (rng.normal(0, 1, (h, w)) * flow_scale).astype(np.float32)
    du = rng.normal(0, 0.3, (h, w)).astype(np.float32)
    dv = rng.normal(0, 0.3, (h, w)).astype(np.float32)
    return f0, f1, u, v, du, dv


def in_container(a, cw, ch, fill=0.0):
    """Place a level-sized image in the top-left corner of a (ch, cw) host container."""
    out = np.full((ch, cw), fill, np.float32)
    out[: a.shape[0], : a.shape[1]] = a
    return out
