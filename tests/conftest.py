import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_cache = {}


def load_golden(name):
    if name not in _cache:
        with np.load(os.path.join(GOLDEN_DIR, name)) as z:
            _cache[name] = {k: z[k] for k in z.files}
    return _cache[name]


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_airm_self.npz")


@pytest.fixture(scope="session")
def g1x():
    return load_golden("g1x_airm_cross.npz")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_fisher_rao.npz")


@pytest.fixture(scope="session")
def g3():
    return load_golden("g3_closure.npz")


@pytest.fixture(scope="session")
def g4():
    return load_golden("g4_fit.npz")


@pytest.fixture(scope="session")
def g5():
    return load_golden("g5_quirks.npz")


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b.ravel())
    return np.linalg.norm((a - b).ravel()) / (den if den > 0 else 1.0)
