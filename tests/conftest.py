"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI symbol export,
gloo data-parallel tests.  `-m gpu` runs on a real MI355X and calls the HIP path through the C-ABI.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def rel_linf(a, b):
    """max|a-b| / max(1e-30, max|b|): the normalised max error every parity test reports."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f'{a.shape} != {b.shape}'
    if b.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(1e-30, float(np.max(np.abs(b)))))


@pytest.fixture(scope='session')
def golden():
    return load_golden
