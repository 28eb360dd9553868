"""pytest configuration: markers, import paths, shared fixtures."""

import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: needs a real MI355X (run with -m gpu through gpurun)')


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture
def golden():
    return load_golden


def rel_l2(a, b):
    """||a - b|| / ||b|| (relative l2 error against reference ``b``)."""
    a = np.asarray(a, dtype=np.complex128 if np.iscomplexobj(a) else np.float64)
    b = np.asarray(b, dtype=a.dtype)
    nb = np.linalg.norm(b.ravel())
    if nb == 0.0:
        return float(np.linalg.norm(a.ravel()))
    return float(np.linalg.norm((a - b).ravel()) / nb)
