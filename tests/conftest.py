"""pytest configuration: markers, import paths, shared helpers."""
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


def pytest_collection_modifyitems(config, items):
    # every test that is not explicitly marked gpu is a CPU test; nothing to do.
    # gpu tests are skipped with a clear reason when no device is visible and
    # the user did not ask for them with -m gpu.
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def bluestein_route():
    """composite lengths go back to the round-2 routing (Bluestein / direct) for the duration of a test: the mixed-radix kernel
    (csrc/fft_mixed.h) owns them by default"""
    from prysm_amd import _lib
    lib = _lib.load()
    lib.pm_set_tuning(b'mix', 0)
    try:
        yield lib
    finally:
        lib.pm_set_tuning(b'mix', 1)


@pytest.fixture
def radix_r_route():
    """the lengths 3 / 5 / 7 x 2^k (up to 8192) go back to the radix-R step around engine transforms (csrc/bigfft.hip) for the duration
    of a test: the mixed-radix kernel owns them by default"""
    from prysm_amd import _lib
    lib = _lib.load()
    lib.pm_set_tuning(b'mix', 2)
    try:
        yield lib
    finally:
        lib.pm_set_tuning(b'mix', 1)


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load


def rel_l2(a, b):
    """||a-b||_2 / ||b||_2 (b is the truth)."""
    a = np.asarray(a)
    b = np.asarray(b)
    nb = np.linalg.norm(b.ravel())
    if nb == 0:
        return float(np.linalg.norm(a.ravel()))
    return float(np.linalg.norm((a - b).ravel()) / nb)


def rel_max(a, b):
    """max|a-b| / max|b|."""
    a = np.asarray(a)
    b = np.asarray(b)
    mb = np.max(np.abs(b))
    if mb == 0:
        return float(np.max(np.abs(a)))
    return float(np.max(np.abs(a - b)) / mb)
