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


# ---- fixtures of the GPU parity tests (tests/test_gpu_*.py)
@pytest.fixture(scope='module')
def pa():
    import torch
    import prysm_amd
    from prysm_amd import _lib
    _lib.load()   # fails loudly when the HIP library is missing
    assert torch.cuda.is_available()
    return prysm_amd


@pytest.fixture(scope='module')
def config5():
    """4096^2 circular amplitude, 500 nm of W040, 64 wavelengths in [0.5, 0.7] um, uniform weights; fp32
    maps.  The oracle sums are computed once (scipy.fft on all host cores -- the checker, not the thing measured)."""
    from scipy import fft as sfft
    from oracle import prysm_oracle as O
    n = 4096
    x, y = O.make_xy_grid(n, diameter=10)
    r, _ = O.cart_to_polar(x, y)
    amp = O.circle(5, r).astype(np.float32)
    opd = O.hopkins_w040(r / 5, 500.0).astype(np.float32)
    dx = float(x[0, 1] - x[0, 0])
    del x, y, r
    wvls = np.linspace(0.5, 0.7, 64)
    wts = np.ones(64)
    want_f = np.zeros((n, n))
    want_m = np.zeros((512, 512))
    opd64 = opd.astype(np.float64)
    with sfft.set_workers(os.cpu_count() or 1):
        for w in wvls:
            P = O.from_amp_and_phase(amp.astype(np.float64), opd64, float(w))
            want_f += O.intensity(O.focus(P, 1))
            want_m += O.intensity(O.prepare_executor(dx, P.shape, 0.55 * 10 / 4, (512, 512), float(w), 100.0)(P))
    return dict(amp=amp, opd=opd, dx=dx, wvls=wvls, wts=wts, want_f=want_f, want_m=want_m)
