"""CPU: the C-ABI library builds, loads and exports every symbol include/prysm_amd.h declares
(no compute calls -- there is no GPU here), and compute entry points fail loudly without a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'prysm_amd.h')


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(pm_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def lib():
    from prysm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for s in ('pm_fft2', 'pm_fft2_workspace', 'pm_fft1', 'pm_cmul', 'pm_abs2', 'pm_pupil_synth', 'pm_mdft_basis',
              'pm_cgemm', 'pm_sample_map', 'pm_as_tf_vectors', 'pm_embed', 'pm_last_error', 'pm_version'):
        assert s in syms


def test_library_exports_every_declared_symbol(lib):
    from prysm_amd import _lib
    for s in declared_symbols():
        assert hasattr(lib, s), f'{s} declared in prysm_amd.h but not exported'
        assert s in _lib.SIGNATURES, f'{s} has no ctypes signature in prysm_amd/_lib.py'
    assert lib.pm_version() == 106


def test_argument_errors_are_reported_without_a_gpu(lib):
    from prysm_amd import _lib as L
    d = L.pm_fft2_desc()
    d.dtype = 7
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 0
    rc = lib.pm_fft2(ctypes.byref(d), None, None, None, 0, None)
    assert rc == L.PM_ERR_ARG
    assert b'dtype' in lib.pm_last_error()
    with pytest.raises(ValueError):
        L.check(rc)
    # workspace query is pure host arithmetic: tiled intermediate of a 4096^2 complex64 transform
    d = L.pm_fft2_desc()
    d.dtype, d.direction = L.PM_C64, -1
    ax = L.pm_axis(4096, 4096, 0, 2048)
    d.in_y = d.in_x = d.out_y = d.out_x = ax
    d.in_ld = d.out_ld = 4096
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 4096 * 4096 * 8
    # Q = 2 pad: only the 2048 stored rows are transformed in the row pass
    d.in_y = d.in_x = L.pm_axis(4096, 2048, 1024, 2048)
    d.in_ld = 2048
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 2048 * 4096 * 8
    # batches: the workspace holds one chunk of fields (<= 128 MiB of intermediates per launch pair), not the batch
    d.in_y = d.in_x = d.out_y = d.out_x = L.pm_axis(1024, 1024, 0, 512)
    d.in_ld = d.out_ld = 1024
    d.batch, d.in_bstride, d.out_bstride = 4, 1024 * 1024, 1024 * 1024
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 4 * 1024 * 1024 * 8
    d.batch = 100
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 16 * 1024 * 1024 * 8
    d.out_bstride = 1000          # outputs of consecutive fields would overlap
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 0
    assert lib.pm_fft2(ctypes.byref(d), ctypes.c_void_p(16), ctypes.c_void_p(16), None, 0, None) == L.PM_ERR_ARG
    assert b'out_bstride' in lib.pm_last_error()
    d.batch = d.in_bstride = d.out_bstride = 0
    d.in_ld = 2048
    d.in_y = d.in_x = L.pm_axis(4096, 2048, 1024, 2048)
    d.out_y = d.out_x = ax
    d.out_ld = 4096
    # lengths beyond both the engine and the direct DFT are refused loudly
    d.in_y = d.out_y = L.pm_axis(40000, 40000, 0, 0)
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 0
    rc = lib.pm_fft2(ctypes.byref(d), ctypes.c_void_p(16), ctypes.c_void_p(16), None, 0, None)
    assert rc == L.PM_ERR_UNSUPPORTED
    with pytest.raises(NotImplementedError):
        L.check(rc)


def test_no_cpu_fallback():
    """The product path must fail loudly, not compute on the CPU, when no GPU is visible."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible')
    from prysm_amd import propagation as P
    with pytest.raises(RuntimeError):
        P.focus(np.ones((8, 8), dtype=np.complex64), 1)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'prysm_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in re.sub(r'""".*?"""', '', src, flags=re.S), f
