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


def test_workspace_queries_of_the_mixed_radix_path(lib):
    """composite lengths (primes <= 13, up to 8192) run in LDS on their own factors: no scratch beyond the natural intermediate, none for
    the 1-D entry; lengths with a larger prime keep Bluestein's workspace"""
    from prysm_amd import _lib as L
    d = L.pm_fft2_desc()
    d.dtype, d.direction = L.PM_C64, -1
    d.in_y = d.in_x = d.out_y = d.out_x = L.pm_axis(1000, 1000, 0, 500)
    d.in_ld = d.out_ld = 1000
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 1000 * 1008 * 8     # rows of the intermediate padded to whole 128 B lines
    d.in_y = d.out_y = L.pm_axis(3000, 3000, 0, 0)
    d.dtype = L.PM_C128
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 3000 * 1000 * 16
    d.in_y = d.out_y = L.pm_axis(6000, 6000, 0, 0)      # a length in (4096, 8192] no longer drags the other axis into a convolution
    d.in_x = d.out_x = L.pm_axis(2048, 2048, 0, 0)
    d.in_ld = d.out_ld = 2048
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 6000 * 2048 * 16
    for n in (1000, 3000, 2592, 1001, 7000):
        assert lib.pm_fft1_workspace(L.PM_C64, 1, 8, n) == 0 and lib.pm_fft1_workspace(L.PM_C128, 0, 8, n) == 0
    assert lib.pm_fft1_workspace(L.PM_C64, 1, 8, 1009) > 0          # prime: Bluestein
    assert lib.pm_fft1_workspace(L.PM_C64, 1, 8, 2 * 17 * 19) == 0  # round 4: 17 and 19 are radices of the mixed-radix kernel
    assert lib.pm_fft1_workspace(L.PM_C64, 1, 8, 2 * 23 * 29) > 0   # primes above 19: Bluestein


def test_workspace_queries_of_the_bluestein_path(lib, bluestein_route):
    """Lengths that are not powers of two: the workspace queries are pure host arithmetic (csrc/bluestein.h: MB = power of two >= 2n - 1)."""
    from prysm_amd import _lib as L

    def a256(b):
        return (b + 255) // 256 * 256
    d = L.pm_fft2_desc()
    d.dtype, d.direction = L.PM_C64, -1
    # both axes on the path: [a (M x N) | c (M x N) | workspace of the fused 2048 x 2048 convolution chain with 1000 stored rows]
    d.in_y = d.in_x = d.out_y = d.out_x = L.pm_axis(1000, 1000, 0, 500)
    d.in_ld = d.out_ld = 1000
    fused = a256(1000 * 2048 * 8) + a256(2048 * 2048 * 8)
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 2 * a256(1000 * 1000 * 8) + fused
    # one axis a power of two: natural intermediate + the per-axis scratch (pre-multiplied lines + their MB-point transforms)
    d.in_y = d.out_y = L.pm_axis(1024, 1024, 0, 512)
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == a256(1024 * 1000 * 8) + a256(1024 * 1000 * 8) + a256(1024 * 2048 * 8)
    # short lengths stay on the direct kernel: just the natural intermediate
    d.in_y = d.in_x = d.out_y = d.out_x = L.pm_axis(36, 36, 0, 18)
    d.in_ld = d.out_ld = 36
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 36 * 36 * 8
    # powers of two above the engine's 8192: two M x N arrays (split planes + sub-lattice transforms)
    d.in_y = d.out_y = L.pm_axis(16384, 16384, 0, 8192)
    d.in_x = d.out_x = L.pm_axis(2048, 2048, 0, 1024)
    d.in_ld = d.out_ld = 2048
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 2 * 16384 * 2048 * 8
    # both axes Bluestein lengths above 4096: [a | c | spectrum (MB1 x MB2) | two more of that for the big transforms]
    d.in_y = d.out_y = L.pm_axis(5000, 5000, 0, 0)
    d.in_x = d.out_x = L.pm_axis(4500, 4500, 0, 0)
    d.in_ld = d.out_ld = 4500
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 2 * a256(5000 * 4500 * 8) + 3 * 16384 * 16384 * 8
    # mixed shapes whose awkward axis the other paths cannot take alone convolve BOTH axes
    d.in_y = d.out_y = L.pm_axis(8000, 8000, 0, 0)
    d.in_x = d.out_x = L.pm_axis(8192, 8192, 0, 0)
    d.in_ld = d.out_ld = 8192
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 2 * a256(8000 * 8192 * 8) + 3 * 16384 * 16384 * 8
    d.in_y = d.out_y = L.pm_axis(16384, 16384, 0, 0)
    d.in_x = d.out_x = L.pm_axis(1000, 1000, 0, 0)
    d.in_ld = d.out_ld = 1000
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 2 * a256(16384 * 1000 * 8) + 3 * 32768 * 2048 * 8
    # ... 32768 points beside a non power of two is beyond that: its columns stay on the direct kernel (the 1000-point rows
    # take the axis-by-axis form: natural intermediate + pre-multiplied rows + their 2048-point transforms)
    d.in_y = d.out_y = L.pm_axis(32768, 32768, 0, 0)
    assert lib.pm_fft2_workspace(ctypes.byref(d)) == 32768 * 1000 * 8 + a256(32768 * 1000 * 8) + a256(32768 * 2048 * 8)
    # 1-D: rows of 1000 points (MB = 2048); powers of two, short lengths and lengths above 4096 need none
    assert lib.pm_fft1_workspace(L.PM_C128, 1, 300, 1000) == a256(300 * 1000 * 16) + a256(300 * 2048 * 16)
    assert lib.pm_fft1_workspace(L.PM_C64, 0, 64, 777) == a256(64 * 777 * 8) + a256(64 * 2048 * 8)
    for n in (1024, 36, 5000):
        assert lib.pm_fft1_workspace(L.PM_C64, 1, 8, n) == 0
    # argument errors of the newer entry points are reported before any device work
    t = L.pm_axis(1000, 1000, 0, 0)
    assert lib.pm_fft1_ws(L.PM_C64, 0, 1, 4, ctypes.byref(t), ctypes.byref(t), 1.0, ctypes.c_void_p(16), 1000, ctypes.c_void_p(16), 1000,
                          None, 0, None) == L.PM_ERR_ARG
    assert b'direction' in lib.pm_last_error()
    assert lib.pm_encircled_energy_workspace() == 1024 * 8 * 8
    assert lib.pm_encircled_energy(L.PM_C128, 8, 8, ctypes.c_void_p(16), 8, 1.0, 0, None, ctypes.cast(ctypes.c_void_p(16), ctypes.c_void_p),
                                   None, 0, None) == L.PM_ERR_WORKSPACE
    assert lib.pm_spline_prefilter(L.PM_C128, 1, 8, 8, ctypes.c_void_p(16), 8, ctypes.c_void_p(16), 32, None) == L.PM_ERR_UNSUPPORTED
    assert lib.pm_spline_prefilter(L.PM_C128, 3, 8, 8, ctypes.c_void_p(16), 8, ctypes.c_void_p(16), 16, None) == L.PM_ERR_ARG


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
