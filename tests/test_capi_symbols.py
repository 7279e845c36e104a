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
    assert lib.pm_version() == 107


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


# ----------------------------------------------------------------------------- routes (pm_plan_explain: host logic, no GPU)

def _route(lib, m, n, dt='c64', op=0, Q=None, real=False, mul=False, epi=0, batch=0, synth=False):
    from prysm_amd import _lib as L
    d = L.pm_fft2_desc()
    d.dtype, d.direction = (L.PM_C64 if dt == 'c64' else L.PM_C128), -1
    if Q:
        d.in_y = L.pm_axis(m, m // Q, (m - m // Q) // 2, m // 2)
        d.in_x = L.pm_axis(n, n // Q, (n - n // Q) // 2, n // 2)
        d.in_ld = n // Q
    else:
        d.in_y, d.in_x, d.in_ld = L.pm_axis(m, m, 0, m // 2), L.pm_axis(n, n, 0, n // 2), n
    d.out_y, d.out_x, d.out_ld = L.pm_axis(m, m, 0, m // 2), L.pm_axis(n, n, 0, n // 2), n
    if real:
        d.flags |= L.PM_FLAG_REAL_INPUT
    if synth:
        d.flags |= L.PM_FLAG_SYNTH_INPUT
    d.epilogue = epi
    if mul:
        d.mul_kind, d.mul, d.mul_x = L.PM_MUL_SEPARABLE, 16, 16
    if batch:
        d.batch, d.in_bstride, d.out_bstride = batch, m * n, m * n
    buf = ctypes.create_string_buffer(256)
    L.check(lib.pm_plan_explain(ctypes.byref(d), op, buf, 256))
    return buf.value.decode()


# (arguments of _route, substrings the line must contain): the routes of round 5.  A change of the planner that moves a shape shows up
# here, on the CPU box -- the 1536 x 16384 row is the routing bug ADVICE r3 found with a stopwatch.
ROUTES = [
    ((4096, 4096), ['route=engine-fold', 'rows=stockham(4096)', 'cols=stockham(2048x2)', 'tile=8', 'log_k=7', 'ws=134217728']),
    ((4096, 4096, 'c128'), ['route=engine-fold', 'cols=stockham(2048x2)', 'log_k=2', 'ws=268435456']),
    ((8192, 8192), ['route=engine-fold', 'rows=stockham(8192)', 'cols=stockham(4096x2)', 'log_k=3']),
    ((8192, 8192, 'c128'), ['route=engine-fold', 'cols=stockham(4096x2)']),
    ((2048, 2048), ['route=engine ', 'rows=stockham(2048)', 'cols=stockham(2048)', 'ws=33554432']),
    ((1024, 1024), ['route=engine ', 'log_k=1']),
    ((512, 512, 'c128'), ['route=engine ']),
    ((64, 16), ['route=engine ', 'rows=stockham(16)', 'cols=stockham(64)']),
    ((2048, 8192), ['route=engine ', 'rows=stockham(8192)', 'cols=stockham(2048)']),
    ((8192, 2048), ['route=engine-fold', 'cols=stockham(4096x2)']),
    ((16384, 16384), ['route=radix-step', 'rows=2xstockham(8192)', 'cols=2xstockham(8192)']),
    ((32768, 4096), ['route=radix-step', 'cols=4xstockham(8192)', 'rows=1xstockham(4096)']),
    ((1536, 16384), ['route=radix-step', 'rows=2xstockham(8192)', 'cols=3xstockham(512)']),
    ((6144, 12288), ['route=radix-step', 'rows=3xstockham(4096)', 'cols=3xstockham(2048)']),
    ((10000, 10000), ['route=radix-step', 'rows=2xmixed-radix(5000)', 'cols=2xmixed-radix(5000)']),
    ((9000, 12000), ['route=radix-step', 'rows=2xmixed-radix(6000)', 'cols=2xmixed-radix(4500)']),
    ((20000, 4096), ['route=radix-step', 'cols=4xmixed-radix(5000)']),
    ((3000, 3000), ['route=natural-mixed', 'rows=mixed-radix-registers(3000)', 'cols=mixed-radix-registers(3000)', 'ws=72192000']),     # round 5: the composite register engine (fft_ce.h)
    ((3000, 3000, 'c128'), ['route=natural-mixed', 'rows=mixed-radix-registers(3000)']),
    ((1000, 1000), ['route=natural-mixed', 'rows=mixed-radix-registers(1000)', 'ws=8064000']),
    ((1800, 4500), ['rows=mixed-radix-registers(4500)', 'cols=mixed-radix-registers(1800)']),
    ((1200, 2400), ['rows=mixed-radix(2400)', 'cols=mixed-radix(1200)']),      # no compile-time plan: the general kernel
    ((6006, 6006), ['route=natural-mixed', 'cols=mixed-radix(6006)']),
    ((1020, 1900), ['route=natural-mixed', 'rows=mixed-radix(1900)', 'cols=mixed-radix(1020)']),
    ((323, 380), ['route=natural-mixed', 'rows=mixed-radix(380)', 'cols=mixed-radix(323)']),
    ((4096, 3000), ['route=natural-mixed', 'rows=mixed-radix-registers(3000)', 'cols=stockham(4096)']),
    ((3000, 4096), ['route=natural-mixed', 'rows=stockham(4096)', 'cols=mixed-radix-registers(3000)']),
    ((1536, 1536), ['route=natural-mixed', 'rows=mixed-radix-registers(1536)']),
    ((36, 36), ['route=natural-mixed']),
    ((24, 24), ['route=natural ', 'rows=direct(24)', 'cols=direct(24)']),
    ((997, 997), ['route=bluestein-2d', 'conv=2048x2048']),
    ((2018, 2018), ['route=bluestein-2d', 'conv=4096x4096']),
    ((1024, 1009), ['route=natural ', 'rows=bluestein(1009)', 'cols=stockham(1024)']),
    ((4099, 4099), ['route=bluestein-2d-big', 'conv=16384x16384']),
]
ROUTES_KW = [
    (dict(m=4096, n=4096, Q=2), ['route=engine ', 'cols=stockham(4096)', 'log_k=2', 'ws=67108864']),      # padded: unfolded, stored rows only
    # round 6: the transposed Hermitian form (real-input column transforms, then rows stored with their mirror image) where it measured faster
    (dict(m=4096, n=4096, real=True, epi=3), ['route=hermitian-transposed', 'cols=stockham-r2c(4096)', 'rows=stockham(4096)x2048', 'ws=67108864']),
    (dict(m=1024, n=1024, real=True, epi=3), ['route=hermitian-transposed', 'rows=stockham(1024)x512']),
    (dict(m=4096, n=4096, dt='c128', real=True, epi=3), ['route=hermitian-fold', 'rows=stockham-r2c(2048)']),      # rows of 4096 complex128 points: the round-2 form
    (dict(m=2048, n=2048, dt='c128', real=True, epi=3), ['route=hermitian-transposed']),
    (dict(m=4096, n=8192, real=True, epi=3), ['route=hermitian-fold']),
    (dict(m=8192, n=2048, real=True, epi=3), ['route=hermitian-transposed']),
    (dict(m=8192, n=8192, real=True, epi=3), ['route=hermitian-fold', 'rows=stockham-r2c(4096)', 'cols=stockham(4096x2)', 'tile=8']),
    (dict(m=2048, n=2048, real=True), ['route=engine ']),               # a plain spectrum of a small real field stays on the complex path
    (dict(m=4096, n=4096, real=True), ['route=hermitian-transposed']),
    (dict(m=3000, n=3000, real=True), ['route=natural-mixed', 'rows=mixed-radix(3000)', 'cols=mixed-radix-registers(3000)']),         # composite grids: no Hermitian path yet (the real array is read by the complex kernels)
    (dict(m=1024, n=1024, batch=100), ['route=engine ', 'chunk=16', 'ws=134217728']),
    (dict(m=4096, n=4096, synth=True), ['route=engine-fold']),
    (dict(m=3000, n=3000, synth=True), ['rows=mixed-radix-registers(3000)']),      # the complex64 pupil is synthesised in the register engine's loads
    (dict(m=3000, n=3000, synth=True, dt='c128'), ['rows=mixed-radix(3000)']),       # ... the complex128 one in the general kernel's
    (dict(m=3000, n=3000, epi=1), ['cols=mixed-radix-registers(3000)']),    # |.|^2 in the engine's column store
    (dict(m=4096, n=4096, dt='c128', op=1, mul=True), ['route=fused ', 'passes=3', 'mid=stockham-pair fold', 'ws=268435456']),
    (dict(m=4096, n=4096, dt='c64', op=1, mul=True), ['route=fused ', 'fold']),
    (dict(m=2048, n=2048, dt='c128', op=1, mul=True), ['route=fused ', 'mid=stockham-pair ws=']),
    (dict(m=3000, n=3000, dt='c128', op=1, mul=True), ['route=fused-composite', 'mid=mixed-radix-resident']),
    (dict(m=3000, n=4096, dt='c64', op=1, mul=True), ['route=fused-composite', 'rows=stockham']),
    (dict(m=997, n=997, op=1, mul=True), ['route=composed']),
    (dict(m=4096, n=3000, op=1, mul=True), ['route=composed']),          # a composite ROW length beside engine columns: two transforms
]


def test_routes_of_a_table_of_shapes(lib):
    for args, want in ROUTES:
        line = _route(lib, *args)
        assert all(w in line for w in want), (args, line)
    for kw, want in ROUTES_KW:
        line = _route(lib, **kw)
        assert all(w in line for w in want), (kw, line)
    from prysm_amd import _lib as L
    d = L.pm_fft2_desc()
    d.dtype = 9
    buf = ctypes.create_string_buffer(256)
    assert lib.pm_plan_explain(ctypes.byref(d), 0, buf, 256) == L.PM_ERR_ARG
    assert lib.pm_plan_explain(ctypes.byref(d), 0, buf, 8) == L.PM_ERR_ARG


def test_python_asks_the_planner_which_grids_stack_on_the_register_engine(lib):
    """prysm_amd._ops.on_register_engine (what polychromatic_psf's default consults): both passes of a plain complex transform on the
    composite register engine -- from pm_plan_explain, no GPU"""
    import torch
    from prysm_amd import _lib as L, _ops
    assert _ops.on_register_engine(500, 500, torch.complex64) and _ops.on_register_engine(1000, 1536, torch.complex128)
    assert not _ops.on_register_engine(600, 600, torch.complex64)        # no compile-time plan
    assert not _ops.on_register_engine(1000, 1024, torch.complex64)      # one pass on the power-of-two engine
    with L.tuning_local(mix_engine=0):          # the cache follows tuning_local blocks
        assert not _ops.on_register_engine(500, 500, torch.complex64)
    assert _ops.on_register_engine(500, 500, torch.complex64)


def test_routes_follow_the_knobs_and_tuning_local_nests(lib):
    """leaving an inner tuning_local block used to discard the outer block's knobs; a block that fails to start (a knob the
    product build refuses) must leave nothing half-applied"""
    from prysm_amd import _lib as L
    assert 'mixed-radix' in _route(lib, 1000, 1000) and 'engine-fold' in _route(lib, 4096, 4096)
    with L.tuning_local(mix=0):
        assert 'bluestein' in _route(lib, 1000, 1000)
        with L.tuning_local(fold=0):
            assert 'bluestein' in _route(lib, 1000, 1000) and 'route=engine ' in _route(lib, 4096, 4096)
        assert 'bluestein' in _route(lib, 1000, 1000) and 'engine-fold' in _route(lib, 4096, 4096)
        with pytest.raises(NotImplementedError):
            with L.tuning_local(fold=0, spectral2=3):
                pass
        assert 'bluestein' in _route(lib, 1000, 1000) and 'engine-fold' in _route(lib, 4096, 4096)
    assert 'mixed-radix' in _route(lib, 1000, 1000)
    with L.tuning_local(log_k=0):
        assert 'log_k=0' in _route(lib, 4096, 4096)
    assert 'log_k=7' in _route(lib, 4096, 4096)
    with L.tuning_local(mix_engine=0):      # round 5: composite lengths with a compile-time plan back on the general kernel
        assert 'rows=mixed-radix(3000)' in _route(lib, 3000, 3000)
    assert 'rows=mixed-radix-registers(3000)' in _route(lib, 3000, 3000)
