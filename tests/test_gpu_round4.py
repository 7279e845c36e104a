"""GPU parity tests added in round 4:

* the grouped-wavelength kernels that keep four waves per SIMD (csrc/fft_spectral2.h): groups of 2 / 3 / 4, raw values re-read or kept,
  plain, padded (Q = 2), folded (4096-row) shapes, ragged last groups -- against the per-wavelength loop AND against numpy fp64;
* per-thread tuning (pm_set_tuning_local): two host threads on two streams with different route knobs, results against numpy;
* a misaligned real input with a Hermitian-only epilogue is refused (not silently run on the complex path);
  MDFT.intensity's fallback finishes with ONE more product; a composite length beside a length that needs the radix-R split.

Tolerances as in test_gpu_parity.py (max error / max magnitude against fp64): 1e-10 complex128, 5e-6 complex64 transforms.
"""
import ctypes
import math
import threading

import numpy as np
import pytest
import torch

from conftest import rel_max
from oracle import prysm_oracle as O

pytestmark = pytest.mark.gpu

TOL64 = 1e-10
TOL32 = 5e-6


@pytest.fixture(scope='module')
def pa():
    import prysm_amd
    from prysm_amd import _lib
    _lib.load()   # fails loudly when the HIP library is missing
    assert torch.cuda.is_available()
    return prysm_amd


def tonp(x):
    from prysm_amd.mathops import array_to_true_numpy
    if hasattr(x, 'data') and not isinstance(x, (np.ndarray, torch.Tensor)):
        x = x.data
    return array_to_true_numpy(x)


def crandn(rng, shape, dtype=np.complex128):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)


# ----------------------------------------------------------------------------- grouped wavelengths, four waves per SIMD

def _poly_numpy(amp, opd, ks, wts, Q):
    """sum_b w_b |focus(amp exp(i k_b opd), Q)|^2 in fp64 (the how-to's loop: Polychromatic Propagation.ipynb cell 3)"""
    acc = 0.0
    for k, w in zip(ks, wts):
        acc = acc + w * O.intensity(O.focus(amp.astype(np.float64) * np.exp(1j * k * opd.astype(np.float64)), Q))
    return acc


# (test_spectral2_groups_vs_loop_and_numpy went with the kernels it tested: the grouped-wavelength kernels at four waves per SIMD lost to
# the loop at every size -- profiles/r04/exp_spectral2.log -- and left the library in round 5; experiments/README.md)


def test_spectral_call_config5_shape_vs_numpy(pa):
    """pm_fft2_spectral at BASELINE config 5's shape (4096^2 fp32 maps, Q = 1) on the default route: 5 wavelengths against numpy fp64
    and against the loop; the knob of the removed grouped kernels is refused"""
    from prysm_amd import _lib, _ops
    from prysm_amd.propagation import focus_intensity
    lib = _lib.load()
    assert lib.pm_set_tuning_local(b'spectral2', 4) == _lib.PM_ERR_UNSUPPORTED
    lib.pm_reset_tuning_local()
    n = 4096
    rng = np.random.default_rng(5)
    ax = (np.arange(n) - n // 2) * (10.0 / n)
    r = np.hypot(ax[None, :], ax[:, None])
    amp = (r <= 5).astype(np.float32)
    opd = (500.0 * (r / 5) ** 4 + 5 * rng.standard_normal((n, n))).astype(np.float32)
    wl = np.linspace(0.5, 0.7, 5)
    ks = [2 * math.pi / w / 1e3 for w in wl]
    wts = [1.0, 0.5, 2.0, 1.5, 0.75]
    packed = _ops.pack_amp_opd(torch.from_numpy(amp).cuda(), torch.from_numpy(opd).cuda())

    def run(**knobs):
        acc = torch.zeros((n, n), device='cuda', dtype=torch.float32)
        with _lib.tuning_local(**knobs):
            focus_intensity(packed, 1, out=acc, synth=('packed', ks[0]), spectral=(ks, wts))
        return acc.cpu().numpy().astype(np.float64)

    ref = _poly_numpy(amp, opd, ks, wts, 1)
    loop = run(spectral=1)
    assert rel_max(loop, ref) < 2 * TOL32
    assert rel_max(run(), loop) < 1e-6


# ----------------------------------------------------------------------------- per-thread tuning

def test_two_threads_with_private_tuning(pa):
    """two host threads, each on its own stream with its own route knobs (pm_set_tuning_local): thread A transforms a composite grid
    on the mixed-radix kernel with the fold off, thread B the same grid through Bluestein (mix = 0) with the fold forced -- 30 rounds
    each, interleaved by the scheduler; every result against numpy, and the process-wide values untouched afterwards"""
    from prysm_amd import _lib, _ops
    lib = _lib.load()
    rng = np.random.default_rng(11)
    xa = crandn(rng, (600, 750), np.complex128)
    xb = crandn(rng, (256, 2048), np.complex64)      # rows of 2048 samples: the forced fold is legal
    wa, wb = np.fft.fft2(xa), np.fft.fft2(xb.astype(np.complex128))
    errs, fails = {}, []

    def worker(name, knobs):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st), _lib.tuning_local(**knobs):
                da, db = torch.from_numpy(xa).cuda(), torch.from_numpy(xb).cuda()
                worst = 0.0
                for _ in range(30):
                    ga = _ops.fft2(da, direction=-1, scale=1.0)
                    gb = _ops.fft2(db, direction=-1, scale=1.0)
                    st.synchronize()
                    worst = max(worst, rel_max(ga.cpu().numpy(), wa) / TOL64, rel_max(gb.cpu().numpy(), wb) / TOL32)
                errs[name] = worst
        except Exception as exc:      # surfaced in the main thread
            fails.append((name, repr(exc)))

    ta = threading.Thread(target=worker, args=('A', dict(mix=1, fold=0)))
    tb = threading.Thread(target=worker, args=('B', dict(mix=0, fold=1)))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not fails, fails
    assert errs['A'] < 1.0 and errs['B'] < 1.0, errs
    # the main thread never took a private copy: it still plans with the process-wide defaults (mix = 1: no Bluestein scratch)
    got = _ops.fft2(torch.from_numpy(xa).cuda(), direction=-1, scale=1.0).cpu().numpy()
    assert rel_max(got, wa) < TOL64
    assert lib.pm_set_tuning_local(b'no_such_knob', 1) == 0      # unknown keys are ignored, as in pm_set_tuning
    lib.pm_reset_tuning_local()


# ----------------------------------------------------------------------------- ADVICE r3

def test_misaligned_real_input_refuses_hermitian_epilogues(pa):
    """a float32 field whose base address is 4 mod 8 cannot take the Hermitian path (it reads the array as pairs); the complex path has
    no |.| / angle / centre normalisation, so PM_EPI_ABS, PM_EPI_ARG and PM_FLAG_NORM_DC are refused (rc = PM_ERR_UNSUPPORTED) instead
    of returning accumulated |.|^2 with rc = 0; a plain spectrum of the same view still runs (complex path) and is right"""
    from prysm_amd import _lib as L, _ops
    lib = L.load()
    n = 256
    rng = np.random.default_rng(3)
    base = torch.from_numpy(rng.random(n * n + 1).astype(np.float32)).cuda()
    view = base[1:].view(n, n)              # 4 bytes past an 8-byte boundary
    assert view.data_ptr() % 8 == 4
    d = L.pm_fft2_desc()
    d.dtype = L.PM_C64
    d.direction = -1
    d.scale = 1.0
    d.weight = 1.0
    d.in_y = d.in_x = d.out_y = d.out_x = _ops._axis(n, n, 0, 0)
    d.in_ld = d.out_ld = n
    d.flags = L.PM_FLAG_REAL_INPUT
    nbytes = lib.pm_fft2_workspace(ctypes.byref(d))
    ws = torch.empty(max(int(nbytes), 1) * 2, dtype=torch.uint8, device='cuda')
    outr = torch.zeros((n, n), dtype=torch.float32, device='cuda')
    for epi, flags in ((L.PM_EPI_ABS, 0), (L.PM_EPI_ARG, 0), (L.PM_EPI_ABS2, L.PM_FLAG_NORM_DC)):
        d.epilogue = epi
        d.flags = L.PM_FLAG_REAL_INPUT | flags
        rc = lib.pm_fft2(ctypes.byref(d), view.data_ptr(), outr.data_ptr(), ws.data_ptr(), ws.numel(), L.stream_ptr())
        assert rc == L.PM_ERR_UNSUPPORTED, (epi, flags, rc)
        assert float(outr.abs().max()) == 0.0
    d.epilogue = L.PM_EPI_NONE
    d.flags = L.PM_FLAG_REAL_INPUT
    outc = torch.zeros((n, n), dtype=torch.complex64, device='cuda')
    L.check(lib.pm_fft2(ctypes.byref(d), view.data_ptr(), outc.data_ptr(), ws.data_ptr(), ws.numel(), L.stream_ptr()))
    assert rel_max(outc.cpu().numpy(), np.fft.fft2(view.cpu().numpy().astype(np.float64))) < TOL32


def test_mdft_intensity_fallback_finishes_with_one_product(pa, monkeypatch):
    """MDFT.intensity on a shape / precision the fused |.|^2 epilogue does not take (complex128; 50 x 70 samples) must equal
    |executor(x)|^2 and run TWO products, not three (the fallback used to start over with self(ary))"""
    from prysm_amd import _ops
    rng = np.random.default_rng(8)
    x = crandn(rng, (96, 80))
    ex = pa.propagation.prepare_executor(0.05, (96, 80), 1.0, (50, 70), O.HeNe, 100.0, kind='mdft')
    calls = []
    real = _ops.cgemm
    monkeypatch.setattr(_ops, 'cgemm', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    acc = torch.full((50, 70), 2.0, dtype=torch.float64, device='cuda')
    got = tonp(ex.intensity(torch.from_numpy(x).cuda(), out=acc, weight=0.5))
    assert len(calls) == 2
    ref = O.prepare_executor(0.05, (96, 80), 1.0, (50, 70), O.HeNe, 100.0)(x)
    assert rel_max(got - 2.0, 0.5 * np.abs(ref) ** 2) < TOL64


def test_composite_length_beside_a_split_length(pa):
    """a composite length the mixed-radix kernel owns (96 = 3 * 32) beside a power of two that needs the radix-2 step (native length
    lowered to 64: 128 splits) takes the radix-R path on BOTH axes (such shapes -- 1536 x 16384 at full size -- fell to the
    both-axes Bluestein form); against numpy, both orientations and both precisions"""
    from prysm_amd import _lib, _ops
    rng = np.random.default_rng(21)
    with _lib.tuning_local(big_native_log=6):
        for shape in ((96, 128), (128, 96), (160, 256)):
            for dtype, tol in ((np.complex64, TOL32), (np.complex128, TOL64)):
                x = crandn(rng, shape, dtype)
                got = _ops.fft2(torch.from_numpy(x).cuda(), direction=-1, scale=1.0).cpu().numpy()
                assert rel_max(got, np.fft.fft2(x.astype(np.complex128))) < tol, (shape, dtype)
                ref = O.focus(x.astype(np.complex128), 1)
                assert rel_max(tonp(pa.propagation.focus(x, 1)), ref) < tol, (shape, dtype)


# ----------------------------------------------------------------------------- fused chain on composite grids

@pytest.mark.parametrize('shape,dtype,tol', [((1000, 1500), np.complex128, TOL64), ((600, 750), np.complex64, TOL32),
                                             ((1000, 1024), np.complex128, TOL64), ((360, 2048), np.complex64, TOL32),
                                             ((105, 154), np.complex128, TOL64)])
def test_angular_spectrum_on_composite_grids(pa, shape, dtype, tol):
    """angular_spectrum / its adjoint / tf= on grids whose column length is composite (primes <= 13) -- three passes with the
    mixed-radix middle pass (forward stages, x H, transposed stages in LDS) -- against the oracle and against the composed route
    (two pm_fft2 calls, knob mix_fused = 0); row lengths composite and powers of two; Q = 1 and a padded Q = 2 input"""
    from prysm_amd import _lib
    rng = np.random.default_rng(shape[0] + shape[1])
    prec = pa.config.precision
    pa.config.precision = 32 if dtype == np.complex64 else 64
    try:
        x = crandn(rng, shape, dtype)
        ref = O.angular_spectrum(x.astype(np.complex128), O.HeNe, 0.01, 10.0, Q=1)
        got = tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1))
        assert got.dtype == dtype and rel_max(got, ref) < tol
        with _lib.tuning_local(mix_fused=0):
            comp = tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1))
        assert rel_max(comp, ref) < tol and rel_max(got, comp) < 2 * tol
        # padded input (only the stored rows are transformed in the first pass; the middle pass synthesises the zero rows), and
        # the adjoint: conj(H) and a crop of the rows / columns in the last pass
        small = np.ascontiguousarray(x[:shape[0] // 2, :shape[1] // 2])
        refq = O.angular_spectrum(small.astype(np.complex128), O.HeNe, 0.01, 10.0, Q=2)
        assert rel_max(tonp(pa.propagation.angular_spectrum(small, O.HeNe, 0.01, 10.0, Q=2)), refq) < tol
        refa = O.angular_spectrum_adjoint(x.astype(np.complex128), O.HeNe, 0.01, 10.0, Q=2)
        assert rel_max(tonp(pa.propagation.angular_spectrum_adjoint(x, O.HeNe, 0.01, 10.0, Q=2)), refa) < tol
        tf = O.angular_spectrum_transfer_function(shape, O.HeNe, 0.01, 7.0)
        reft = O.angular_spectrum(x.astype(np.complex128), O.HeNe, 0.01, 7.0, Q=1, tf=tf)
        assert rel_max(tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 7.0, Q=1, tf=tf.astype(dtype))), reft) < 2 * tol
    finally:
        pa.config.precision = prec


@pytest.mark.parametrize('shape', [(300, 500), (375, 250), (96, 1000)])
def test_conv_on_composite_grids(pa, shape):
    """conv / apply_transfer_functions on composite grids: both rotations ride on the chain -- the input rotation in the first two passes'
    loads, the output rotation as two runs of rows in the last pass -- complex and real objects, against the oracle"""
    from prysm_amd import convolution as C
    rng = np.random.default_rng(sum(shape))
    o = crandn(rng, shape)
    h = crandn(rng, shape)
    assert rel_max(tonp(C.conv(o, h)), O.conv(o, h)) < 1e-9
    orl = rng.standard_normal(shape)
    assert rel_max(tonp(C.conv(orl, h)), O.conv(orl, h)) < 1e-9
    tf = crandn(rng, shape)
    assert rel_max(tonp(C.apply_transfer_functions(o, 1.0, [tf])), O.apply_transfer_functions(o, 1.0, [tf])) < 1e-9
    assert rel_max(tonp(C.apply_transfer_functions(o.astype(np.complex64), 1.0, [tf.astype(np.complex64)], shift=True)),
                   O.apply_transfer_functions(o, 1.0, [tf], shift=True)) < 2e-5


# ----------------------------------------------------------------------------- pupil synthesis inside the mixed-radix row kernel

@pytest.mark.parametrize('shape,Q', [((600, 750), 1), ((500, 500), 1.5), ((1000, 1536), 1), ((300, 400), 2)])
def test_pupil_synthesis_in_the_load_on_composite_grids(pa, shape, Q):
    """Wavefront.from_amp_and_phase(...).focus() / .focus_intensity() and the polychromatic driver on grids whose (padded) row length is a
    composite of primes <= 13: the pupil is synthesised by the first stage of the mixed-radix row kernel (the lazy wavefront never
    materialises it), float32 and float64 maps, float / bool / no amplitude, packed maps through the wavelength loop -- vs the oracle"""
    from prysm_amd.polychromatic import polychromatic_psf
    rng = np.random.default_rng(int(shape[0] * Q))
    ampf = (rng.random(shape) * (rng.random(shape) > 0.2)).astype(np.float32)
    ampb = rng.random(shape) > 0.3
    for rd, tol in ((np.float32, 1e-5), (np.float64, TOL64)):
        opd = (150 * rng.standard_normal(shape)).astype(rd)
        for amp in (ampf.astype(rd), ampb, None):
            a64 = np.ones(shape) if amp is None else amp.astype(np.float64)
            pref = O.focus(O.from_amp_and_phase(a64, opd.astype(np.float64), O.HeNe), Q)
            wf = pa.propagation.Wavefront.from_amp_and_phase(amp if amp is not None else np.ones(shape, dtype=rd), opd, O.HeNe, 0.04)
            assert wf._fusable(Q) is not None
            got = tonp(wf.focus(100.0, Q=Q).data)
            assert wf._data is None                    # never materialised
            assert rel_max(got, pref) < tol, (rd, None if amp is None else amp.dtype)
            I = tonp(wf.focus_intensity(100.0, Q=Q).data)
            assert rel_max(I, O.intensity(pref)) < 2 * tol
    opd = (150 * rng.standard_normal(shape)).astype(np.float32)
    wv, wt = np.linspace(0.5, 0.7, 5), np.linspace(1.0, 2.0, 5)
    poly = tonp(polychromatic_psf(ampf, opd, wv, wt, 0.04, 100.0, Q=Q))
    pw = sum(w * O.intensity(O.focus(O.from_amp_and_phase(ampf.astype(np.float64), opd.astype(np.float64), float(l)), Q)) for l, w in zip(wv, wt))
    assert poly.dtype == np.float32 and rel_max(poly, pw) < 2e-5


# ----------------------------------------------------------------------------- composites above 8192

@pytest.mark.parametrize('shape', [(10000, 96), (96, 10000), (9000, 3000), (12000, 128), (64, 20000)])
def test_composite_lengths_above_8192(pa, shape):
    """lengths above 8192 whose cofactor of 2 .. 7 is a composite the mixed-radix kernel takes (10000 = 2 x 5000, 9000 = 2 x 4500,
    12000 = 2 x 6000, 20000 = 4 x 5000) run as one radix-R step around mixed-radix sub-transforms instead of a Bluestein convolution at
    32768 points -- beside short composite / power-of-two axes; fft2, ifft2 and the focus family (rotations, pad window) vs numpy"""
    from prysm_amd import _ops
    rng = np.random.default_rng(sum(shape))
    for dtype, tol in ((np.complex64, TOL32), (np.complex128, TOL64)):
        x = crandn(rng, shape, dtype)
        xd = torch.from_numpy(x).cuda()
        assert rel_max(_ops.fft2(xd, direction=-1, scale=1.0).cpu().numpy(), np.fft.fft2(x.astype(np.complex128))) < tol, dtype
        M, N = shape
        assert rel_max(_ops.fft2(xd, direction=+1, scale=1.0 / (M * N)).cpu().numpy(), np.fft.ifft2(x.astype(np.complex128))) < tol
        assert rel_max(tonp(pa.propagation.focus(x, 1)), O.focus(x.astype(np.complex128), 1)) < tol
    small = crandn(rng, (shape[0] // 2, shape[1] // 2), np.complex128)
    assert rel_max(tonp(pa.propagation.focus(small, 2)), O.focus(small, 2)) < TOL64
    xr = rng.standard_normal(shape)
    assert rel_max(_ops.fft2(torch.from_numpy(xr).cuda(), direction=-1, scale=1.0).cpu().numpy(), np.fft.fft2(xr)) < TOL64


# ----------------------------------------------------------------------------- small host-side additions

def test_stream_ring_sequence_matches_one_stream(pa):
    """prysm_amd.graph.StreamRing: a sequence of independent propagations alternating between two HIP streams gives, bit for bit, what
    one stream gives (per-stream workspaces, stream-aware allocator), and join() orders the caller's stream behind all of them"""
    from prysm_amd.graph import StreamRing
    rng = np.random.default_rng(4)
    fields = [torch.from_numpy(crandn(rng, (512, 512), np.complex64)).cuda() for _ in range(6)]
    want = [pa.propagation.focus(f, 2).clone() for f in fields]
    torch.cuda.synchronize()
    ring = StreamRing(2)
    ring.fork()
    outs = [ring.run(pa.propagation.focus, f, 2) for f in fields]
    ring.join()
    total = sum(o.abs().sum() for o in outs)        # consumed on the caller's stream, after the join
    torch.cuda.synchronize()
    assert all(torch.equal(o, w) for o, w in zip(outs, want)) and float(total) > 0
    assert StreamRing.worth_it((2048, 2048)) and not StreamRing.worth_it((4096, 4096))


def test_transfer_function_vectors_are_cached_per_scalars(pa):
    """angular_spectrum re-uses the two transfer-function vectors of (shape, wavelength, dx, z): same tensors on a repeat, new ones for
    another distance, results right either way; the materialised transfer function never hands cached storage out"""
    from prysm_amd import _ops
    rng = np.random.default_rng(6)
    x = crandn(rng, (256, 256))
    a = _ops.as_tf_vectors((256, 256), O.HeNe, 0.01, 10.0, torch.complex128)
    b = _ops.as_tf_vectors((256, 256), O.HeNe, 0.01, 10.0, torch.complex128)
    c = _ops.as_tf_vectors((256, 256), O.HeNe, 0.01, 11.0, torch.complex128)
    assert a[0] is b[0] and a[1] is b[1] and c[0] is not a[0]
    for z in (10.0, 11.0, 10.0):
        assert rel_max(tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, z, Q=1)), O.angular_spectrum(x, O.HeNe, 0.01, z, Q=1)) < TOL64
    tf = pa.propagation.angular_spectrum_transfer_function((256, 256), O.HeNe, 0.01, 10.0)
    assert rel_max(tonp(tf), O.angular_spectrum_transfer_function((256, 256), O.HeNe, 0.01, 10.0)) < TOL64
    assert len(_ops._AS_TF_CACHE) <= _ops._AS_TF_CACHE_MAX


@pytest.mark.parametrize('shape', [(1020, 1900), (323, 380), (272, 4913), (2048, 1020)])
def test_lengths_with_the_primes_17_and_19(pa, shape):
    """radices 17 and 19 in the mixed-radix kernel (round 4): 1020 = 6 x 10 x 17, 1900 = 10 x 10 x 19, 323 = 17 x 19, 4913 = 17^3 no longer
    convolve through Bluestein; fft2 / ifft2 / focus in both precisions and the fused chain (column lengths 1020, 323) vs numpy"""
    from prysm_amd import _ops
    rng = np.random.default_rng(sum(shape))
    M, N = shape
    for dtype, tol in ((np.complex64, TOL32), (np.complex128, TOL64)):
        x = crandn(rng, shape, dtype)
        xd = torch.from_numpy(x).cuda()
        assert rel_max(_ops.fft2(xd, direction=-1, scale=1.0).cpu().numpy(), np.fft.fft2(x.astype(np.complex128))) < tol, dtype
        assert rel_max(_ops.fft2(xd, direction=+1, scale=1.0 / (M * N)).cpu().numpy(), np.fft.ifft2(x.astype(np.complex128))) < tol
        assert rel_max(tonp(pa.propagation.focus(x, 1)), O.focus(x.astype(np.complex128), 1)) < tol
    x = crandn(rng, shape)
    assert rel_max(tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1)), O.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1)) < TOL64


@pytest.mark.gpu
def test_knobs_of_removed_variants_are_refused(pa):
    """the variants that lost their measurements left the library in round 5 (experiments/README.md): their knob values answer
    PM_ERR_UNSUPPORTED -- nothing else runs in their place -- and the shipped values are still accepted"""
    from prysm_amd import _lib
    lib = _lib.load()
    try:
        for key, v in ((b'mix_fold', 1), (b'mix_pers', 1), (b'two_units', 1), (b'engine_p8', 1), (b'spectral2', 2), (b'colmul_mode', 1),
                       (b'colmul_mode', 2), (b'gemm_wk', 2), (b'gemm_3m', 0), (b'mix_ablate', 1), (b'spectral_mode', 0)):
            assert lib.pm_set_tuning_local(key, v) == _lib.PM_ERR_UNSUPPORTED, key
        for key, v in ((b'colmul_mode', 3), (b'colmul_mode', 0), (b'gemm_wk', 1), (b'spectral_mode', 3), (b'stagger_group', 1)):
            assert lib.pm_set_tuning_local(key, v) == 0, key
    finally:
        lib.pm_reset_tuning_local()


@pytest.mark.gpu
def test_start_up_stagger_changes_timing_only(pa):
    """The start-up stagger (engine: fft_stagger / fft_stagger_col, 100 + units forces it on single-round launches; mixed-radix column
    kernel: mix_stagger) delays workgroups of the first round by a hash of their index and must not change a bit of any result."""
    from prysm_amd import _lib, _ops
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(11)
    cases = [((4096, 4096), torch.complex64), ((2048, 2048), torch.complex64), ((1024, 4096), torch.complex128), ((3000, 3000), torch.complex64)]
    try:
        for shape, dt in cases:
            rdt = torch.float32 if dt == torch.complex64 else torch.float64
            x = torch.complex(torch.randn(shape, device='cuda', dtype=rdt, generator=g), torch.randn(shape, device='cuda', dtype=rdt, generator=g))
            h = (shape[0] // 2, shape[1] // 2)
            outs = []
            for r, c, m in ((0, 0, 0), (3, 5, 9), (108, 104, 1)):
                assert lib.pm_set_tuning_local(b'stagger_group', 1 if m == 9 else 0) == 0
                assert lib.pm_set_tuning_local(b'fft_stagger', r) == 0
                assert lib.pm_set_tuning_local(b'fft_stagger_col', c) == 0
                assert lib.pm_set_tuning_local(b'mix_stagger', m) == 0
                outs.append(_ops.fft2(x, direction=-1, scale=1.0, in_shift=h, out_shift=h))
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), shape
    finally:
        lib.pm_reset_tuning_local()
