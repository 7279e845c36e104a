"""GPU parity tests added in round 2:

* BASELINE config 5 at its stated size -- 64 wavelengths x 4096^2 fp32, variant F (FFT focus, both `batched` forms) and
  variant M (matrix-DFT focus to 512^2) -- against the oracle's single-process sum;
* the reference identities that were not yet restated on the device: array orientation / +y tilt
  (tests/test_physics.py:56-74), thin lens == Hopkins defocus (tests/test_propagation.py:440-460), finite-difference checks of
  the focal-plane-mask / Lyot gradients through the Wavefront API (tests/test_propagation.py:353-427, 589-622);
* the N > 1 paths (bench.py and the polychromatic driver) as two ranks: RCCL when two GPUs are visible, gloo with both
  ranks on one GPU otherwise;
* caller-supplied `out=` validation, boolean occulters, the Jones adapter's pass-through of stacks;
* later in the round: the wavelength loop as one call (pm_fft2_spectral, both precisions), 1-D transforms of 16384 / 32768 and
  3 / 5 / 7 x 2^k points, every np.pad mode of pad2d, the real convolution chain on half spectra (PM_FLAG_REAL_OUTPUT, folded and
  not), complex128 pupil synthesis inside the row load, the chirp-Z executor built from grid parameters, config 5 variant M through it.

Tolerances as in test_gpu_parity.py.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, rel_max
from oracle import prysm_oracle as O

pytestmark = pytest.mark.gpu

TOL64 = 1e-10
TOL32 = 5e-6
TOL32_MDFT = 3e-5


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


def _real_vdot(a, b):
    return float(np.real(np.vdot(np.asarray(a), np.asarray(b))))


# ----------------------------------------------------------------------------- BASELINE config 5 at size

@pytest.fixture(scope='module')
def config5():
    """4096^2 circular amplitude, 500 nm of W040, 64 wavelengths in [0.5, 0.7] um, uniform weights; fp32
    maps.  The oracle sums are computed once (scipy.fft on all host cores -- the checker, not the thing measured)."""
    from scipy import fft as sfft
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


@pytest.mark.parametrize('batched', [False, True])
def test_config5_variant_f_full_size(pa, config5, batched):
    """64 wavelengths x 4096^2 fp32, FFT focus Q = 1, |.|^2 and the weighted sum on the device: field by field with the
    accumulate epilogue (pupil synthesised inside the row pass), and as stacks + sum_of_2d_modes."""
    from prysm_amd.polychromatic import polychromatic_psf
    c = config5
    got = tonp(polychromatic_psf(c['amp'], c['opd'], c['wvls'], c['wts'], c['dx'], 100.0, Q=1, batched=batched))
    assert got.dtype == np.float32 and got.shape == (4096, 4096)
    assert rel_max(got, c['want_f']) < 2e-5     # 64 fp32 intensities accumulated in fp32 against the fp64 oracle sum
    assert abs(got.sum(dtype=np.float64) / c['want_f'].sum() - 1) < 1e-5     # energy (unitary transform: 64 x sum amp^2)


def test_config5_variant_m_full_size(pa, config5):
    """the how-to's variant: per wavelength prepare_executor + matrix-DFT focus 4096^2 -> 512^2 (MFMA) + |.|^2 accumulate"""
    from prysm_amd.conf import config
    from prysm_amd.polychromatic import polychromatic_psf
    c = config5
    prec = config.precision
    try:
        config.precision = 32
        got = tonp(polychromatic_psf(c['amp'], c['opd'], c['wvls'], c['wts'], c['dx'], 100.0, focal_dx=0.55 * 10 / 4,
                                     samples=512, kind='mdft'))
    finally:
        config.precision = prec
    assert got.dtype == np.float32 and got.shape == (512, 512)
    assert rel_max(got, c['want_m']) < 1e-4     # K = 4096 complex64 contractions, squared and summed 64 times


def test_config5_variant_m_by_chirp_z(pa, config5):
    """the same 512^2 focal grid per wavelength through the chirp-Z executor (kind='czt': fused convolution kernels, K = 8192) gives
    the image of the matrix-DFT variant"""
    from prysm_amd.conf import config
    from prysm_amd.polychromatic import polychromatic_psf
    c = config5
    prec = config.precision
    try:
        config.precision = 32
        got = tonp(polychromatic_psf(c['amp'], c['opd'], c['wvls'], c['wts'], c['dx'], 100.0, focal_dx=0.55 * 10 / 4,
                                     samples=512, kind='czt'))
    finally:
        config.precision = prec
    assert got.dtype == np.float32 and got.shape == (512, 512)
    assert rel_max(got, c['want_m']) < 1e-4


def test_packed_amp_opd_synthesis_equals_two_array_synthesis(pa):
    """PM_FLAG_SYNTH_PACKED: the pupil synthesised from (amplitude, OPD) pairs read as one 8-byte element is bit for bit the
    pupil synthesised from the two arrays (same arithmetic, different loads), folded and unfolded, padded and not"""
    from prysm_amd import _ops
    P = pa.propagation
    rng = np.random.default_rng(21)
    for n, Q in ((256, 1), (256, 2), (4096, 1)):
        amp = torch.from_numpy((rng.random((n, n)) > 0.3).astype(np.float32)).cuda()
        opd = torch.from_numpy((300 * rng.standard_normal((n, n))).astype(np.float32)).cuda()
        k = 2 * np.pi / 0.55 / 1e3
        a = P.focus_intensity(opd, Q, synth=(amp, k))
        b = P.focus_intensity(_ops.pack_amp_opd(amp, opd), Q, synth=('packed', k))
        assert torch.equal(a, b), (n, Q)
    ref = O.intensity(O.focus(O.from_amp_and_phase(amp.cpu().numpy().astype(np.float64), opd.cpu().numpy().astype(np.float64), 0.55), 1))
    assert rel_max(tonp(b), ref) < 2e-5


@pytest.mark.parametrize('shape', [(96, 160), (1536, 1536), (2560, 1024), (448, 1536), (3072, 5120)])
@pytest.mark.parametrize('dtype', [np.complex64, np.complex128])
def test_mixed_radix_lengths_vs_numpy(pa, radix_r_route, shape, dtype):
    """lengths 3 / 5 / 7 x 2^k (Q = 1.5 pads, scipy's next_fast_len values) take one radix-R step around engine transforms instead of
    Bluestein's convolution at the next power of two above 2 n: same results as numpy, and as the Bluestein route (knob mixed_radix = 0)"""
    from prysm_amd import _lib
    P = pa.propagation
    rng = np.random.default_rng(shape[0] + shape[1])
    x = crandn_(rng, shape, dtype)
    xd = torch.from_numpy(x).cuda()
    tol = TOL32 if dtype == np.complex64 else TOL64
    want = O.focus(x.astype(np.complex128), 1)
    got = tonp(P.focus(xd, 1))
    assert got.dtype == dtype and rel_max(got, want) < tol
    # windows, crops and the inverse through the same path: unfocus of a padded field, and its adjoint (crop)
    if shape[0] <= 1536:
        m, n = (shape[0] * 2) // 3, (shape[1] * 2) // 3
        y = crandn_(rng, (m, n), dtype)
        assert rel_max(tonp(P.unfocus(y, 1.5)), O.unfocus(y.astype(np.complex128), 1.5)) < tol
        assert rel_max(tonp(P.focus_adjoint(xd, 1.5)), O.focus_adjoint(x.astype(np.complex128), 1.5)) < tol
        lib = _lib.load()
        try:
            lib.pm_set_tuning(b'mixed_radix', 0)
            ref = tonp(P.focus(xd, 1))
        finally:
            lib.pm_set_tuning(b'mixed_radix', 1)
        assert rel_max(got, ref) < tol


def crandn_(rng, shape, dtype):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)


# ----------------------------------------------------------------------------- reference identities

def test_array_orientation_consistency_tilt(pa):
    """arr[y, x]: a positive +y tilt in the pupil moves the PSF to +y and leaves x centred (tests/test_physics.py:56-74)"""
    P = pa.propagation
    N, wvl, Q = 128, .5, 3
    x, y = O.make_xy_grid(N, diameter=2.1)
    r, _ = O.cart_to_polar(x, y)
    amp = O.circle(1, r)
    phs = 1000 * y
    for dt in (np.float64, np.float32):
        wf = P.Wavefront.from_amp_and_phase(amp.astype(dt), phs.astype(dt), wvl, x[0, 1] - x[0, 0])
        psf = tonp(wf.focus(1, Q=Q).intensity)
        idx_y, idx_x = np.unravel_index(psf.argmax(), psf.shape)
        assert idx_x == (N * Q) // 2
        assert idx_y > (N * Q) // 2
        ref = O.intensity(O.focus(O.from_amp_and_phase(amp, phs, wvl), Q))
        assert rel_max(psf, ref) < (TOL64 if dt == np.float64 else 2e-5)


def test_thinlens_hopkins_agree(pa):
    """a weak thin lens in front of the pupil == the matching Hopkins defocus (tests/test_propagation.py:440-460)"""
    P = pa.propagation
    x, y = O.make_xy_grid(128, diameter=11)
    dx = x[0, 1] - x[0, 0]
    r = np.hypot(x, y)
    amp = O.circle(5, r)
    phs = (r / 5) ** 2 * (1.975347661 * O.HeNe * 1000)     # hopkins(0, 2, 0, rho, 0, 1) = rho^2
    psf = tonp(P.Wavefront.from_amp_and_phase(amp, phs, O.HeNe, dx).focus(efl=100, Q=2).intensity)
    no_phs_wf = P.Wavefront.from_amp_and_phase(amp, None, O.HeNe, dx)
    tl = P.Wavefront.thin_lens(10_000, O.HeNe, x, y)
    psf2 = tonp((no_phs_wf * tl).focus(efl=100, Q=2).intensity)
    assert np.allclose(psf, psf2, rtol=1e-5)
    ref = O.intensity(O.focus(O.from_amp_and_phase(amp, phs, O.HeNe), 2))
    assert rel_max(psf, ref) < TOL64


def test_to_fpm_and_back_adjoint_returns_fpm_gradient(pa):
    """Wavefront.to_fpm_and_back_adjoint(..., return_fpm_grad=True, field_at_fpm=...) against a central difference
    (tests/test_propagation.py:353-381); ADVICE r1: the keywords were missing from the object API"""
    P = pa.propagation
    rng = np.random.default_rng(123)
    z = rng.normal(size=(8, 8)) + 1j * rng.normal(size=(8, 8))
    wf = P.Wavefront(cmplx_field=z, dx=1.0, wavelength=O.HeNe, space='pupil')
    fpm_data = rng.normal(size=(8, 8))
    fpm = P.Wavefront(cmplx_field=fpm_data, dx=0.1, wavelength=O.HeNe, space='psf')
    mdft = wf.prepare_executor(efl=10.0, dx=fpm.dx, samples=fpm.data.shape)
    out, at_fpm, _ = wf.to_fpm_and_back(fpm=fpm, executor=mdft, return_more=True)
    outbar_data = rng.normal(size=(8, 8)) + 1j * rng.normal(size=(8, 8))
    outbar = P.Wavefront(cmplx_field=outbar_data, dx=out.dx, wavelength=O.HeNe, space=out.space)
    abar, fpm_bar = outbar.to_fpm_and_back_adjoint(fpm=fpm, executor=mdft, return_fpm_grad=True, field_at_fpm=at_fpm)
    assert fpm_bar.space == 'psf' and fpm_bar.dx == mdft.focal_dx and abar.space == 'pupil'
    more = outbar.to_fpm_and_back_adjoint(fpm=fpm, executor=mdft, return_more=True, return_fpm_grad=True, field_at_fpm=at_fpm)
    assert len(more) == 4 and [w.space for w in more] == ['pupil', 'psf', 'psf', 'psf']
    assert rel_max(tonp(more[0]), tonp(abar)) < 1e-14 and rel_max(tonp(more[3]), tonp(fpm_bar)) < 1e-14
    yy, xx, eps = 3, 4, 1e-6
    fp, fm = fpm_data.copy(), fpm_data.copy()
    fp[yy, xx] += eps
    fm[yy, xx] -= eps
    j_plus = _real_vdot(outbar_data, tonp(wf.to_fpm_and_back(fpm=fp, executor=mdft)))
    j_minus = _real_vdot(outbar_data, tonp(wf.to_fpm_and_back(fpm=fm, executor=mdft)))
    fd = (j_plus - j_minus) / (2 * eps)
    assert float(np.real(tonp(fpm_bar)[yy, xx])) == pytest.approx(fd, rel=1e-6, abs=1e-8)


def test_babinet_adjoint_returns_fpm_and_lyot_gradients(pa):
    """tests/test_propagation.py:384-427 through the Wavefront API"""
    P = pa.propagation
    rng = np.random.default_rng(456)
    z = rng.normal(size=(8, 8)) + 1j * rng.normal(size=(8, 8))
    wf = P.Wavefront(cmplx_field=z, dx=1.0, wavelength=O.HeNe, space='pupil')
    fpm_data = rng.normal(size=(8, 8))
    lyot_data = rng.normal(size=(8, 8))
    fpm = P.Wavefront(cmplx_field=fpm_data, dx=0.1, wavelength=O.HeNe, space='psf')
    lyot = P.Wavefront(cmplx_field=lyot_data, dx=1.0, wavelength=O.HeNe, space='pupil')
    mdft = wf.prepare_executor(efl=10.0, dx=fpm.dx, samples=fpm.data.shape)
    out, at_fpm, _, at_lyot = wf.babinet(lyot=lyot, fpm=fpm, executor=mdft, return_more=True)
    outbar_data = rng.normal(size=(8, 8)) + 1j * rng.normal(size=(8, 8))
    outbar = P.Wavefront(cmplx_field=outbar_data, dx=out.dx, wavelength=O.HeNe, space=out.space)
    abar, fpm_bar, lyot_bar = outbar.babinet_adjoint(lyot=lyot, fpm=fpm, executor=mdft, field_at_fpm=at_fpm, field_at_lyot=at_lyot,
                                                     return_fpm_grad=True, return_lyot_grad=True)
    assert (abar.space, fpm_bar.space, lyot_bar.space) == ('pupil', 'psf', 'pupil') and fpm_bar.dx == mdft.focal_dx
    # the plain call still returns one wavefront, equal to the oracle's adjoint
    plain = outbar.babinet_adjoint(lyot=lyot, fpm=fpm, executor=mdft)
    ex = O.prepare_executor(1.0, (8, 8), 0.1, (8, 8), O.HeNe, 10.0)
    assert rel_max(tonp(plain), O.babinet_adjoint(outbar_data, lyot_data, fpm_data, ex)) < TOL64
    eps = 1e-6

    def J(f, l):
        return _real_vdot(outbar_data, tonp(wf.babinet(lyot=l, fpm=f, executor=mdft)))

    fy, fx = 2, 5
    fp, fm = fpm_data.copy(), fpm_data.copy()
    fp[fy, fx] += eps
    fm[fy, fx] -= eps
    fd_fpm = (J(fp, lyot_data) - J(fm, lyot_data)) / (2 * eps)
    ly, lx = 6, 1
    lp, lm = lyot_data.copy(), lyot_data.copy()
    lp[ly, lx] += eps
    lm[ly, lx] -= eps
    fd_lyot = (J(fpm_data, lp) - J(fpm_data, lm)) / (2 * eps)
    assert float(np.real(tonp(fpm_bar)[fy, fx])) == pytest.approx(fd_fpm, rel=1e-6, abs=1e-8)
    assert float(np.real(tonp(lyot_bar)[ly, lx])) == pytest.approx(fd_lyot, rel=1e-6, abs=1e-8)


def test_multiresolution_fpm_grad_matches_fd(pa):
    """tests/test_propagation.py:589-622"""
    P = pa.propagation
    rng = np.random.default_rng(20260704)
    npup = 16
    executor = P.prepare_multiresolution(pupil_dx=0.25, pupil_samples=npup, focal_dx=4.0, focal_samples=16, wavelength=O.HeNe,
                                         efl=10.0, num_levels=2, fine_samples=12)
    fpm = P.vortex_phase_mask(2)
    x = rng.standard_normal((npup, npup)) + 1j * rng.standard_normal((npup, npup))
    out, at_fpm, after_fpm = P.to_fpm_and_back_multiresolution(x, fpm, executor, return_more=True)
    assert len(at_fpm) == len(after_fpm) == len(executor)
    ybar = rng.standard_normal((npup, npup)) + 1j * rng.standard_normal((npup, npup))
    _, fpm_bars = P.to_fpm_and_back_multiresolution_adjoint(ybar, fpm, executor, return_fpm_grad=True, field_at_fpm=at_fpm)
    k, iy, ix = 1, 3, 5
    x0 = float(tonp(executor.xf[k])[iy, ix])
    y0 = float(tonp(executor.yf[k])[iy, ix])
    eps = 1e-6

    def bumped(sign):
        def f(xf, yf):
            return fpm(xf, yf) + sign * eps * ((xf == x0) & (yf == y0))
        return f

    j_plus = _real_vdot(ybar, tonp(P.to_fpm_and_back_multiresolution(x, bumped(+1), executor)))
    j_minus = _real_vdot(ybar, tonp(P.to_fpm_and_back_multiresolution(x, bumped(-1), executor)))
    fd = (j_plus - j_minus) / (2 * eps)
    assert float(np.real(tonp(fpm_bars[k])[iy, ix])) == pytest.approx(fd, rel=1e-6, abs=1e-8)


def test_prepare_executor_grid_bases_equal_vector_bases(pa):
    """prepare_executor(kind='mdft') generates the bases from the grid parameters (pm_mdft_basis_grid); MDFT(x, y, fx, fy) of the
    coordinate vectors is the reference construction (prysm/propagation/dft.py:97-105, fttools.py:187-191): identical matrices"""
    from prysm_amd.conf import config
    from prysm_amd.fttools import MDFT
    P = pa.propagation
    prec = config.precision
    try:
        for precision in (32, 64):
            config.precision = precision
            for args in ((10 / 2048, (2048, 2048), 0.6328 * 10 / 8, (512, 512), 0.6328, 100.0, (0, 0)),
                         (0.05, (96, 130), 0.7, (33, 64), 0.55, 80.0, (0.35, -1.25))):
                ex = P.prepare_executor(*args[:6], focal_shift=args[6])
                ref = MDFT(*P.coordinates_for_focus(*args[:6], focal_shift=args[6]), sign=-1, norm=ex.norm)
                assert ex.Ex.dtype == ref.Ex.dtype and torch.equal(ex.Ex, ref.Ex) and torch.equal(ex.Ey, ref.Ey)
                assert (ex._forward_left_first, ex._adjoint_left_first) == (ref._forward_left_first, ref._adjoint_left_first)
                assert ex.pupil_dx == args[0] and ex.focal_dx == args[2]
    finally:
        config.precision = prec


# ----------------------------------------------------------------------------- real-input (Hermitian) transforms

def _np_transform_psf(psf):
    return np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(psf.astype(np.float64))))


@pytest.mark.parametrize('shape', [(32, 32), (64, 256), (256, 64), (2, 32), (512, 512), (1024, 2048), (4096, 4096)])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_real_input_hermitian_transform_vs_numpy(pa, shape, dtype):
    """transform_psf of a real array (prysm/otf.py:28-33): the library computes N/2 + 1 columns and stores every bin twice; the
    complex path (knob r2c = 0) must agree with it to rounding and both with numpy"""
    from prysm_amd import _lib, otf
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    psf = rng.random(shape).astype(dtype) + 0.01
    want = _np_transform_psf(psf)
    tol = TOL32 if dtype == np.float32 else TOL64
    lib = _lib.load()
    try:
        lib.pm_set_tuning(b'r2c', 2)      # the Hermitian path also for the plain complex spectrum (by default only where it pays)
        for fold in (-1, 1, 0):           # auto; the radix-2 step of the column transform folded into the row pass; never
            lib.pm_set_tuning(b'fold', fold)
            got, df = otf.transform_psf(psf, 0.5)
            assert tonp(got).dtype == (np.complex64 if dtype == np.float32 else np.complex128)
            assert rel_max(tonp(got), want) < tol, fold
        lib.pm_set_tuning(b'r2c', 0)
        ref, _ = otf.transform_psf(psf, 0.5)
    finally:
        lib.pm_set_tuning(b'r2c', 1)
        lib.pm_set_tuning(b'fold', -1)
    assert rel_max(tonp(got), tonp(ref)) < tol
    assert df == pytest.approx(1000 / (shape[0] * 0.5))


@pytest.mark.parametrize('shape', [(64, 64), (128, 512), (2048, 2048)])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_fused_mtf_ptf_otf_vs_numpy(pa, shape, dtype):
    """mtf_from_psf / ptf_from_psf / otf_from_psf with the centre normalisation and abs / angle in the column pass's epilogue
    (prysm/otf.py:62-164), against the reference expressions in numpy and against the composed (return_more) route"""
    from prysm_amd import otf
    rng = np.random.default_rng(shape[1])
    y, x = np.indices(shape)
    psf = (np.exp(-((y - shape[0] // 2 - 1.5) ** 2 + (x - shape[1] // 2 + 2.25) ** 2) / 40.0) + 0.05 * rng.random(shape)).astype(dtype)
    F = _np_transform_psf(psf)
    norm = F / F[shape[0] // 2, shape[1] // 2]
    tol = TOL32 if dtype == np.float32 else TOL64
    from prysm_amd import _lib
    lib = _lib.load()
    try:
        for fold in (1, 0, -1):
            lib.pm_set_tuning(b'fold', fold)
            mtf = tonp(otf.mtf_from_psf(psf, 0.5))
            assert mtf.dtype == dtype and rel_max(mtf, np.abs(norm)) < tol, fold
            o = tonp(otf.otf_from_psf(psf, 0.5))
            assert rel_max(o, norm) < tol, fold
    finally:
        lib.pm_set_tuning(b'fold', -1)
    o = tonp(otf.otf_from_psf(psf, 0.5))
    assert rel_max(o, norm) < tol
    ptf = tonp(otf.ptf_from_psf(psf, 0.5))
    # the phase is ill-conditioned where the modulus vanishes: compare where |OTF| is well above the rounding floor
    ok = np.abs(norm) > (1e-3 if dtype == np.float32 else 1e-8)
    dphi = np.angle(np.exp(1j * (ptf - np.angle(norm))))
    assert np.abs(dphi[ok]).max() < (2e-3 if dtype == np.float32 else 1e-6)
    # composed route (also returns the unnormalised transform)
    mtf2, data = otf.mtf_from_psf(psf, 0.5, return_more=True)
    assert rel_max(tonp(mtf2), mtf) < tol and rel_max(tonp(data), F) < tol
    # a negative DC flips the sign of the normalised transform: abs unchanged, phase by pi
    neg = tonp(otf.otf_from_psf(-psf, 0.5))
    assert rel_max(neg, norm) < tol


def test_hermitian_epilogues_refused_elsewhere(pa):
    """PM_EPI_ABS / PM_FLAG_NORM_DC exist on the Hermitian path only: complex input or awkward lengths fall back to the composed
    route in otf.py, and the C ABI says PM_ERR_UNSUPPORTED"""
    from prysm_amd import _lib as L, _ops, otf
    rng = np.random.default_rng(5)
    psf = rng.random((48, 100))            # not powers of two: composed route
    F = _np_transform_psf(psf)
    assert rel_max(tonp(otf.mtf_from_psf(psf, 1.0)), np.abs(F / F[24, 50])) < TOL64
    z = torch.randn(64, 64, dtype=torch.complex64, device='cuda')
    with pytest.raises(NotImplementedError):
        _ops.fft2(z, direction=-1, scale=1.0, epilogue=L.PM_EPI_ABS)
    with pytest.raises(NotImplementedError):
        _ops.fft2(z.real.contiguous(), direction=-1, scale=1.0, shape=(128, 128), flags=L.PM_FLAG_NORM_DC)     # padded input


@pytest.mark.parametrize('mode', ['edge', 'reflect', 'symmetric', 'wrap'])
def test_pad2d_modes_match_numpy_pad(pa, mode):
    """fttools.pad2d(mode != 'constant') forwards to np.pad with widths (d - d // 2, d // 2) (prysm/fttools.py:79-98)"""
    from prysm_amd import fttools
    rng = np.random.default_rng(11)
    for shape, out_shape in (((9, 12), (14, 18)), ((5, 4), (17, 21)), ((1, 6), (4, 6)), ((8, 8), (8, 8))):
        for dt in (np.float32, np.complex128, np.bool_):
            a = (rng.random(shape) > 0.5) if dt is np.bool_ else rng.standard_normal(shape).astype(dt)
            diff = [o - i for o, i in zip(out_shape, shape)]
            want = np.pad(a, [(d - d // 2, d // 2) for d in diff], mode=mode)
            got = tonp(fttools.pad2d(a, out_shape=out_shape, mode=mode))
            assert got.dtype == want.dtype and np.array_equal(got, want), (shape, out_shape, dt)
    with pytest.raises(NotImplementedError):
        fttools.pad2d(np.ones((4, 4)), Q=2, mode='no_such_mode')


@pytest.mark.parametrize('mode', ['mean', 'maximum', 'minimum', 'median', 'linear_ramp'])
def test_pad2d_statistical_modes_match_numpy_pad(pa, mode):
    """np.pad's statistical modes and linear_ramp at their defaults (statistics over the whole axis, end value 0), axis by axis"""
    from prysm_amd import fttools
    rng = np.random.default_rng(12)
    for shape, out_shape in (((9, 12), (14, 18)), ((5, 4), (17, 21)), ((1, 6), (4, 6)), ((8, 8), (8, 8)), ((6, 7), (6, 12))):
        for dt in (np.float32, np.float64) + ((np.complex128,) if mode in ('mean', 'linear_ramp') else ()):
            a = rng.standard_normal(shape).astype(dt)
            if dt is np.complex128:
                a = a + 1j * rng.standard_normal(shape)
            diff = [o - i for o, i in zip(out_shape, shape)]
            want = np.pad(a, [(d - d // 2, d // 2) for d in diff], mode=mode)
            got = tonp(fttools.pad2d(a, out_shape=out_shape, mode=mode))
            assert got.dtype == want.dtype and got.shape == want.shape
            assert np.allclose(got, want, rtol=1e-6 if dt is np.float32 else 1e-13, atol=1e-6 if dt is np.float32 else 1e-13), (shape, out_shape, dt)


# ----------------------------------------------------------------------------- ADVICE r1

def test_babinet_takes_a_boolean_occulter(pa):
    """geometry.circle masks are boolean; numpy's `1 - bool_array` works in the reference (coronagraph.py:339)"""
    P = pa.propagation
    rng = np.random.default_rng(9)
    n = 32
    x, y = O.make_xy_grid(n, diameter=8)
    r, _ = O.cart_to_polar(x, y)
    field = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) * O.circle(4, r)
    occ = O.circle(1.5, r)       # bool
    lyot = O.circle(3.5, r)      # bool
    assert occ.dtype == bool
    ex = P.prepare_executor(8 / n, (n, n), 1.0, (n, n), O.HeNe, 20.0)
    exo = O.prepare_executor(8 / n, (n, n), 1.0, (n, n), O.HeNe, 20.0)
    got = tonp(P.babinet(field, lyot, occ, ex))
    assert rel_max(got, O.babinet(field, lyot, occ, exo)) < TOL64
    g = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    gota = tonp(P.babinet_adjoint(g, lyot, occ, ex))
    assert rel_max(gota, O.babinet_adjoint(g, lyot.astype(float), occ, exo)) < TOL64


def test_focus_intensity_rejects_a_mismatched_accumulator(pa):
    P = pa.propagation
    x = torch.randn(64, 64, dtype=torch.complex64, device='cuda')
    good = torch.zeros(128, 128, dtype=torch.float32, device='cuda')
    P.focus_intensity(x, 2, out=good, weight=0.5)
    for bad in (torch.zeros(128, 128, dtype=torch.float64, device='cuda'),       # dtype of another precision
                torch.zeros(64, 64, dtype=torch.float32, device='cuda'),         # shape of another Q
                torch.zeros(128, 256, dtype=torch.float32, device='cuda')[:, ::2],   # last axis not contiguous
                torch.zeros(128, 128, dtype=torch.float32)):                     # host tensor
        with pytest.raises(ValueError):
            P.focus_intensity(x, 2, out=bad, weight=0.5)
    # a row-strided view is fine (only the leading dimension is read)
    wide = torch.zeros(128, 160, dtype=torch.float32, device='cuda')
    P.focus_intensity(x, 2, out=wide[:, :128], weight=0.5)
    assert rel_max(tonp(wide[:, :128]), tonp(good)) < 1e-6 and float(wide[:, 128:].abs().max()) == 0.0


def test_jones_adapter_passes_stacks_through(pa):
    from prysm_amd.x import polarization as pol
    rng = np.random.default_rng(3)
    wrapped = pol.jones_adapter(pa.propagation.focus)
    st = (rng.standard_normal((3, 32, 32)) + 1j * rng.standard_normal((3, 32, 32)))
    got = tonp(wrapped(st, 2))
    assert got.shape == (3, 64, 64)
    for b in range(3):
        assert rel_max(got[b], O.focus(st[b], 2)) < TOL64
    J = rng.standard_normal((16, 16, 2, 2)) + 1j * rng.standard_normal((16, 16, 2, 2))
    gj = tonp(wrapped(J, 2))
    assert gj.shape == (32, 32, 2, 2)
    assert rel_max(gj[..., 1, 0], O.focus(J[..., 1, 0], 2)) < TOL64


# ----------------------------------------------------------------------------- N > 1

def _two_rank_backend():
    return 'nccl' if torch.cuda.device_count() >= 2 else 'gloo'


def _env():
    env = dict(os.environ)
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    return env


def test_bench_two_ranks_selflaunch(pa):
    """`python bench.py --gpus 2` with no launcher: bench.py re-executes itself under torch.distributed.run, one rank per GPU
    over RCCL when two GPUs are visible (both ranks on GPU 0 over gloo otherwise) and prints ONE line with n_gpus = 2 and a
    timed config-5 run."""
    be = _two_rank_backend()
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2', '--edge', '1024',
           '--no-cpu-baseline', '--backend', be]
    env = _env()
    env.pop('WORLD_SIZE', None)
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['scaling'] == 'weak'
    poly = line['polychromatic']
    assert poly['wavelengths_per_gpu'] == 32
    assert poly['variant_F_fft_focus']['psf_ms'] > 0 and poly['variant_M_mdft_512']['psf_ms'] > 0
    assert line['n2048']['value'] > 0 and line['reduce_ms'] > 0


def test_polychromatic_two_ranks_vs_oracle(pa):
    """tests/multi_rank_poly.py: stacks, field-by-field and matrix-DFT variants sharded over two ranks, reduce and the
    all-to-all reduce, against the oracle's single-process sum"""
    be = _two_rank_backend()
    env = _env()
    env['PM_TEST_BACKEND'] = be
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29541', os.path.join(ROOT, 'tests', 'multi_rank_poly.py')]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
    assert res.stdout.count('OK') >= 2


def test_polychromatic_one_rank_rccl_group(pa):
    """the same script as ONE rank under an `nccl` process group -- RCCL initialises and reduce,
    all_to_all_single, gather and all_reduce execute on the device for real (a group of one rank still runs its collective,
    polychromatic._group_info), plus the pipelined form with its side stream"""
    env = _env()
    env['PM_TEST_BACKEND'] = 'nccl'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29543', os.path.join(ROOT, 'tests', 'multi_rank_poly.py')]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
    assert res.stdout.count('OK') >= 1


# ----------------------------------------------------------------------------- the wavelength loop as one call (pm_fft2_spectral)

def _spectral_case(rng, m, n):
    amp = torch.from_numpy((rng.random((m, n)) > 0.25).astype(np.float32)).cuda()
    opd = torch.from_numpy((200 * rng.standard_normal((m, n))).astype(np.float32)).cuda()
    return amp, opd


@pytest.mark.parametrize('m,n,Q,count', [(64, 64, 1, 3), (256, 256, 1, 11), (256, 512, 1, 8), (256, 256, 2, 5), (1024, 1024, 1, 9),
                                         (2048, 2048, 1, 4), (32, 2048, 1, 2), (64, 2048, 1, 3), (4096, 2048, 1, 3)])
def test_spectral_call_equals_the_wavelength_loop(pa, m, n, Q, count):
    """pm_fft2_spectral (groups of wavelengths per launch pair: packed map read once, w |.|^2 summed in registers) against the
    loop it replaces -- one accumulate-epilogue transform pair per wavelength -- and against the fp64 oracle sum; every group size
    and both register / memory forms of its two kernels (tuning keys spectral, spectral_mode)."""
    from prysm_amd import _lib, _ops
    P = pa.propagation
    lib = _lib.load()
    rng = np.random.default_rng(m * 7 + n + count)
    amp, opd = _spectral_case(rng, m, n)
    packed = _ops.pack_amp_opd(amp, opd)
    wvls = np.linspace(0.5, 0.7, count)
    ks = [2 * np.pi / w / 1e3 for w in wvls]
    wts = list(np.linspace(0.5, 1.5, count))
    M, N = int(m * Q), int(n * Q)
    loop = torch.zeros((M, N), device='cuda')
    for k, w in zip(ks, wts):
        P.focus_intensity(packed, Q, out=loop, weight=w, synth=('packed', k))
    a64, o64 = amp.cpu().numpy().astype(np.float64), opd.cpu().numpy().astype(np.float64)
    want = sum(w * O.intensity(O.focus(O.from_amp_and_phase(a64, o64, float(wl)), Q)) for wl, w in zip(wvls, wts))
    assert rel_max(tonp(loop), want) < 2e-5
    try:
        if n == 2048 and m <= 64:
            lib.pm_set_tuning(b'fold', 1)      # the folded kernels (automatic from 4096 rows) on short columns too
        for group in (1, 2, 3, 8):
            for mode in (0, 1, 2, 3):
                assert lib.pm_set_tuning(b'spectral', group) == 0
                if lib.pm_set_tuning(b'spectral_mode', mode) != 0:     # forms 0 - 2 lost their measurements and left the library (experiments/README.md)
                    assert mode != 3
                    continue
                got = torch.full((M, N), 0.0, device='cuda')
                P.focus_intensity(packed, Q, out=got, synth=('packed', ks[0]), spectral=(ks, wts))
                # same terms in the same order; only the association of the fp32 sum differs (per group: acc + (w0 i0 + w1 i1 ...))
                assert rel_max(tonp(got), tonp(loop)) < 2e-6, (group, mode)
                assert rel_max(tonp(got), want) < 2e-5, (group, mode)
    finally:
        lib.pm_set_tuning(b'spectral', 8)
        lib.pm_set_tuning(b'spectral_mode', 3)
        lib.pm_set_tuning(b'fold', -1)


def test_spectral_call_accumulates_and_falls_back(pa):
    """the call ADDS to its accumulator; descriptors outside the fused form (two-array synthesis; 4096^2 bins, where the loop is as
    fast) run the plain loop and give the same image"""
    from prysm_amd import _lib, _ops
    P = pa.propagation
    lib = _lib.load()
    rng = np.random.default_rng(5)
    amp, opd = _spectral_case(rng, 256, 256)
    packed = _ops.pack_amp_opd(amp, opd)
    ks = [2 * np.pi / w / 1e3 for w in (0.5, 0.6, 0.7)]
    wts = [1.0, 2.0, 0.5]
    base = torch.rand((256, 256), device='cuda')
    once = torch.zeros((256, 256), device='cuda')
    P.focus_intensity(packed, 1, out=once, synth=('packed', ks[0]), spectral=(ks, wts))
    got = base.clone()
    P.focus_intensity(packed, 1, out=got, synth=('packed', ks[0]), spectral=(ks, wts))
    assert rel_max(tonp(got), tonp(base + once)) < 1e-6
    two = torch.zeros((256, 256), device='cuda')
    P.focus_intensity(opd, 1, out=two, synth=(amp, ks[0]), spectral=(ks, wts))     # amplitude and OPD as two arrays: the loop
    assert rel_max(tonp(two), tonp(once)) < 2e-6
    try:
        assert lib.pm_set_tuning(b'spectral_area_log', 10) == 0     # 256^2 = 2^16 bins >= 2^10: the loop
        loop = torch.zeros((256, 256), device='cuda')
        P.focus_intensity(packed, 1, out=loop, synth=('packed', ks[0]), spectral=(ks, wts))
    finally:
        lib.pm_set_tuning(b'spectral_area_log', 24)
    assert rel_max(tonp(loop), tonp(once)) < 2e-6
    with pytest.raises(ValueError):
        P.focus_intensity(packed, 1, out=once, synth=('packed', ks[0]), spectral=(ks, wts[:2]))


@pytest.mark.parametrize('n,Q', [(512, 2), (1024, 1)])
def test_polychromatic_psf_spectral_equals_loop_and_oracle(pa, n, Q):
    """the driver's default below 4096^2 transforms (one pm_fft2_spectral call per rank) against its per-wavelength loop
    (spectral=False), the stacked form (batched=True) and the oracle sum"""
    from prysm_amd.polychromatic import polychromatic_psf
    x, y = O.make_xy_grid(n, diameter=10)
    r, _ = O.cart_to_polar(x, y)
    amp = O.circle(5, r).astype(np.float32)
    opd = O.hopkins_w040(r / 5, 400.0).astype(np.float32)
    dx = float(x[0, 1] - x[0, 0])
    wvls = np.linspace(0.5, 0.7, 13)
    wts = np.linspace(1.0, 2.0, 13)
    want = sum(w * O.intensity(O.focus(O.from_amp_and_phase(amp.astype(np.float64), opd.astype(np.float64), float(wl)), Q))
               for wl, w in zip(wvls, wts))
    fused = tonp(polychromatic_psf(amp, opd, wvls, wts, dx, 100.0, Q=Q))
    loop = tonp(polychromatic_psf(amp, opd, wvls, wts, dx, 100.0, Q=Q, spectral=False, batched=False))
    stacks = tonp(polychromatic_psf(amp, opd, wvls, wts, dx, 100.0, Q=Q, batched=True))
    assert fused.dtype == np.float32 and fused.shape == want.shape
    assert rel_max(fused, want) < 2e-5
    assert rel_max(fused, loop) < 2e-6
    assert rel_max(fused, stacks) < 2e-6


# ----------------------------------------------------------------------------- 1-D transforms of 16384 / 32768 and 3 / 5 / 7 x 2^k points

@pytest.mark.parametrize('cdtype', [np.complex64, np.complex128])
@pytest.mark.parametrize('n', [96, 1536, 2560, 3584, 16384, 20480])
def test_fft1_radix_r_lengths_vs_numpy(pa, radix_r_route, n, cdtype):
    """pm_fft1 at the lengths that used to take Bluestein's detour (mixed radix) or the O(n^2) kernel (above 8192): both axes, both
    directions, zero padded inputs (numpy's fft(x, n)) and cropped outputs, odd and even batch extents"""
    from prysm_amd import _ops
    rng = np.random.default_rng(n)
    tol = 2e-5 if cdtype == np.complex64 else 1e-10
    for batch, length, in_off in ((6, n, 0), (5, n - n // 3, 0), (8, n // 2 + 1, n // 4)):
        x = (rng.standard_normal((batch, length)) + 1j * rng.standard_normal((batch, length))).astype(cdtype)
        for axis in (1, 0):
            xa = x if axis == 1 else np.ascontiguousarray(x.T)
            xd = torch.from_numpy(xa).cuda()
            padded = np.zeros((batch, n), dtype=np.complex128)
            padded[:, in_off:in_off + length] = x
            for direction in (-1, +1):
                want = np.fft.fft(padded, axis=1) if direction < 0 else np.fft.ifft(padded, axis=1) * n
                got = tonp(_ops.fft1(xd, n, axis=axis, direction=direction, in_off=in_off))
                got = got if axis == 1 else got.T
                assert rel_max(got, want) < tol, (batch, length, in_off, axis, direction)
            # a window of the bins, with a scale
            lo, ln = n // 5, n // 3
            got = tonp(_ops.fft1(xd, n, axis=axis, direction=-1, in_off=in_off, out_off=lo, out_len=ln, scale=0.5))
            got = got if axis == 1 else got.T
            assert rel_max(got, 0.5 * np.fft.fft(padded, axis=1)[:, lo:lo + ln]) < tol, (batch, length, in_off, axis, 'window')


def test_czt_long_convolution_uses_fast_length(pa):
    """a chirp-Z axis whose convolution no longer fits one fused kernel (K > 8192) convolves at the next fast length -- 12288 = 3 x
    4096 for 6000 + 6000 - 1 points, where a power of two would be 16384 -- through pm_fft1's radix-R path; against the matrix DFT"""
    ft = pa.fttools
    assert ft.next_fast_len(2559) == 2560 and ft.next_fast_len(97) == 112 and ft.next_fast_len(4096) == 4096
    assert ft.next_fast_len(8703) == 10240 and ft.next_fast_len(20000) == 20480 and ft.next_fast_len(30000) == 32768
    rng = np.random.default_rng(3)
    nx, mx, ny, my = 6000, 6000, 16, 12
    r = lambda n: tonp(ft.fftrange(n)).astype(float)   # noqa: E731
    x, y = r(nx) * 0.2, r(ny) * 0.17
    fx, fy = (r(mx) + 0.25) / (nx * 0.2 * 1.3), (r(my) - 0.5) * 0.11
    inp = rng.standard_normal((ny, nx)) + 1j * rng.standard_normal((ny, nx))
    czt = ft.CZT(x, y, fx, fy)
    assert czt._Kx == 12288
    want = tonp(ft.MDFT(x, y, fx, fy)(inp))
    assert rel_max(tonp(czt(inp)), want) < 1e-8     # quadratic chirp phases of ~1e6 turns at 6000 points: ~1e-9 in fp64
    g = rng.standard_normal((my, mx)) + 1j * rng.standard_normal((my, mx))
    lhs = np.vdot(tonp(czt(inp)), g)
    rhs = np.vdot(inp, tonp(czt.adjoint(g)))
    assert abs(lhs - rhs) < 1e-8 * abs(lhs)


# ----------------------------------------------------------------------------- real object -> real result on half spectra (fft_c2r.h)

@pytest.mark.parametrize('shape,dtype', [((64, 64), np.float64), ((32, 128), np.float32), ((256, 64), np.float64), ((2, 64), np.float64),
                                         ((512, 2048), np.float32), ((2048, 1024), np.float64), ((4096, 4096), np.float32),
                                         ((16, 8192), np.float32)])
def test_real_convolution_on_half_spectra_vs_numpy(pa, shape, dtype):
    """conv / apply_transfer_functions of a REAL object keep the real part of ifft2(fft2(o) H) (prysm/convolution.py:29-31,110-113);
    PM_FLAG_REAL_OUTPUT runs the chain on half spectra: against numpy for a Hermitian H (a real PSF's transfer function), a general
    complex H (only its Hermitian part survives the real part), conj(H), centred and uncentred, and against the complex chain"""
    from prysm_amd import _lib, _ops
    rng = np.random.default_rng(shape[0] * 11 + shape[1])
    M, N = shape
    cdt = np.complex64 if dtype == np.float32 else np.complex128
    tol = 2e-5 if dtype == np.float32 else 1e-10
    o = rng.standard_normal(shape).astype(dtype)
    od = torch.from_numpy(o).cuda()
    psf = rng.random(shape).astype(dtype)
    H_real_psf = np.fft.fft2(psf.astype(np.float64))
    H_any = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    lib = _lib.load()
    assert lib.pm_set_tuning(b'r2c', 2) == 0       # also below 2048^2, where the library prefers the complex chain (it is as fast there)
    try:
        for H, conj in ((H_real_psf, False), (H_any, False), (H_any, True)):
            Hd = torch.from_numpy(H.astype(cdt)).cuda()
            for sh in ((0, 0), (M // 2, N // 2), (1 if M > 2 else 0, 0)):
                x = np.roll(o.astype(np.float64), (-sh[0], -sh[1]), axis=(0, 1))
                full = np.fft.ifft2(np.fft.fft2(x) * (np.conj(H) if conj else H))
                want = np.roll(full.real, sh, axis=(0, 1))
                got = _ops.fft2_mul_ifft2(od, scale=1.0 / (M * N), mul=Hd, mul_conj=conj, in_shift=sh, out_shift=sh, real_out=True)
                assert got.dtype == (torch.float32 if dtype == np.float32 else torch.float64) and not got.is_complex()
                assert got.is_contiguous()       # the half-spectrum chain (a `.real` view of the complex result would not be)
                assert rel_max(tonp(got), want) < tol, (conj, sh)
                cplx = _ops.fft2_mul_ifft2(od, scale=1.0 / (M * N), mul=Hd, mul_conj=conj, in_shift=sh, out_shift=sh)
                assert rel_max(tonp(got), tonp(cplx).real) < tol
    finally:
        lib.pm_set_tuning(b'r2c', 1)


def test_real_convolution_callers_and_fallback(pa):
    """convolution.conv / apply_transfer_functions take the half-spectrum chain for real power-of-two objects and fall back to the
    complex chain's real part elsewhere (odd sizes, rows under 64 samples, x rotations other than N/2)"""
    from prysm_amd import _ops, convolution as C
    rng = np.random.default_rng(8)
    for shape in ((128, 256), (100, 256), (32, 32), (9, 12)):
        o = rng.standard_normal(shape)
        psf = rng.random(shape)
        want = np.fft.fftshift(np.fft.ifft2(np.fft.fft2(np.fft.ifftshift(o)) * np.fft.fft2(np.fft.ifftshift(psf)))).real
        got = C.conv(o, psf)
        assert not got.is_complex() and rel_max(tonp(got), want) < 1e-10, shape
        tf = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
        want2 = np.fft.fftshift(np.fft.ifft2(np.fft.fft2(np.fft.ifftshift(o)) * tf)).real
        got2 = C.apply_transfer_functions(o, 1.0, [tf])
        assert not got2.is_complex() and rel_max(tonp(got2), want2) < 1e-10, shape
    o = rng.standard_normal((64, 128))
    H = torch.from_numpy(rng.standard_normal((64, 128)) + 1j * rng.standard_normal((64, 128))).cuda()
    got = _ops.fft2_mul_ifft2(torch.from_numpy(o).cuda(), scale=1.0, mul=H, in_shift=(0, 5), out_shift=(0, 5), real_out=True)   # x rotation by 5
    want = np.roll(np.fft.ifft2(np.fft.fft2(np.roll(o, (0, -5), axis=(0, 1))) * H.cpu().numpy()).real * o.size, (0, 5), axis=(0, 1))
    assert rel_max(tonp(got), want) < 1e-10


@pytest.mark.parametrize('shape,dtype', [((4, 4096), np.float64), ((64, 4096), np.float32), ((16, 8192), np.float32), ((4096, 4096), np.float64)])
def test_real_convolution_folded_form(pa, shape, dtype):
    """the half-spectrum chain with the radix-2 step of the column transforms folded into its first and last row pass (automatic from
    4096-row objects with rows of 4096 / 8192 samples; forced here on short columns too), against numpy and the unfolded form"""
    from prysm_amd import _lib, _ops
    lib = _lib.load()
    rng = np.random.default_rng(shape[0] + shape[1])
    M, N = shape
    cdt = np.complex64 if dtype == np.float32 else np.complex128
    tol = 2e-5 if dtype == np.float32 else 1e-10
    o = rng.standard_normal(shape).astype(dtype)
    od = torch.from_numpy(o).cuda()
    H = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    Hd = torch.from_numpy(H.astype(cdt)).cuda()
    try:
        lib.pm_set_tuning(b'r2c', 2)
        for sh in ((0, 0), (M // 2, N // 2), (0, N // 2)):
            x = np.roll(o.astype(np.float64), (-sh[0], -sh[1]), axis=(0, 1))
            want = np.roll(np.fft.ifft2(np.fft.fft2(x) * H).real, sh, axis=(0, 1))
            assert lib.pm_set_tuning(b'fold', 1) == 0
            got = _ops.fft2_mul_ifft2(od, scale=1.0 / (M * N), mul=Hd, in_shift=sh, out_shift=sh, real_out=True)
            assert got.is_contiguous() and rel_max(tonp(got), want) < tol, sh
            lib.pm_set_tuning(b'fold', 0)
            flat = _ops.fft2_mul_ifft2(od, scale=1.0 / (M * N), mul=Hd, in_shift=sh, out_shift=sh, real_out=True)
            assert rel_max(tonp(got), tonp(flat)) < tol
    finally:
        lib.pm_set_tuning(b'fold', -1)
        lib.pm_set_tuning(b'r2c', 1)


@pytest.mark.parametrize('prec', [32, 64])
def test_czt_executor_from_grid_parameters_equals_the_constructor(pa, prec):
    """prepare_executor(kind='czt') builds its chirps from the grid parameters (pm_czt_vectors, no coordinate vectors, no device reads);
    the result equals CZT(*coordinates_for_focus(...)) and the oracle's executor, forward and adjoint, with a focal shift"""
    from prysm_amd.conf import config
    from prysm_amd.propagation import dft
    P = pa.propagation
    ft = pa.fttools
    rng = np.random.default_rng(prec)
    old = config.precision
    try:
        config.precision = prec
        cdt = np.complex64 if prec == 32 else np.complex128
        tol = 3e-5 if prec == 32 else 1e-10
        for ps, fs, shift in (((64, 96), (40, 24), (0.0, 0.0)), ((128, 128), (64, 64), (1.3, -0.7)), ((33, 20), (12, 17), (0.2, 0.1))):
            args = (0.05, ps, 1.1, fs, O.HeNe, 100.0)
            ex = P.prepare_executor(*args, focal_shift=shift, kind='czt')
            ref = ft.CZT(*dft.coordinates_for_focus(*args, focal_shift=shift), sign=-1, norm=ex.norm)
            x = (rng.standard_normal(ps) + 1j * rng.standard_normal(ps)).astype(cdt)
            g = (rng.standard_normal(fs) + 1j * rng.standard_normal(fs)).astype(cdt)
            assert rel_max(tonp(ex(x)), tonp(ref(x))) < tol / 10
            assert rel_max(tonp(ex.adjoint(g)), tonp(ref.adjoint(g))) < tol / 10
            want = O.prepare_executor(*args, focal_shift=shift)(x.astype(np.complex128))
            assert rel_max(tonp(ex(x)), want) < tol
            assert ex.nbytes() == ref.nbytes()
    finally:
        config.precision = old


def test_conv_golden_fixture_on_half_spectra(pa):
    """the reference's own conv output (tests/golden/wavefront.npz, generated by importing prysm) through the half-spectrum chain
    (forced: a 64 x 64 object is below the size from which the library prefers it)"""
    from prysm_amd import _lib
    lib = _lib.load()
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'wavefront.npz'))
    try:
        assert lib.pm_set_tuning(b'r2c', 2) == 0
        out = pa.convolution.conv(g['conv_obj'], g['conv_psf'])
        assert not out.is_complex() and out.is_contiguous()
        assert rel_max(tonp(out), g['conv_out']) < TOL64
    finally:
        lib.pm_set_tuning(b'r2c', 1)


def test_fused_pupil_synthesis_complex128(pa):
    """PM_FLAG_SYNTH_INPUT for float64 maps (complex128 transforms: fp64 sincospi per sample inside the row pass): the lazy wavefront
    stays lazy, the result is bit for bit the separate synthesis kernel + transform, equals the oracle, bool / float32 / float64 / no
    amplitude, folded (4096 rows) and not, packed pairs too; and the polychromatic loop on float64 maps"""
    from prysm_amd import _ops
    from prysm_amd.polychromatic import polychromatic_psf
    P = pa.propagation
    rng = np.random.default_rng(15)
    for n, Q in ((256, 1), (128, 2), (4096, 1)):
        x, y = O.make_xy_grid(n, diameter=10)
        r, _ = O.cart_to_polar(x, y)
        opd = O.hopkins_w040(r / 5, 800.0) + 30 * rng.standard_normal((n, n))
        for amp in (O.circle(5, r), rng.random((n, n)).astype(np.float32), rng.random((n, n)), None):
            wf = P.Wavefront.from_amp_and_phase(amp, opd, 0.55, 10.0 / n)
            assert wf._fusable(Q) is not None
            got = wf.focus(100.0, Q)
            assert got.data.dtype == torch.complex128 and wf._data is None
            want = O.focus(O.from_amp_and_phase(np.ones((n, n)) if amp is None else amp, opd, 0.55), Q)
            assert rel_max(tonp(got), want) < TOL64
            inten = wf.focus_intensity(100.0, Q)
            assert rel_max(tonp(inten), O.intensity(want)) < 4 * TOL64
            if n <= 256:
                field = wf.data                                       # materialised by the separate kernel
                assert torch.equal(P.focus(field, Q), got.data)       # same arithmetic, different loads
                a_dev = None if amp is None else torch.from_numpy(np.asarray(amp)).cuda()
                pk = _ops.pack_amp_opd(a_dev, torch.from_numpy(opd).cuda())
                assert pk.dtype == torch.complex128
                assert torch.equal(P.focus_intensity(pk, Q, synth=('packed', 2 * np.pi / 0.55 / 1e3)), inten.data)
    n = 256
    x, y = O.make_xy_grid(n, diameter=10)
    r, _ = O.cart_to_polar(x, y)
    amp, opd = O.circle(5, r), O.hopkins_w040(r / 5, 300.0)
    wv, wt = np.linspace(0.5, 0.7, 5), np.linspace(1.0, 2.0, 5)
    want = sum(w * O.intensity(O.focus(O.from_amp_and_phase(amp, opd, float(l)), 2)) for l, w in zip(wv, wt))
    got = polychromatic_psf(amp, opd, wv, wt, 10.0 / n, 100.0, Q=2)
    assert got.dtype == torch.float64 and rel_max(tonp(got), want) < 4 * TOL64


@pytest.mark.parametrize('m,n,Q,count', [(64, 64, 1, 3), (256, 256, 1, 9), (256, 512, 1, 8), (128, 128, 2, 5), (1024, 1024, 1, 4),
                                         (2048, 2048, 1, 3), (64, 4096, 1, 2)])
def test_spectral_call_complex128(pa, m, n, Q, count):
    """the grouped wavelength kernels for float64 maps (complex128 transforms; rows of up to 2048 samples -- longer rows keep the loop):
    against the loop they replace and the fp64 oracle sum, every group size and kernel form"""
    from prysm_amd import _lib, _ops
    P = pa.propagation
    lib = _lib.load()
    rng = np.random.default_rng(m + 3 * n + count)
    amp = (rng.random((m, n)) > 0.25).astype(np.float64)
    opd = 200 * rng.standard_normal((m, n))
    packed = _ops.pack_amp_opd(torch.from_numpy(amp).cuda(), torch.from_numpy(opd).cuda())
    assert packed.dtype == torch.complex128
    wvls = np.linspace(0.5, 0.7, count)
    ks = [2 * np.pi / w / 1e3 for w in wvls]
    wts = list(np.linspace(0.5, 1.5, count))
    M, N = int(m * Q), int(n * Q)
    loop = torch.zeros((M, N), device='cuda', dtype=torch.float64)
    for k, w in zip(ks, wts):
        P.focus_intensity(packed, Q, out=loop, weight=w, synth=('packed', k))
    want = sum(w * O.intensity(O.focus(O.from_amp_and_phase(amp, opd, float(wl)), Q)) for wl, w in zip(wvls, wts))
    assert rel_max(tonp(loop), want) < 4 * TOL64
    try:
        for group in (1, 3, 8):
            for mode in (0, 1, 2, 3):
                assert lib.pm_set_tuning(b'spectral', group) == 0
                if lib.pm_set_tuning(b'spectral_mode', mode) != 0:
                    assert mode != 3
                    continue
                got = torch.zeros((M, N), device='cuda', dtype=torch.float64)
                P.focus_intensity(packed, Q, out=got, synth=('packed', ks[0]), spectral=(ks, wts))
                assert rel_max(tonp(got), tonp(loop)) < 1e-13, (group, mode)
                assert rel_max(tonp(got), want) < 4 * TOL64, (group, mode)
    finally:
        lib.pm_set_tuning(b'spectral', 8)
        lib.pm_set_tuning(b'spectral_mode', 3)
