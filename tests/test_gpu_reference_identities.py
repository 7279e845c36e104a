"""The identities the reference's OWN test-suite pins for the rows around the path, run on the HIP path: same checks and
tolerances as the reference tests cited per function, written against prysm_amd (arrays live on the device)."""
import numpy as np
import pytest
import torch

from oracle import prysm_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pa():
    import prysm_amd
    from prysm_amd import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return prysm_amd


def tonp(x):
    from prysm_amd.mathops import array_to_true_numpy
    if hasattr(x, 'data') and not isinstance(x, (np.ndarray, torch.Tensor)):
        x = x.data
    return array_to_true_numpy(x)


def gaussian_psf(n=14, sig=0.6, x0=0.8, y0=-0.4):
    """off-centre, narrow PSF: broad OTF whose modulus stays away from zero (reference tests/test_otf.py:43-53)."""
    c = np.arange(n) - n // 2
    xx, yy = np.meshgrid(c, c)
    return np.exp(-((xx - x0) ** 2 + (yy - y0) ** 2) / (2 * sig ** 2))


def test_mtf_ptf_otf_centre_values(pa):
    """tests/test_otf.py:16-40: the centre-normalised transform is exactly 1 / 0 / 1+0j at the centre sample."""
    x = O.forward_ft_unit(1 / 1e3, 128)
    dat = np.sin(np.meshgrid(x, x)[0])
    dx = x[1] - x[0]
    c = (64, 64)
    assert tonp(pa.otf.mtf_from_psf(dat, dx))[c] == 1
    assert tonp(pa.otf.ptf_from_psf(dat, dx))[c] == 0
    assert tonp(pa.otf.otf_from_psf(dat, dx))[c] == 1 + 0j


def test_encircled_energy_monotonic_and_bounded(pa):
    """tests/test_otf.py:102-111."""
    psf = gaussian_psf(n=64, sig=2.0, x0=0.0, y0=0.0)
    psf = psf / psf.sum()
    radii = np.array([2.0, 5.0, 10.0, 20.0, 40.0])
    ee = tonp(pa.otf.encircled_energy(psf, dx=1.0, radius=radii))
    assert np.all(np.diff(ee) > 0)
    assert ee[-1] <= 1.0 + 1e-6
    assert np.isclose(float(tonp(pa.otf.encircled_energy(psf, 1.0, 10.0))), ee[2])


@pytest.mark.parametrize('radius', [12.0, [6.0, 18.0, 35.0]])
def test_encircled_energy_adjoint_matches_finite_differences(pa, radius):
    """tests/test_otf.py:114-138: <psf_bar, v> against a central difference of the loss, and the cached-transform path."""
    rng = np.random.default_rng(3)
    psf = gaussian_psf()
    dx = 1.0
    v = rng.standard_normal(psf.shape)
    if np.isscalar(radius):
        ee_bar = rng.standard_normal()
        loss = lambda p: float(ee_bar * tonp(pa.otf.encircled_energy(p, dx, radius)))  # noqa: E731
    else:
        ee_bar = rng.standard_normal(len(radius))
        loss = lambda p: float(np.sum(ee_bar * tonp(pa.otf.encircled_energy(p, dx, radius))))  # noqa: E731
    psf_bar = tonp(pa.otf.encircled_energy_adjoint(ee_bar, psf, dx, radius))
    analytic = float(np.sum(psf_bar * v))
    eps = 1e-6
    fd = (loss(psf + eps * v) - loss(psf - eps * v)) / (2 * eps)
    assert np.allclose(analytic, fd, rtol=1e-5, atol=1e-7)
    _, data = pa.otf.encircled_energy(psf, dx, radius, return_more=True)
    cached = tonp(pa.otf.encircled_energy_adjoint(ee_bar, dx=dx, radius=radius, data=data))
    assert np.allclose(cached, psf_bar, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('which', ['mtf', 'ptf', 'otf'])
def test_from_psf_adjoints_match_finite_differences(pa, which):
    """tests/test_otf.py:70-99: the MTF / PTF / OTF adjoints against central differences of a linear loss."""
    rng = np.random.default_rng(1)
    psf = gaussian_psf()
    v = rng.standard_normal(psf.shape)
    if which == 'otf':
        bar = rng.standard_normal(psf.shape) + 1j * rng.standard_normal(psf.shape)
        fwd = lambda p: tonp(pa.otf.otf_from_psf(p, 1.0))                       # noqa: E731
        loss = lambda p: float(np.real(np.sum(np.conj(bar) * fwd(p))))          # noqa: E731
        adj = tonp(pa.otf.otf_from_psf_adjoint(bar, psf, 1.0))
    else:
        bar = rng.standard_normal(psf.shape)
        f = pa.otf.mtf_from_psf if which == 'mtf' else pa.otf.ptf_from_psf
        loss = lambda p: float(np.sum(bar * tonp(f(p, 1.0))))                   # noqa: E731
        adj = tonp((pa.otf.mtf_from_psf_adjoint if which == 'mtf' else pa.otf.ptf_from_psf_adjoint)(bar, psf, 1.0))
    eps = 1e-6
    fd = (loss(psf + eps * v) - loss(psf - eps * v)) / (2 * eps)
    assert np.allclose(float(np.sum(adj * v)), fd, rtol=1e-5, atol=1e-7)


def test_conv_with_centred_delta_is_identity_and_keeps_dtype(pa):
    """tests/test_convolution.py:10-17,50-58."""
    obj = np.arange(25, dtype=float).reshape(5, 5)
    psf = np.zeros_like(obj)
    psf[2, 2] = 1
    out = pa.convolution.conv(obj, psf)
    assert not out.is_complex()
    np.testing.assert_allclose(tonp(out), obj, atol=1e-12)
    cobj = np.arange(25).reshape(5, 5) * (1 + 1j)
    cpsf = np.zeros_like(cobj)
    cpsf[2, 2] = 1
    cout = pa.convolution.conv(cobj, cpsf)
    assert cout.is_complex()
    np.testing.assert_allclose(tonp(cout), cobj, atol=1e-12)


@pytest.mark.parametrize('shift', [False, True])
def test_apply_transfer_functions_identity(pa, shift):
    """tests/test_convolution.py:34-47."""
    obj = np.arange(16, dtype=float).reshape(4, 4)
    out = pa.convolution.apply_transfer_functions(obj, 1, [np.ones_like(obj)], shift=shift)
    np.testing.assert_allclose(tonp(out), obj, atol=1e-12)


def test_prepare_measured_fpm_recovers_its_own_grid_and_continues_as_a_vortex(pa):
    """tests/test_propagation.py:646-670: exact recovery on the measurement's grid, ideal vortex / scalar fill far away."""
    P = pa.propagation
    n, dx = 129, 0.4
    x, y = O.make_xy_grid(n, dx=dx)
    measurement = np.exp(1j * 2 * np.arctan2(y, x))
    for order in (1, 3):
        fpm = P.prepare_measured_fpm(measurement, dx, charge=2, order=order)
        np.testing.assert_allclose(tonp(fpm(x, y)), measurement, atol=1e-12)
        far = np.full((1, 1), 1e5)
        np.testing.assert_allclose(tonp(fpm(far, far)), np.exp(1j * 2 * np.arctan2(far, far)), atol=1e-12)
    ones = np.ones((65, 65), dtype=complex)
    far = np.full((1, 1), 1e3)
    assert tonp(P.prepare_measured_fpm(ones, 1.0, fill=0.0)(far, far))[0, 0] == 0.0


@pytest.mark.parametrize('zoom', [0.5, 2, (2, 3)])
def test_fourier_resample_preserves_a_constant_field(pa, zoom):
    """tests/test_fttools.py:230-235."""
    out = tonp(pa.fttools.fourier_resample(np.ones((8, 8)), zoom))
    np.testing.assert_allclose(out, 1, atol=1e-12)


def test_scalar_helpers(pa):
    """tests/test_propagation.py:17-21,269-282: sampling helpers and the Talbot / Fresnel numbers."""
    P = pa.propagation
    for dzeta in (1 / 128.0, 1 / 256.0, 11.123 / 128.0):
        psf_s = P.pupil_sample_to_psf_sample(dzeta, 128, 0.55, 10)
        assert P.psf_sample_to_pupil_sample(psf_s, 128, 0.55, 10) == pytest.approx(dzeta)
    wvl, a, z = 123.456, 987.654321, 5
    assert wvl / (1 - np.sqrt(1 - wvl ** 2 / a ** 2)) == pytest.approx(P.talbot_distance(a, wvl), abs=.1)
    assert P.fresnel_number(a, z, wvl) == (a ** 2 / (z * wvl))


def test_fftdft_rejects_incompatible_or_nonuniform_grids(pa):
    """tests/test_fttools.py:212-226 and tests/test_propagation.py:169-175: error classes and messages of FFTDFT."""
    n = 8
    x = y = np.arange(-(n // 2), n - n // 2).astype(float)
    fx = fy = x / n
    bad = fx.copy()
    bad[-1] += 0.01
    with pytest.raises(ValueError, match='uniformly spaced'):
        pa.fttools.FFTDFT(x, y, bad, fy)
    with pytest.raises(ValueError, match='not FFT-compatible'):
        pa.fttools.FFTDFT(x, y, fx * 1.1, fy)
    assert pa.fttools.FFTDFT(x, y, fx, fy).nbytes() == 4 * 8 * 16
    with pytest.raises(ValueError, match='not FFT-compatible'):
        pa.propagation.prepare_executor(pupil_dx=0.1, pupil_samples=32, focal_dx=1.0, focal_samples=32, wavelength=O.HeNe, efl=10.0,
                                        kind='fftdft')


def _grey_circle(radius, npup, dx, ss=16):
    """supersampled (anti-aliased) circular aperture (reference tests/test_propagation.py:464-469)."""
    xx, yy = O.make_xy_grid(npup * ss, dx=dx / ss)
    fine = (np.hypot(xx, yy) < radius).astype(np.float32)
    return fine.reshape(npup, ss, npup, ss).mean(axis=(1, 3))


@pytest.fixture(scope='module')
def vortex_rig(pa):
    """charge-2 vortex coronagraph of the reference's tests (tests/test_propagation.py:473-519): 384^2 pupil, undersized Lyot stop,
    six-level multiresolution executor, final focus at lambda/D / 4."""
    P = pa.propagation
    wvl, efl, pupil_dx, npup, nd = O.HeNe, 100.0, 0.05, 384, 320
    Dap = nd * pupil_dx
    lamD = efl / Dap * wvl
    period = wvl * efl / pupil_dx
    pupil = _grey_circle(Dap / 2, npup, pupil_dx).astype(complex)
    lyot = _grey_circle(0.8 * Dap / 2, npup, pupil_dx)
    nf0 = 2 * nd
    executor = P.prepare_multiresolution(pupil_dx, npup, period / nf0, nf0, wvl, efl, num_levels=6, fine_samples=256, kind='mdft')
    nf, fdx = 256, lamD / 4
    final = P.prepare_executor(pupil_dx, npup, fdx, nf, wvl, efl, kind='mdft')
    ref_peak = (np.abs(tonp(P.focus_dft(pupil, final))) ** 2).max()
    fx = np.arange(-(nf // 2), nf // 2) * fdx
    XF, YF = np.meshgrid(fx, fx)
    return dict(pupil=pupil, lyot=lyot, executor=executor, final=final, ref_peak=ref_peak, rad=np.hypot(XF, YF) / lamD, lamD=lamD)


def _dark_hole_max(pa, rig, fpm):
    P = pa.propagation
    lyot_field = P.to_fpm_and_back_multiresolution(rig['pupil'], fpm, rig['executor'])
    from prysm_amd import _lib as L
    stopped = lyot_field * L.as_device(rig['lyot'].astype(np.float64))
    psf = np.abs(tonp(P.focus_dft(stopped, rig['final']))) ** 2 / rig['ref_peak']
    return psf[(rig['rad'] > 3) & (rig['rad'] < 10)].max()


def test_multiresolution_vortex_dark_hole_below_1e12(pa, vortex_rig):
    """tests/test_propagation.py:535-541: behind the 0.8 R Lyot stop the next focus is dark to below 1e-12 of the
    non-coronagraphic peak -- an end-to-end check of the matrix-DFT pair, the level windows and the vortex mask in fp64."""
    assert _dark_hole_max(pa, vortex_rig, pa.propagation.vortex_phase_mask(2)) < 1e-12


def test_measured_fpm_captures_manufacturing_error(pa, vortex_rig):
    """tests/test_propagation.py:671-700: a measured ideal map still suppresses strongly; a 50 mrad fabrication ripple
    makes the dark hole measurably brighter."""
    P = pa.propagation
    lamD = vortex_rig['lamD']

    def measured(error=None, charge=2, extent=40, per_lamD=8):
        mdx = lamD / per_lamD
        n = int(extent * per_lamD) // 2 * 2 + 1
        mx, my = O.make_xy_grid(n, dx=mdx)
        phase = charge * np.arctan2(my, mx)
        if error is not None:
            phase = phase + error(np.hypot(mx, my) / lamD)
        return np.exp(1j * phase), mdx

    ideal_map, mdx = measured()
    dh_ideal = _dark_hole_max(pa, vortex_rig, P.prepare_measured_fpm(ideal_map, mdx, charge=2))
    err_map, mdx = measured(error=lambda r: 0.05 * np.sin(2 * np.pi * r / 3.0))
    dh_error = _dark_hole_max(pa, vortex_rig, P.prepare_measured_fpm(err_map, mdx, charge=2))
    assert dh_ideal < 1e-5
    assert dh_error > 3 * dh_ideal
