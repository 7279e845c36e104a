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
