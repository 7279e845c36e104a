"""The reference's own unit tests for the path and the rows around it, restated one for one against prysm_amd on the HIP path.

Every function names the reference test it follows (tests/<file>:<line> under the upstream tree) and keeps its inputs, its check
and its tolerance; arrays are handed over as numpy and come back from the device.  Tests the other GPU files already restate
(test_gpu_reference_identities.py, test_gpu_parity.py) are not repeated here.  Helpers that live outside the path upstream
(coordinates.make_xy_grid, geometry.circle, degradations.smear_ft / jitter_ft) are written out locally as the few lines they are.
"""
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HeNe = 0.6328
SAMPLES = 32
ARRAY_SIZES = (8, 64, 512)


@pytest.fixture(scope='module')
def pa():
    import prysm_amd
    from prysm_amd import _lib
    _lib.load()
    assert torch.cuda.is_available()
    return prysm_amd


@pytest.fixture(scope='module')
def P(pa):
    return pa.propagation


def tonp(x):
    from prysm_amd.mathops import array_to_true_numpy
    if hasattr(x, 'data') and not isinstance(x, (np.ndarray, torch.Tensor)):
        x = x.data
    return array_to_true_numpy(x)


def cnormal(rng, shape):
    return rng.normal(size=shape) + 1j * rng.normal(size=shape)


def make_xy_grid(n, dx):
    c = (np.arange(n) - n // 2) * dx
    return np.meshgrid(c, c)


def fft_equivalent_coords(pa, samples):
    """tests/test_fttools.py:16-23: MDFT(...)(inp) / N equals the centred ortho FFT of an N x N input."""
    r = tonp(pa.fttools.fftrange(samples)).astype(float)
    return r, r.copy(), r / samples, r / samples


# ------------------------------------------------------------------------------------------------ tests/test_propagation.py

@pytest.mark.parametrize('dzeta', [1 / 128.0, 1 / 256.0, 11.123 / 128.0, 1e10 / 2048.0])
def test_psf_to_pupil_sample_inverts_pupil_to_psf_sample(P, dzeta):
    """tests/test_propagation.py:16-21 (exact equality)."""
    samples, wvl, efl = 128, 0.55, 10
    psf_sample = P.pupil_sample_to_psf_sample(dzeta, samples, wvl, efl)
    assert P.psf_sample_to_pupil_sample(psf_sample, samples, wvl, efl) == dzeta


def test_obj_oriented_wavefront_focusing_reverses(P):
    """tests/test_propagation.py:24-29."""
    z = np.random.rand(128, 128)
    wf = P.Wavefront(dx=1, cmplx_field=z, wavelength=HeNe)
    wf2 = wf.focus(1, 1).unfocus(1, 1)
    assert np.allclose(tonp(wf), tonp(wf2))


def test_wavefront_focus_adjoint_metadata_and_data(P):
    """tests/test_propagation.py:58-75."""
    rng = np.random.default_rng(135)
    dx, efl, Q = 0.25, 10, 2
    data = cnormal(rng, (8, 8))
    wf = P.Wavefront(dx=dx, cmplx_field=data, wavelength=HeNe, space='pupil')
    psf = wf.focus(efl=efl, Q=Q)
    grad_data = cnormal(rng, tuple(psf.data.shape))
    grad = P.Wavefront(dx=psf.dx, cmplx_field=grad_data, wavelength=HeNe, space='psf')
    back = grad.focus_adjoint(efl=efl, Q=Q)
    np.testing.assert_allclose(tonp(back), tonp(P.focus_adjoint(grad_data, Q=Q)))
    assert tuple(back.data.shape) == tuple(wf.data.shape)
    assert back.dx == pytest.approx(wf.dx)
    assert back.space == 'pupil'


def test_wavefront_unfocus_adjoint_metadata_and_data(P):
    """tests/test_propagation.py:78-95."""
    rng = np.random.default_rng(246)
    dx, efl, Q = 0.1, 10, 2
    data = cnormal(rng, (8, 8))
    wf = P.Wavefront(dx=dx, cmplx_field=data, wavelength=HeNe, space='psf')
    pupil = wf.unfocus(efl=efl, Q=Q)
    grad_data = cnormal(rng, tuple(pupil.data.shape))
    grad = P.Wavefront(dx=pupil.dx, cmplx_field=grad_data, wavelength=HeNe, space='pupil')
    back = grad.unfocus_adjoint(efl=efl, Q=Q)
    np.testing.assert_allclose(tonp(back), tonp(P.unfocus_adjoint(grad_data, Q=Q)))
    assert tuple(back.data.shape) == tuple(wf.data.shape)
    assert back.dx == pytest.approx(wf.dx)
    assert back.space == 'psf'


def test_unfocus_fft_mdft_equivalent_wavefront(P):
    """tests/test_propagation.py:98-106."""
    z = np.random.rand(128, 128)
    wf = P.Wavefront(dx=1, cmplx_field=z, wavelength=HeNe, space='psf')
    unfocus_fft = wf.unfocus(Q=2, efl=1)
    mdft = wf.prepare_executor(efl=1, dx=unfocus_fft.dx, samples=tuple(unfocus_fft.data.shape))
    assert np.allclose(tonp(unfocus_fft), tonp(wf.unfocus_dft(mdft)))


def test_prepare_executor_builds_fftdft_and_matches_mdft(pa, P):
    """tests/test_propagation.py:120-144."""
    rng = np.random.default_rng(2468)
    pupil_dx, pupil_shape, focal_shape, efl, fft_samples = 0.1, (32, 40), (48, 64), 10.0, 64
    focal_dx = HeNe * efl / (pupil_dx * fft_samples)
    focal_shift = (0.25 * focal_dx, -0.5 * focal_dx)
    fftdft = P.prepare_executor(pupil_dx, pupil_shape, focal_dx, focal_shape, HeNe, efl, focal_shift=focal_shift, kind='fftdft')
    mdft = P.prepare_executor(pupil_dx, pupil_shape, focal_dx, focal_shape, HeNe, efl, focal_shift=focal_shift, kind='mdft')
    pupil = cnormal(rng, pupil_shape)
    assert isinstance(fftdft, pa.fttools.FFTDFT)
    assert fftdft.pupil_dx == pupil_dx
    assert fftdft.focal_dx == focal_dx
    np.testing.assert_allclose(tonp(fftdft(pupil)), tonp(mdft(pupil)), rtol=1e-12, atol=1e-12)


def test_wavefront_prepare_executor_builds_fftdft(pa, P):
    """tests/test_propagation.py:147-166."""
    pupil_dx, samples, efl = 0.1, 32, 10.0
    focal_dx = HeNe * efl / (pupil_dx * samples)
    wf = P.Wavefront(dx=pupil_dx, cmplx_field=np.ones((samples, samples)), wavelength=HeNe, space='pupil')
    executor = wf.prepare_executor(efl, focal_dx, samples, kind='fftdft')
    assert isinstance(executor, pa.fttools.FFTDFT)
    np.testing.assert_allclose(tonp(wf.focus_dft(executor)), tonp(wf.focus_dft(wf.prepare_executor(efl, focal_dx, samples))),
                               rtol=1e-12, atol=1e-12)


def test_prepare_executor_fftdft_rejects_incompatible_sampling(P):
    """tests/test_propagation.py:169-175."""
    with pytest.raises(ValueError, match='not FFT-compatible'):
        P.prepare_executor(pupil_dx=0.1, pupil_samples=32, focal_dx=1.0, focal_samples=32, wavelength=HeNe, efl=10.0,
                           kind='fftdft')


def test_focus_dft_adjoint_is_adjoint(P):
    """tests/test_propagation.py:178-191."""
    rng = np.random.default_rng(159)
    x = cnormal(rng, (7, 9))
    mdft = P.prepare_executor(pupil_dx=0.25, pupil_samples=x.shape, focal_dx=0.1, focal_samples=(8, 11), wavelength=HeNe,
                              efl=10.0)
    y = cnormal(rng, (8, 11))
    lhs = np.vdot(tonp(P.focus_dft(x, mdft)), y)
    rhs = np.vdot(x, tonp(P.focus_dft_adjoint(y, mdft)))
    np.testing.assert_allclose(lhs, rhs, atol=1e-12)


def test_unfocus_dft_adjoint_is_adjoint(P):
    """tests/test_propagation.py:194-207."""
    rng = np.random.default_rng(7531)
    x = cnormal(rng, (8, 11))
    mdft = P.prepare_executor(pupil_dx=0.25, pupil_samples=(7, 9), focal_dx=0.1, focal_samples=x.shape, wavelength=HeNe,
                              efl=10.0)
    y = cnormal(rng, (7, 9))
    lhs = np.vdot(tonp(P.unfocus_dft(x, mdft)), y)
    rhs = np.vdot(x, tonp(P.unfocus_dft_adjoint(y, mdft)))
    np.testing.assert_allclose(lhs, rhs, atol=1e-12)


def test_free_space_zero_distance_is_identity(P):
    """tests/test_propagation.py:210-218."""
    z = np.random.rand(SAMPLES, SAMPLES)
    wf = P.Wavefront(dx=1, cmplx_field=z, wavelength=HeNe, space='pupil')
    out = wf.free_space(0)
    np.testing.assert_allclose(tonp(out), tonp(wf), atol=1e-12)
    assert out.dx == wf.dx
    assert out.wavelength == wf.wavelength


@pytest.mark.parametrize('Q', [1, 1.5, 2])
def test_angular_spectrum_adjoint_is_adjoint(P, Q):
    """tests/test_propagation.py:221-231."""
    rng = np.random.default_rng(321)
    x = cnormal(rng, (9, 12))
    fwd = tonp(P.angular_spectrum(x, wvl=HeNe, dx=0.25, z=1.2, Q=Q))
    y = rng.normal(size=fwd.shape)
    y = y + 1j * rng.normal(size=y.shape)
    lhs = np.vdot(fwd, y)
    rhs = np.vdot(x, tonp(P.angular_spectrum_adjoint(y, wvl=HeNe, dx=0.25, z=1.2, Q=Q)))
    np.testing.assert_allclose(lhs, rhs, atol=1e-12)


def test_angular_spectrum_adjoint_with_tf_is_adjoint(P):
    """tests/test_propagation.py:234-243."""
    rng = np.random.default_rng(654)
    x = cnormal(rng, (9, 12))
    y = cnormal(rng, x.shape)
    tf = P.angular_spectrum_transfer_function(x.shape, HeNe, 0.25, z=1.2)
    lhs = np.vdot(tonp(P.angular_spectrum(x, wvl=HeNe, dx=0.25, z=np.nan, tf=tf)), y)
    rhs = np.vdot(x, tonp(P.angular_spectrum_adjoint(y, wvl=HeNe, dx=0.25, z=np.nan, tf=tf)))
    np.testing.assert_allclose(lhs, rhs, atol=1e-12)


def test_wavefront_free_space_adjoint_metadata_and_data(P):
    """tests/test_propagation.py:246-266."""
    rng = np.random.default_rng(753)
    dx, dz, Q = 0.25, 1.2, 2
    data = cnormal(rng, (8, 8))
    wf = P.Wavefront(dx=dx, cmplx_field=data, wavelength=HeNe, space='pupil')
    out = wf.free_space(dz=dz, Q=Q)
    grad_data = cnormal(rng, tuple(out.data.shape))
    grad = P.Wavefront(dx=out.dx, cmplx_field=grad_data, wavelength=HeNe, space=out.space)
    back = grad.free_space_adjoint(dz=dz, Q=Q)
    np.testing.assert_allclose(tonp(back), tonp(P.angular_spectrum_adjoint(grad_data, wvl=HeNe, dx=dx, z=dz, Q=Q)))
    assert tuple(back.data.shape) == tuple(wf.data.shape)
    assert back.dx == pytest.approx(wf.dx)
    assert back.space == wf.space


def test_talbot_distance_and_fresnel_number(P):
    """tests/test_propagation.py:269-282."""
    wvl, a, z = 123.456, 987.654321, 5
    assert wvl / (1 - np.sqrt(1 - wvl ** 2 / a ** 2)) == pytest.approx(P.talbot_distance(a, wvl), abs=.1)
    assert P.fresnel_number(a, z, wvl) == (a ** 2 / (z * wvl))


def test_wavefront_multiply_and_divide_apply_to_data(P):
    """tests/test_propagation.py:285-290."""
    data = np.arange(4, dtype=float).reshape(2, 2).astype(np.complex128)
    wf = P.Wavefront(cmplx_field=data, dx=1, wavelength=.6328)
    np.testing.assert_allclose(tonp(wf * 2), data * 2)
    np.testing.assert_allclose(tonp(wf / 2), data / 2)


def test_wavefront_scalar_arithmetic_operand_order(P):
    """tests/test_propagation.py:293-298: sub and truediv compute self OP other."""
    data = (np.random.rand(2, 2) + 1).astype(np.complex128)
    wf = P.Wavefront(cmplx_field=data, dx=1, wavelength=.6328)
    assert np.allclose(tonp(wf - 1.0), data - 1.0)
    assert np.allclose(tonp(wf / 2.0), data / 2.0)


def test_wavefront_reverse_scalar_arithmetic(P):
    """tests/test_propagation.py:301-307."""
    data = (np.random.rand(2, 2) + 1).astype(np.complex128)
    wf = P.Wavefront(cmplx_field=data, dx=1, wavelength=.6328)
    np.testing.assert_allclose(tonp(2 * wf), 2 * data)
    np.testing.assert_allclose(tonp(2 + wf), 2 + data)
    np.testing.assert_allclose(tonp(2 - wf), 2 - data)
    np.testing.assert_allclose(tonp(2 / wf), 2 / data)


def test_wavefront_arithmetic_rejects_different_spaces(P):
    """tests/test_propagation.py:310-315."""
    data = np.ones((2, 2), dtype=complex)
    pupil = P.Wavefront(data, .6328, 1, 'pupil')
    psf = P.Wavefront(data, .6328, 1, 'psf')
    with pytest.raises(ValueError, match='space'):
        pupil + psf


def test_to_fpm_and_back_adjoint_accepts_wavefront_fpm(P):
    """tests/test_propagation.py:318-329."""
    z = np.random.rand(SAMPLES, SAMPLES) + 1j * np.random.rand(SAMPLES, SAMPLES)
    wf = P.Wavefront(cmplx_field=z, dx=1.0, wavelength=HeNe, space='pupil')
    fpm_data = (np.random.rand(SAMPLES, SAMPLES) + 1j * np.random.rand(SAMPLES, SAMPLES)).astype(np.complex128)
    fpm = P.Wavefront(cmplx_field=fpm_data, dx=0.1, wavelength=HeNe, space='psf')
    mdft = wf.prepare_executor(efl=10.0, dx=fpm.dx, samples=tuple(fpm.data.shape))
    out = wf.to_fpm_and_back(fpm=fpm, executor=mdft)
    grad = out.to_fpm_and_back_adjoint(fpm=fpm, executor=mdft)
    assert tuple(grad.data.shape) == tuple(wf.data.shape)


def test_to_fpm_and_back_adjoint_is_adjoint_for_input_field(P):
    """tests/test_propagation.py:336-350."""
    rng = np.random.default_rng(2468)
    x = cnormal(rng, (7, 9))
    fpm = cnormal(rng, (8, 11))
    y = cnormal(rng, x.shape)
    mdft = P.prepare_executor(pupil_dx=0.25, pupil_samples=x.shape, focal_dx=0.1, focal_samples=fpm.shape, wavelength=HeNe,
                              efl=10.0)
    lhs = np.vdot(tonp(P.to_fpm_and_back(x, fpm=fpm, executor=mdft)), y)
    rhs = np.vdot(x, tonp(P.to_fpm_and_back_adjoint(y, fpm=fpm, executor=mdft)))
    np.testing.assert_allclose(lhs, rhs, atol=1e-12)


def test_precomputed_angular_spectrum_matches_direct_zero_distance(P):
    """tests/test_propagation.py:430-437."""
    data = np.random.rand(4, 4)
    wf = P.Wavefront(cmplx_field=data, dx=1, wavelength=.6328)
    tf = P.angular_spectrum_transfer_function(tuple(wf.data.shape), wf.wavelength, wf.dx, z=0)
    out = wf.free_space(tf=tf)
    np.testing.assert_allclose(tonp(out), tonp(wf), atol=1e-12)


@pytest.mark.parametrize('kind', ['mdft', 'czt'])
def test_multiresolution_to_fpm_and_back_adjoint_is_adjoint(P, kind):
    """tests/test_propagation.py:543-558."""
    rng = np.random.default_rng(20240530)
    npup = 64
    executor = P.prepare_multiresolution(pupil_dx=0.1, pupil_samples=npup, focal_dx=2.0, focal_samples=32, wavelength=HeNe,
                                         efl=10.0, num_levels=3, fine_samples=32, kind=kind)
    fpm = P.vortex_phase_mask(2)
    x = rng.standard_normal((npup, npup)) + 1j * rng.standard_normal((npup, npup))
    y = rng.standard_normal((npup, npup)) + 1j * rng.standard_normal((npup, npup))
    lhs = np.vdot(tonp(P.to_fpm_and_back_multiresolution(x, fpm, executor)), y)
    rhs = np.vdot(x, tonp(P.to_fpm_and_back_multiresolution_adjoint(y, fpm, executor)))
    np.testing.assert_allclose(lhs, rhs, rtol=1e-10)


def test_vortex_phase_mask_rejects_non_integer_charge(P):
    """tests/test_propagation.py:561-564."""
    with pytest.raises(TypeError):
        P.vortex_phase_mask(2.5)
    P.vortex_phase_mask(np.int64(2))


def test_unit_cell_focal_grid_roundtrip_is_unitary(P):
    """tests/test_propagation.py:567-574."""
    pupil_dx, npup, efl = 0.1, 64, 50.0
    x, y = make_xy_grid(npup, dx=pupil_dx)
    pupil = (np.hypot(x, y) <= 2.4).astype(complex)
    fdx, nf = P.unit_cell_focal_grid(pupil_dx, 4.8, HeNe, efl)
    ex = P.prepare_executor(pupil_dx, npup, fdx, nf, HeNe, efl)
    rt = tonp(P.unfocus_dft(P.focus_dft(pupil, ex), ex))
    assert np.abs(rt - pupil).max() < 1e-12


def test_prepare_multiresolution_accepts_tuple_samples(P):
    """tests/test_propagation.py:577-586."""
    executor = P.prepare_multiresolution(pupil_dx=0.1, pupil_samples=32, focal_dx=2.0, focal_samples=(24, 40), wavelength=HeNe,
                                         efl=10.0, num_levels=2, fine_samples=16)
    assert tuple(executor.xf[0].shape) == (24, 40)
    fpm = P.vortex_phase_mask(2)
    x = np.random.rand(32, 32).astype(complex)
    out = P.to_fpm_and_back_multiresolution(x, fpm, executor)
    assert tuple(out.shape) == x.shape


def test_wavefront_multiresolution_wrappers(P):
    """tests/test_propagation.py:625-643."""
    rng = np.random.default_rng(11)
    npup, dx = 16, 0.25
    z = rng.standard_normal((npup, npup)) + 1j * rng.standard_normal((npup, npup))
    wf = P.Wavefront(cmplx_field=z, dx=dx, wavelength=HeNe, space='pupil')
    executor = wf.prepare_multiresolution(efl=10.0, focal_dx=4.0, focal_samples=16, num_levels=2, fine_samples=12)
    fpm = P.vortex_phase_mask(2)
    out, at_fpm, after_fpm = wf.to_fpm_and_back_multiresolution(fpm, executor, return_more=True)
    assert out.dx == wf.dx and out.space == 'pupil'
    assert at_fpm[1].dx == executor.executors[1].focal_dx and at_fpm[1].space == 'psf'
    np.testing.assert_allclose(tonp(out), tonp(P.to_fpm_and_back_multiresolution(z, fpm, executor)))
    grad, fpm_bars = out.to_fpm_and_back_multiresolution_adjoint(fpm, executor, return_fpm_grad=True, field_at_fpm=at_fpm)
    assert tuple(grad.data.shape) == z.shape and grad.space == 'pupil'
    assert len(fpm_bars) == len(executor)


def test_prepare_measured_fpm_interpolates_and_fills(P):
    """tests/test_propagation.py:646-659."""
    n, dx = 129, 0.4
    x, y = make_xy_grid(n, dx=dx)
    measurement = np.exp(1j * 2 * np.arctan2(y, x))
    fpm = P.prepare_measured_fpm(measurement, dx, charge=2)
    np.testing.assert_allclose(tonp(fpm(x, y)), measurement, atol=1e-12)
    far = np.full((1, 1), 1e5)
    ideal = np.exp(1j * 2 * np.arctan2(far, far))
    np.testing.assert_allclose(tonp(fpm(far, far)), ideal, atol=1e-12)


def test_prepare_measured_fpm_scalar_fill(P):
    """tests/test_propagation.py:662-669."""
    n = 65
    measurement = np.ones((n, n), dtype=complex)
    fpm = P.prepare_measured_fpm(measurement, 1.0, fill=0.0)
    far = np.full((1, 1), 1e3)
    assert tonp(fpm(far, far))[0, 0] == 0.0


# ---------------------------------------------------------------------------------------------------- tests/test_fttools.py

@pytest.mark.parametrize('samples', ARRAY_SIZES)
def test_mtp_equivalent_to_fft(pa, samples):
    """tests/test_fttools.py:26-32."""
    inp = np.random.rand(samples, samples)
    fft = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(inp), norm='ortho'))
    x, y, fx, fy = fft_equivalent_coords(pa, samples)
    assert np.allclose(fft, tonp(pa.fttools.MDFT(x, y, fx, fy)(inp)) / samples)


@pytest.mark.parametrize('samples', ARRAY_SIZES)
def test_mtp_reverses_self(pa, samples):
    """tests/test_fttools.py:35-44: adjoint(forward(inp)) = N^2 inp."""
    inp = np.random.rand(samples, samples)
    x, y, fx, fy = fft_equivalent_coords(pa, samples)
    op = pa.fttools.MDFT(x, y, fx, fy)
    back = tonp(op.adjoint(op(inp))) / (samples * samples)
    assert np.allclose(inp, back)


@pytest.mark.parametrize('input_shape,output_shape', [((3, 9), (2, 9)), ((9, 3), (9, 2))])
def test_mdft_rectangular_matches_explicit_chain(pa, input_shape, output_shape):
    """tests/test_fttools.py:55-87, the part that is behaviour: forward and adjoint of a rectangular MDFT against the explicit
    Ey . f . Ex^T chain built from the operator's own bases.  (The reference also pins which matrix its numpy path multiplies
    first -- a host-side flop heuristic; the device GEMM picks its order from the tile plan)"""
    rng = np.random.default_rng(123)
    (ny, nx), (my, mx) = input_shape, output_shape
    r = lambda n: tonp(pa.fttools.fftrange(n)).astype(float)   # noqa: E731
    op = pa.fttools.MDFT(r(nx), r(ny), r(mx) / nx, r(my) / ny, norm=0.25)
    inp = cnormal(rng, input_shape)
    grad = cnormal(rng, output_shape)
    Ex, Ey = tonp(op.Ex), tonp(op.Ey)
    np.testing.assert_allclose(tonp(op(inp)), (Ey @ inp @ Ex.T) * op.norm, atol=1e-12)
    np.testing.assert_allclose(tonp(op.adjoint(grad)), (Ey.conj().T @ grad @ Ex.conj()) * op.norm, atol=1e-12)


@pytest.mark.parametrize('shape', (8, 9, 12))
def test_pad2d_cropcenter_adjoints(pa, shape):
    """tests/test_fttools.py:90-95."""
    inp = np.random.rand(shape, shape)
    out = pa.fttools.crop_center(pa.fttools.pad2d(inp, Q=2), inp.shape)
    assert np.allclose(inp, tonp(out))


@pytest.mark.parametrize('samples', ARRAY_SIZES)
def test_czt_equiv_to_fft(pa, samples):
    """tests/test_fttools.py:98-104."""
    inp = np.random.rand(samples, samples)
    fft = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(inp), norm='ortho'))
    x, y, fx, fy = fft_equivalent_coords(pa, samples)
    assert np.allclose(fft, tonp(pa.fttools.CZT(x, y, fx, fy)(inp)) / samples)


@pytest.mark.parametrize('samples', ARRAY_SIZES)
def test_czt_reverses_self_complex(pa, samples):
    """tests/test_fttools.py:107-113."""
    inp = np.random.rand(samples, samples) + 1.0j * np.random.rand(samples, samples)
    x, y, fx, fy = fft_equivalent_coords(pa, samples)
    fwd = tonp(pa.fttools.CZT(x, y, fx, fy, sign=-1)(inp)) / samples
    back = tonp(pa.fttools.CZT(x, y, fx, fy, sign=+1)(fwd)) / samples
    assert np.allclose(inp, back)


@pytest.mark.parametrize('sign', (-1, 1))
@pytest.mark.parametrize('input_shape,output_shape', [((7, 9), (5, 6)), ((5, 6), (7, 9))])
def test_czt_adjoint_is_adjoint(pa, sign, input_shape, output_shape):
    """tests/test_fttools.py:116-139."""
    rng = np.random.default_rng(123)
    (ny, nx), (my, mx) = input_shape, output_shape
    r = lambda n: tonp(pa.fttools.fftrange(n)).astype(float)   # noqa: E731
    op = pa.fttools.CZT(r(nx) * 0.2, r(ny) * 0.17, (r(mx) + 0.25) * 0.13, (r(my) - 0.5) * 0.11, sign=sign, norm=0.3)
    inp = cnormal(rng, input_shape)
    grad = cnormal(rng, output_shape)
    lhs = np.vdot(tonp(op(inp)), grad)
    rhs = np.vdot(inp, tonp(op.adjoint(grad)))
    np.testing.assert_allclose(lhs, rhs, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('sign', (-1, 1))
def test_czt_matches_mdft_for_shifted_uniform_grids(pa, sign):
    """tests/test_fttools.py:142-157."""
    rng = np.random.default_rng(123)
    (ny, nx), (my, mx) = (7, 9), (5, 6)
    r = lambda n: tonp(pa.fttools.fftrange(n)).astype(float)   # noqa: E731
    x, y = r(nx) * 0.2 + 0.33, r(ny) * 0.17 - 0.41
    fx, fy = (r(mx) + 0.25) * 0.13, (r(my) - 0.5) * 0.11
    inp = cnormal(rng, (ny, nx))
    mdft = pa.fttools.MDFT(x, y, fx, fy, sign=sign)
    czt = pa.fttools.CZT(x, y, fx, fy, sign=sign)
    np.testing.assert_allclose(tonp(czt(inp)), tonp(mdft(inp)), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('sign', (-1, 1))
@pytest.mark.parametrize('input_shape,output_shape,fft_shape', [((7, 9), (5, 6), (12, 16)), ((5, 6), (7, 9), (12, 16))])
def test_fftdft_matches_mdft_for_shifted_rectangular_grids(pa, sign, input_shape, output_shape, fft_shape):
    """tests/test_fttools.py:160-184 (dy < 0 exercises the inverse-transform axis)."""
    rng = np.random.default_rng(123)
    (ny, nx), (my, mx), (ky, kx) = input_shape, output_shape, fft_shape
    r = lambda n: tonp(pa.fttools.fftrange(n)).astype(float)   # noqa: E731
    dx, dy = 0.2, -0.17
    x, y = r(nx) * dx + 0.33, r(ny) * dy - 0.41
    fx, fy = (r(mx) + 0.25) / (kx * dx), (r(my) - 0.5) / (ky * abs(dy))
    inp = cnormal(rng, input_shape)
    mdft = pa.fttools.MDFT(x, y, fx, fy, sign=sign, norm=0.3)
    fftdft = pa.fttools.FFTDFT(x, y, fx, fy, sign=sign, norm=0.3)
    np.testing.assert_allclose(tonp(fftdft(inp)), tonp(mdft(inp)), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('sign', (-1, 1))
@pytest.mark.parametrize('input_shape,output_shape,fft_shape', [((7, 9), (5, 6), (12, 16)), ((5, 6), (7, 9), (12, 16))])
def test_fftdft_adjoint_is_adjoint(pa, sign, input_shape, output_shape, fft_shape):
    """tests/test_fttools.py:187-211."""
    rng = np.random.default_rng(456)
    (ny, nx), (my, mx), (ky, kx) = input_shape, output_shape, fft_shape
    r = lambda n: tonp(pa.fttools.fftrange(n)).astype(float)   # noqa: E731
    dx, dy = 0.2, 0.17
    x, y = r(nx) * dx + 0.33, r(ny) * dy - 0.41
    fx, fy = (r(mx) + 0.25) / (kx * dx), (r(my) - 0.5) / (ky * dy)
    op = pa.fttools.FFTDFT(x, y, fx, fy, sign=sign, norm=0.3)
    inp = cnormal(rng, input_shape)
    grad = cnormal(rng, output_shape)
    lhs = np.vdot(tonp(op(inp)), grad)
    rhs = np.vdot(inp, tonp(op.adjoint(grad)))
    np.testing.assert_allclose(lhs, rhs, rtol=1e-12, atol=1e-12)


def test_fourier_resample_preserves_complex_data(pa):
    """tests/test_fttools.py:236-243."""
    data = np.ones((8, 8), dtype=complex) * (1 + 2j)
    out = tonp(pa.fttools.fourier_resample(data, 2))
    assert np.iscomplexobj(out)
    np.testing.assert_allclose(out, 1 + 2j, atol=1e-12)


# -------------------------------------------------------------------------------------------------------- tests/test_otf.py

def gaussian_psf(n=14, sig=0.6, x0=0.8, y0=-0.4):
    """tests/test_otf.py:43-53."""
    c = np.arange(n) - n // 2
    xx, yy = np.meshgrid(c, c)
    return np.exp(-((xx - x0) ** 2 + (yy - y0) ** 2) / (2 * sig ** 2))


def test_transform_psf_adjoint_dot_test(pa):
    """tests/test_otf.py:56-67."""
    rng = np.random.default_rng(0)
    n = 16
    x = rng.standard_normal((n, n))
    y = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    Ax, _ = pa.otf.transform_psf(x, dx=1.0)
    Aty = tonp(pa.otf.transform_psf_adjoint(y))
    lhs = np.sum(np.conj(tonp(Ax)) * y)
    rhs = np.sum(np.conj(x) * Aty)
    assert np.allclose(lhs, rhs, rtol=1e-10, atol=1e-10)


def test_mtf_ptf_otf_from_psf_matches_individual(pa):
    """tests/test_otf.py:141-156: the single-transform routine agrees BIT FOR BIT with the three per-quantity functions."""
    psf = gaussian_psf()
    dx = 1.0
    mtf, ptf, otf_, data = pa.otf.mtf_ptf_otf_from_psf(psf, dx, return_more=True)
    mtf_ref, data_ref = pa.otf.mtf_from_psf(psf, dx, return_more=True)
    ptf_ref = pa.otf.ptf_from_psf(psf, dx, return_more=True)[0]
    otf_ref = pa.otf.otf_from_psf(psf, dx, return_more=True)[0]
    assert np.array_equal(tonp(mtf), tonp(mtf_ref))
    assert np.array_equal(tonp(ptf), tonp(ptf_ref))
    assert np.array_equal(tonp(otf_), tonp(otf_ref))
    assert np.array_equal(tonp(data), tonp(data_ref))
    assert mtf.dx == mtf_ref.dx


def test_fused_and_composed_mtf_agree(pa):
    """the fused Hermitian-epilogue calls (no return_more) against the composed single-transform routine: same numbers to
    rounding (they are different kernels, so not bit for bit)."""
    psf = gaussian_psf()
    mtf, ptf, otf_ = pa.otf.mtf_ptf_otf_from_psf(psf, 1.0)
    np.testing.assert_allclose(tonp(pa.otf.mtf_from_psf(psf, 1.0)), tonp(mtf), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(tonp(pa.otf.otf_from_psf(psf, 1.0)), tonp(otf_), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(tonp(pa.otf.ptf_from_psf(psf, 1.0)), tonp(ptf), rtol=0, atol=1e-12)


# ------------------------------------------------------------------------------------------------ tests/test_convolution.py

def test_apply_transfer_functions_uses_callable_frequency_arguments(pa):
    """tests/test_convolution.py:21-32."""
    obj = np.arange(16, dtype=float).reshape(4, 4)

    def zero_lowpass(fx, fy, fr):
        assert tuple(fx.shape) == (1, obj.shape[1])
        assert tuple(fy.shape) == (obj.shape[0], 1)
        assert tuple(fr.shape) == obj.shape
        return fr * 0

    out = pa.convolution.apply_transfer_functions(obj, 1, [zero_lowpass])
    np.testing.assert_allclose(tonp(out), 0, atol=1e-12)


def test_apply_transfer_functions_identity_with_and_without_shift(pa):
    """tests/test_convolution.py:35-50."""
    obj = np.arange(16, dtype=float).reshape(4, 4)
    np.testing.assert_allclose(tonp(pa.convolution.apply_transfer_functions(obj, 1, [np.ones_like(obj)], shift=True)), obj,
                               atol=1e-12)
    np.testing.assert_allclose(tonp(pa.convolution.apply_transfer_functions(obj, 1, [np.ones_like(obj)])), obj, atol=1e-12)


def test_convolution_preserves_complex_input_dtype(pa):
    """tests/test_convolution.py:53-61."""
    obj = np.arange(25).reshape(5, 5) * (1 + 1j)
    psf = np.zeros_like(obj)
    psf[2, 2] = 1
    out = tonp(pa.convolution.conv(obj, psf))
    assert np.iscomplexobj(out)
    np.testing.assert_allclose(out, obj, atol=1e-12)


def test_apply_transfer_functions_rejects_callable_with_no_recognized_params(pa):
    """tests/test_convolution.py:64-71."""
    obj = np.arange(16, dtype=float).reshape(4, 4)

    def not_a_transfer_function(wavelength):
        return np.ones_like(obj)

    with pytest.raises(ValueError):
        pa.convolution.apply_transfer_functions(obj, 1, [not_a_transfer_function])


def _sinc(x):
    import torch as _t
    if isinstance(x, _t.Tensor):
        return _t.sinc(x)
    return np.sinc(x)


def smear_ft(fx, fy, width, height):
    """degradations.smear_ft as the two lines it is: the transform of a width x height box."""
    return _sinc(fx * width) * _sinc(fy * height)


def jitter_ft(fr, scale):
    """degradations.jitter_ft: the transform of a Gaussian blur of the given scale."""
    import torch as _t
    arg = -2 * (np.pi * scale * fr) ** 2
    return _t.exp(arg) if isinstance(arg, _t.Tensor) else np.exp(arg)


def test_apply_transfer_functions_composes_smear_and_jitter(pa):
    """tests/test_convolution.py:74-83: callables bound with functools.partial are called by their remaining parameter names."""
    sm = partial(smear_ft, width=1, height=1)
    ji = partial(jitter_ft, scale=1)
    obj = np.ones((8, 8), dtype=float)
    out = tonp(pa.convolution.apply_transfer_functions(obj, 1, [sm, ji]))
    assert out.shape == obj.shape
    assert np.isfinite(out).all()
    # a constant object only has a DC term, and both transfer functions are 1 there
    np.testing.assert_allclose(out, 1, atol=1e-12)


# ---------------------------------------------------------------------------------------------------- tests/test_physics.py

@pytest.mark.parametrize('efl, epd, wvl', [(10.0, 1.000, 0.5), (10.0, 1.000, 1.0), (3.00, 1.125, 3.0)])
def test_diffprop_matches_analyticmtf(pa, P, efl, epd, wvl):
    """tests/test_physics.py:41-57: the MTF of the FFT-propagated PSF of a circular pupil against the diffraction-limited MTF
    2/pi (acos s - s sqrt(1 - s^2)), s = f / (1 / (lambda fno)), along both axes (atol 1e-3, as upstream); the analytic formula
    (prysm/otf.py:496-545) and the centre slices (RichData.slices) are written out here"""
    fno = efl / epd
    x, y = make_xy_grid(128, dx=epd / 128)
    amp = (np.hypot(x, y) <= epd / 2).astype(float)
    wf = P.Wavefront.from_amp_and_phase(amp, None, wvl, float(x[0, 1] - x[0, 0]))
    psf = wf.focus(efl, Q=3).intensity
    mtf = pa.otf.mtf_from_psf(psf.data, psf.dx)
    data = tonp(mtf)
    n = data.shape[0]
    u = (np.arange(n) - n // 2) * float(mtf.dx)
    s = np.minimum(np.abs(u / (1 / (wvl / 1000 * fno))), 1.0)
    analytic = 2 / np.pi * (np.arccos(s) - s * np.sqrt(1 - s ** 2))
    assert np.allclose(analytic, data[n // 2, :], atol=1e-3)
    assert np.allclose(analytic, data[:, n // 2], atol=1e-3)
