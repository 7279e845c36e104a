"""Pin the CPU oracle (oracle/prysm_oracle.py) against the reference.

Three layers:
  1. committed golden fixtures generated FROM the reference (tests/golden/);
  2. the identities the reference's own tests pin (FFT==MDFT==CZT==FFTDFT,
     adjoint dot-products, round trips, Airy disk known answer);
  3. when /root/reference is importable (build container only): live comparison.
"""
import os
import sys

import numpy as np
import pytest

from oracle import prysm_oracle as O
from conftest import rel_max

TOL = 1e-12


def _cases(g):
    return range(int(g['ncases']))


def test_fft_family_golden(golden):
    g = golden('fft_family')
    for i in _cases(g):
        x, Q, gg = g[f'c{i}_x'], float(g[f'c{i}_Q']), g[f'c{i}_g']
        assert rel_max(O.focus(x, Q), g[f'c{i}_focus']) < TOL
        assert rel_max(O.unfocus(x, Q), g[f'c{i}_unfocus']) < TOL
        fa = O.focus_adjoint(gg, Q)
        assert fa.shape == g[f'c{i}_focus_adjoint'].shape == x.shape
        assert rel_max(fa, g[f'c{i}_focus_adjoint']) < TOL
        assert rel_max(O.unfocus_adjoint(gg, Q), g[f'c{i}_unfocus_adjoint']) < TOL


def test_padcrop_golden(golden):
    g = golden('padcrop')
    cases = [((8, 8), 2, None), ((9, 9), 2, None), ((12, 12), 1.5, None), ((9, 12), 1.5, None),
             ((9, 12), None, (14, 18)), ((5, 8), None, (16, 16)), ((8, 5), 3, None)]
    for i, (shape, Q, oshape) in enumerate(cases):
        x = g[f'c{i}_x']
        p = O.pad2d(x, Q) if oshape is None else O.pad2d(x, out_shape=oshape)
        assert np.array_equal(p, g[f'c{i}_pad'])
        assert np.array_equal(O.crop_center(p, shape), g[f'c{i}_crop'])
        assert np.array_equal(O.crop_center(p, shape), x)
    assert np.array_equal(O.pad2d(g['fill_x'], Q=2, value=1.5), g['fill_pad'])


def test_angular_spectrum_golden(golden):
    g = golden('angular_spectrum')
    for i in _cases(g):
        x = g[f'c{i}_x']
        Q, wvl, dx, z = (float(v) for v in g[f'c{i}_par'])
        y = O.angular_spectrum(x, wvl, dx, z, Q=Q)
        assert rel_max(y, g[f'c{i}_y']) < TOL
        assert rel_max(O.angular_spectrum_transfer_function(y.shape, wvl, dx, z), g[f'c{i}_tf']) < TOL
        assert rel_max(O.angular_spectrum_adjoint(g[f'c{i}_g'], wvl, dx, z, Q=Q), g[f'c{i}_adj']) < TOL
        tf = g[f'c{i}_usertf']
        assert rel_max(O.angular_spectrum(x, wvl, dx, z, Q=Q, tf=tf), g[f'c{i}_y_usertf']) < TOL
        assert rel_max(O.angular_spectrum_adjoint(x, wvl, dx, z, Q=Q, tf=tf), g[f'c{i}_adj_usertf']) < TOL


def test_executors_golden(golden):
    g = golden('executors')
    for i in _cases(g):
        x, gg = g[f'c{i}_x'], g[f'c{i}_g']
        ps, fs = tuple(g[f'c{i}_ps']), tuple(g[f'c{i}_fs'])
        pdx, fdx, wvl, efl, sx, sy = (float(v) for v in g[f'c{i}_par'])
        cx, cy, cfx, cfy = O.coordinates_for_focus(pdx, ps, fdx, fs, wvl, efl, (sx, sy))
        for a, b in ((cx, 'cx'), (cy, 'cy'), (cfx, 'cfx'), (cfy, 'cfy')):
            assert np.array_equal(a, g[f'c{i}_{b}'])
        for kind in ('mdft', 'czt'):
            ex = O.prepare_executor(pdx, ps, fdx, fs, wvl, efl, focal_shift=(sx, sy), kind=kind)
            assert rel_max(O.focus_dft(x, ex), g[f'c{i}_{kind}_fwd']) < 1e-11
            assert rel_max(O.unfocus_dft(gg, ex), g[f'c{i}_{kind}_adj']) < 1e-11
    # FFT-compatible grid: all three executors and the FFT agree
    x = g['fftdft_x']
    pdx, fdx, wvl, efl = (float(v) for v in g['fftdft_par'])
    samples = tuple(g['fftdft_samples'])
    f = O.focus(x, 2)
    assert rel_max(f, g['fftdft_fftfocus']) < TOL
    for kind in ('mdft', 'czt', 'fftdft'):
        ex = O.prepare_executor(pdx, x.shape, fdx, samples, wvl, efl, kind=kind)
        assert rel_max(ex(x), g[f'fftdft_{kind}_fwd']) < 1e-11
        assert rel_max(ex(x), f) < 1e-10
        assert rel_max(ex.adjoint(f), g[f'fftdft_{kind}_adj']) < 1e-11
    ex = O.prepare_executor(pdx, x.shape, fdx, (24, 40), wvl, efl, kind='fftdft')
    assert rel_max(ex(x), g['fftdft_crop_fwd']) < 1e-11
    assert rel_max(ex.adjoint(g['fftdft_crop_g']), g['fftdft_crop_adj']) < 1e-11


def test_wavefront_golden(golden):
    g = golden('wavefront')
    A = g['cfg1_amp']
    dx = float(g['cfg1_dx'])
    n = A.shape[0]
    x, y = O.make_xy_grid(n, diameter=10)
    r, t = O.cart_to_polar(x, y)
    assert np.array_equal(O.circle(5, r), A)
    assert float(x[0, 1] - x[0, 0]) == dx
    psf = O.focus(O.from_amp_and_phase(A, None, O.HeNe), 2)
    psf_dx = O.pupil_sample_to_psf_sample(dx, psf.shape[1], O.HeNe, 100)
    assert psf_dx == float(g['cfg1_psf_dx'])
    I = O.intensity(psf)
    assert rel_max(I, g['cfg1_intensity']) < TOL
    # Airy-disk known answer, tolerance as the reference (tests/test_physics.py:20-34)
    xx, yy = O.make_xy_grid(psf.shape, dx=psf_dx)
    rr, _ = O.cart_to_polar(xx, yy)
    airy = O.airydisk(rr, 10, O.HeNe)
    assert rel_max(airy, g['cfg1_airy']) < TOL
    c = psf.shape[0] // 2
    sl = I[c, c:c + 8] / I[c, c]
    assert np.allclose(sl, airy[c, c:c + 8], atol=5e-3)

    opd = O.hopkins_w040(r / 5, 500.0)
    assert rel_max(opd, g['opd']) < TOL
    P = O.from_amp_and_phase(A, g['opd'], 0.55)
    assert rel_max(P, g['fap_field']) < TOL
    assert rel_max(O.from_amp_and_phase(1.0, g['opd'], 0.55), g['phase_screen']) < TOL
    assert rel_max(O.thin_lens(250.0, 0.55, g['xgrid'], g['ygrid']), g['thin_lens']) < TOL
    f2 = O.focus(P, 2)
    assert rel_max(O.intensity(f2), g['fap_psf_intensity']) < TOL
    assert rel_max(2 * g['ibar'] * f2, g['intensity_adjoint']) < TOL
    k = O.phase_prefix(0.55)
    assert rel_max(k * np.imag(g['wfbar'] * np.conj(P)), g['fap_adjoint_phase']) < TOL
    assert rel_max(O.angular_spectrum(P, 0.55, dx, 5.0, Q=1), g['free_space']) < TOL
    # polychromatic recipe
    comps = []
    for w in g['poly_wvls']:
        Pw = O.from_amp_and_phase(A, g['opd'], float(w))
        ex = O.prepare_executor(dx, Pw.shape, float(g['poly_fdx']), (32, 32), float(w), 100)
        comps.append(O.intensity(ex(Pw)))
    assert rel_max(O.sum_of_2d_modes(np.asarray(comps), g['poly_weights']), g['poly_sum']) < 1e-11
    # otf / conv
    assert rel_max(O.transform_psf(g['cfg1_intensity']), g['otf_transform']) < TOL
    assert rel_max(O.mtf_from_psf(g['cfg1_intensity']), g['otf_mtf']) < TOL
    assert rel_max(O.conv(g['conv_obj'], g['conv_psf']), g['conv_out']) < 1e-11


def test_coronagraph_golden(golden):
    g = golden('coronagraph')
    pdx, fdx, wvl, efl = (float(v) for v in g['par'])
    x, fpm = g['x'], g['fpm']
    ex = O.prepare_executor(pdx, x.shape, fdx, fpm.shape, wvl, efl)
    nxt, at_fpm, after = O.to_fpm_and_back(x, fpm, ex, return_more=True)
    assert rel_max(nxt, g['tfab']) < 1e-11 and rel_max(at_fpm, g['tfab_at']) < 1e-11 and rel_max(after, g['tfab_after']) < 1e-11
    assert rel_max(O.to_fpm_and_back_adjoint(g['g'], fpm, ex), g['tfab_adj']) < 1e-11
    assert rel_max(O._adjoint_multiply(ex(g['g']), at_fpm), g['tfab_fpmbar']) < 1e-11
    assert rel_max(O.babinet(x, g['lyot'], g['fpm_real'], ex), g['babinet']) < 1e-11
    assert rel_max(O.babinet_adjoint(g['g'], g['lyot'], g['fpm_real'], ex), g['babinet_adj']) < 1e-11


def test_precision32_golden(golden):
    g = golden('precision32')
    x = g['x']
    f = O.focus(x, 2)
    assert f.dtype == np.complex64 == g['focus'].dtype
    assert rel_max(f, g['focus']) < 1e-6
    y = O.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1, precision=np.float32)
    assert y.dtype == g['as'].dtype == np.complex64
    assert rel_max(y, g['as']) < 1e-5
    ex = O.prepare_executor(0.1, (32, 32), 1.0, (16, 16), O.HeNe, 50.0, precision=np.float32)
    m = ex(x)
    assert m.dtype == g['mdft'].dtype == np.complex64
    assert rel_max(m, g['mdft']) < 1e-5


# ---- identities the reference's own tests pin --------------------------------

@pytest.mark.parametrize('Q', [1, 1.5, 2])
def test_adjoint_dot_products(Q):
    """reference tests/test_propagation.py:32-55."""
    rng = np.random.default_rng(789)
    x = rng.normal(size=(9, 12)) + 1j * rng.normal(size=(9, 12))
    for fwd, adj in ((O.focus, O.focus_adjoint), (O.unfocus, O.unfocus_adjoint)):
        y = rng.normal(size=fwd(x, Q).shape) + 1j * rng.normal(size=fwd(x, Q).shape)
        np.testing.assert_allclose(np.vdot(fwd(x, Q), y), np.vdot(x, adj(y, Q)), atol=1e-12)


def test_focus_unfocus_roundtrip_and_unitarity():
    """reference tests/test_propagation.py:24-29."""
    rng = np.random.default_rng(1)
    x = rng.random((128, 128))
    assert np.allclose(O.unfocus(O.focus(x, 1), 1), x)
    f = O.focus(x, 2)
    assert np.isclose(np.sum(O.intensity(f)), np.sum(x * x))


@pytest.mark.parametrize('n', [8, 64, 512])
def test_mdft_equals_shifted_fft(n):
    """reference tests/test_fttools.py:26-32."""
    rng = np.random.default_rng(n)
    a = rng.random((n, n))
    x = O.fftrange(n, dtype=np.float64)
    fx = O.fftrange(n, dtype=np.float64) / n
    m = O.MDFT(x, x, fx, fx, norm=1 / n)(a)
    assert np.allclose(m, O.focus(a, 1))


def test_free_space_zero_is_identity():
    """reference tests/test_propagation.py:210-218."""
    rng = np.random.default_rng(3)
    x = rng.normal(size=(16, 16)) + 1j * rng.normal(size=(16, 16))
    assert np.allclose(O.angular_spectrum(x, 0.5, 0.01, 0.0, Q=1), x)


# ---- live comparison against the reference (build container only) ------------

REF = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'prysm')), reason='reference not present on this box')
def test_oracle_matches_live_reference():
    sys.path.insert(0, REF)
    try:
        from prysm import propagation as P, fttools as F
    finally:
        sys.path.remove(REF)
    rng = np.random.default_rng(99)
    for shape, Q in (((48, 40), 1), ((33, 20), 1.5), ((64, 64), 2)):
        x = rng.normal(size=shape) + 1j * rng.normal(size=shape)
        assert rel_max(O.focus(x, Q), P.focus(x, Q)) < TOL
        assert rel_max(O.unfocus(x, Q), P.unfocus(x, Q)) < TOL
        assert rel_max(O.angular_spectrum(x, 0.6, 0.02, 7.0, Q=Q), P.angular_spectrum(x, 0.6, 0.02, 7.0, Q=Q)) < TOL
        for kind in ('mdft', 'czt'):
            a = O.prepare_executor(0.1, shape, 1.3, (20, 28), 0.6, 40.0, (0.3, 0.1), kind)
            b = P.prepare_executor(0.1, shape, 1.3, (20, 28), 0.6, 40.0, (0.3, 0.1), kind)
            assert rel_max(a(x), b(x)) < 1e-11
            g = rng.normal(size=(20, 28)) + 1j * rng.normal(size=(20, 28))
            assert rel_max(a.adjoint(g), b.adjoint(g)) < 1e-11
    assert np.array_equal(O.pad2d(x, 1.7), F.pad2d(x, 1.7))


def test_next_rows_oracle_vs_golden(golden):
    """apply_transfer_functions, fourier_resample, jones_adapter restatements against the reference's outputs."""
    g = golden('next_rows')
    obj, tf1, tf2 = g['atf_obj'], g['atf_tf1'], g['atf_tf2']
    assert rel_max(O.apply_transfer_functions(obj, None, [tf1, tf2], shift=False), g['atf_arrays_noshift']) < 1e-13
    assert rel_max(O.apply_transfer_functions(obj, None, [tf1, tf2], shift=True), g['atf_arrays_shift']) < 1e-13

    def gauss(fr):
        return np.exp(-(fr / 3.0) ** 2)

    def ramp(fx, fy):
        return np.exp(-2j * np.pi * (0.01 * fx + 0.02 * fy))

    cobj = g['atf_cobj']
    assert rel_max(O.apply_transfer_functions(cobj, 0.05, [gauss, ramp], shift=False), g['atf_callable_noshift']) < 1e-13
    assert rel_max(O.apply_transfer_functions(cobj, 0.05, [gauss, ramp], shift=True), g['atf_callable_shift']) < 1e-13
    f, gg = g['fr_f'], g['fr_g']
    assert rel_max(O.fourier_resample(f, 2), g['fr_up2']) < 1e-12
    assert rel_max(O.fourier_resample(f, 1.5), g['fr_up15']) < 1e-12
    assert rel_max(O.fourier_resample(f, (0.75, 1.25)), g['fr_aniso']) < 1e-12
    assert rel_max(O.fourier_resample(gg, 1.7), g['fr_g_up']) < 1e-12
    assert O.fourier_resample(f, 2).dtype == np.float64 and np.iscomplexobj(O.fourier_resample(gg, 1.7))
    J = g['jones_in']
    assert rel_max(O.jones_adapter(O.focus)(J, 2), g['jones_focus_Q2']) < 1e-13
    assert rel_max(O.jones_adapter(O.unfocus)(J, 1), g['jones_unfocus_Q1']) < 1e-13
    assert rel_max(O.jones_adapter(O.angular_spectrum)(J, 0.6328, 0.01, 25.0, Q=2), g['jones_as']) < 1e-13


def test_multiresolution_oracle_vs_golden(golden):
    """prepare_multiresolution (windows, grids), the multi-level FPM round trip and its adjoint, thin_lens_adjoint."""
    g = golden('multires')
    x, gg = g['x'], g['g']
    n = x.shape[0]
    for kind in ('mdft', 'czt'):
        ex = O.prepare_multiresolution(0.25, (n, n), 3.0, (24, 20), float(g['par'][2]), 80.0, 3, scaling=3.0, fine_samples=16,
                                       window=(0.25, 0.65), kind=kind)
        fpm = O.vortex_phase_mask(2)
        assert rel_max(O.to_fpm_and_back_multiresolution(x, fpm, ex), g[f'{kind}_fwd']) < 1e-12
        assert rel_max(O.to_fpm_and_back_multiresolution_adjoint(gg, fpm, ex), g[f'{kind}_adj']) < 1e-12
    for k in range(3):
        assert rel_max(ex.windows[k], g[f'window{k}']) < 1e-14
        assert rel_max(ex.xf[k], g[f'xf{k}']) < 1e-14
    got = O.thin_lens_adjoint(250.0, float(g['par'][2]), g['tl_x'], g['tl_y'], g['tl_Lbar'])
    assert abs(got - float(g['tl_grad'])) < 1e-12 * abs(float(g['tl_grad']))


def test_measured_fpm_oracle_vs_golden(golden):
    """prepare_measured_fpm: the restated map_coordinates (orders 0 .. 5, mode nearest) against scipy's through the reference."""
    g = golden('multires')
    meas, xf, yf = g['meas'], g['meas_xf'], g['meas_yf']
    for order in (0, 1, 2, 3, 4, 5):   # 2 .. 5: scipy's edge padding + recursive prefilter + B-spline taps, restated
        got = O.prepare_measured_fpm(meas, 0.6, center=(0.3, -0.2), charge=2, order=order)(xf, yf)
        assert rel_max(got, g[f'meas_o{order}_vortex']) < (1e-13 if order < 4 else 1e-12)
    assert rel_max(O.prepare_measured_fpm(meas, 0.6)(xf, yf), g['meas_o1_one']) < 1e-13
    assert rel_max(O.prepare_measured_fpm(meas, 0.6, fill=0.25 - 0.5j)(xf, yf), g['meas_o1_fill']) < 1e-13
    n = g['x'].shape[0]
    ex = O.prepare_multiresolution(0.25, (n, n), 3.0, (24, 20), float(g['par'][2]), 80.0, 3, scaling=3.0, fine_samples=16,
                                   window=(0.25, 0.65), kind='mdft')
    fpm = O.prepare_measured_fpm(meas, 0.6, center=(0.3, -0.2), charge=2)
    assert rel_max(O.to_fpm_and_back_multiresolution(g['x'], fpm, ex), g['meas_fwd']) < 1e-12


def test_otf_adjoints_oracle_vs_golden(golden):
    g = golden('multires')
    psf = g['otf_psf']
    assert rel_max(O.mtf_from_psf_adjoint(g['otf_mtf_bar'], psf), g['otf_mtf_adj']) < 1e-12
    assert rel_max(O.ptf_from_psf_adjoint(g['otf_ptf_bar'], psf), g['otf_ptf_adj']) < 1e-12
    assert rel_max(O.otf_from_psf_adjoint(g['otf_otf_bar'], psf), g['otf_otf_adj']) < 1e-12


def test_encircled_energy_oracle_vs_golden(golden):
    """otf.encircled_energy (+ adjoint), prysm/otf.py:319-472, against the reference's outputs."""
    g = golden('encircled')
    dx, radii = float(g['psf_dx']), g['radii']
    assert abs(O.encircled_energy(g['psf'], dx, 7.72) - float(g['ee_scalar'])) < TOL
    assert rel_max(O.encircled_energy(g['psf'], dx, radii), g['ee_many']) < TOL
    assert rel_max(O.encircled_energy(g['psf_noisy'], dx, radii), g['ee_noisy']) < TOL
    assert rel_max(O.encircled_energy_adjoint(g['ee_bar'], g['psf_noisy'], dx, radii), g['ee_adj_many']) < TOL
    assert rel_max(O.encircled_energy_adjoint(0.7, g['psf_noisy'], dx, 12.0), g['ee_adj_scalar']) < TOL
    assert rel_max(O.encircled_energy(g['rect'], 0.8, radii[:3]), g['rect_ee']) < TOL
    assert rel_max(O.encircled_energy_adjoint(g['ee_bar'][:3], g['rect'], 0.8, radii[:3]), g['rect_adj']) < TOL
    # known answer: a diffraction-limited circular aperture encircles ~83.8 % inside the first Airy zero (1.22 lambda N)
    x = np.arange(-128, 128) * (10 / 256)
    r = np.hypot(*np.meshgrid(x, x))
    psf = np.abs(O.focus((r <= 5).astype(np.complex128), 4)) ** 2
    psf_dx = 100 * O.HeNe / (10 / 256 * 1024)       # efl * wvl / (dx * N), microns
    ee = O.encircled_energy(psf, psf_dx, 1.22 * O.HeNe * 10)   # F/10
    assert abs(ee - 0.838) < 5e-3
