"""Generate golden input/output fixtures from the REFERENCE itself.

Run in the build container (the only place /root/reference exists):

    python tests/golden/make_golden.py

Imports brandondube/prysm v0.22 from /root/reference (read-only), runs the hot
path on small seeded inputs on its default numpy/scipy backend and stores
inputs + outputs as ``tests/golden/*.npz``.  The fixtures travel to the GPU
box; the reference does not.  Sizes are deliberately small and awkward
(9x12, 7x9, Q=1.5, odd/even, rectangular) -- they are the shapes the
reference's own tests use (tests/test_propagation.py, tests/test_fttools.py).
"""
import os
import sys

import numpy as np

REF = os.environ.get('PRYSM_REFERENCE', '/root/reference')
sys.path.insert(0, REF)

from prysm import propagation, fttools, coordinates, geometry, polynomials  # noqa: E402
from prysm import otf as potf, convolution as pconv, psf as ppsf  # noqa: E402
from prysm.conf import config  # noqa: E402
from prysm.wavelengths import HeNe  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def crandn(rng, shape, dtype=np.complex128):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)


def fft_family():
    out = {}
    rng = np.random.default_rng(20260925)
    cases = [((9, 12), 1), ((9, 12), 1.5), ((9, 12), 2), ((16, 16), 1), ((16, 16), 2),
             ((32, 64), 1), ((64, 32), 2), ((7, 9), 1), ((7, 9), 2), ((64, 64), 1),
             ((24, 40), 1.5), ((8, 8), 4)]
    for i, (shape, Q) in enumerate(cases):
        x = crandn(rng, shape)
        out[f'c{i}_x'] = x
        out[f'c{i}_Q'] = np.float64(Q)
        f = propagation.focus(x, Q)
        u = propagation.unfocus(x, Q)
        out[f'c{i}_focus'] = f
        out[f'c{i}_unfocus'] = u
        g = crandn(rng, f.shape)
        out[f'c{i}_g'] = g
        out[f'c{i}_focus_adjoint'] = propagation.focus_adjoint(g, Q)
        out[f'c{i}_unfocus_adjoint'] = propagation.unfocus_adjoint(g, Q)
    out['ncases'] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, 'fft_family.npz'), **out)


def padcrop():
    out = {}
    rng = np.random.default_rng(7)
    cases = [((8, 8), 2, None), ((9, 9), 2, None), ((12, 12), 1.5, None), ((9, 12), 1.5, None),
             ((9, 12), None, (14, 18)), ((5, 8), None, (16, 16)), ((8, 5), 3, None)]
    for i, (shape, Q, oshape) in enumerate(cases):
        x = crandn(rng, shape)
        out[f'c{i}_x'] = x
        if oshape is None:
            p = fttools.pad2d(x, Q)
        else:
            p = fttools.pad2d(x, out_shape=oshape)
        out[f'c{i}_pad'] = p
        out[f'c{i}_crop'] = fttools.crop_center(p, shape)
    out['ncases'] = np.int64(len(cases))
    # non-zero fill value and real dtype
    xr = rng.standard_normal((6, 7))
    out['fill_x'] = xr
    out['fill_pad'] = fttools.pad2d(xr, Q=2, value=1.5)
    np.savez_compressed(os.path.join(HERE, 'padcrop.npz'), **out)


def angular():
    out = {}
    rng = np.random.default_rng(11)
    cases = [((16, 16), 1, 0.5, 0.01, 10.0), ((9, 12), 1, 0.55, 0.02, 3.0), ((9, 12), 1.5, 0.55, 0.02, 3.0),
             ((32, 32), 2, HeNe, 0.01, 25.0), ((16, 32), 1, HeNe, 0.005, 1.0), ((7, 9), 2, 1.0, 0.1, 100.0)]
    for i, (shape, Q, wvl, dx, z) in enumerate(cases):
        x = crandn(rng, shape)
        out[f'c{i}_x'] = x
        out[f'c{i}_par'] = np.array([Q, wvl, dx, z], dtype=np.float64)
        y = propagation.angular_spectrum(x, wvl, dx, z, Q=Q)
        out[f'c{i}_y'] = y
        out[f'c{i}_tf'] = propagation.angular_spectrum_transfer_function(y.shape, wvl, dx, z)
        g = crandn(rng, y.shape)
        out[f'c{i}_g'] = g
        out[f'c{i}_adj'] = propagation.angular_spectrum_adjoint(g, wvl, dx, z, Q=Q)
        tf = crandn(rng, shape)
        out[f'c{i}_usertf'] = tf
        out[f'c{i}_y_usertf'] = propagation.angular_spectrum(x, wvl, dx, z, Q=Q, tf=tf)
        out[f'c{i}_adj_usertf'] = propagation.angular_spectrum_adjoint(x, wvl, dx, z, Q=Q, tf=tf)
    out['ncases'] = np.int64(len(cases))
    np.savez_compressed(os.path.join(HERE, 'angular_spectrum.npz'), **out)


def executors():
    out = {}
    rng = np.random.default_rng(13)
    # (pupil_samples, focal_samples, pupil_dx, focal_dx, wvl, efl, shift)
    cases = [((32, 32), (16, 16), 0.1, 1.0, HeNe, 50.0, (0.0, 0.0)),
             ((9, 12), (8, 11), 0.25, 2.0, 0.55, 10.0, (0.0, 0.0)),
             ((7, 9), (8, 11), 0.25, 2.0, 0.55, 10.0, (0.7, -1.3)),
             ((64, 48), (20, 36), 0.05, 0.8, 0.8, 120.0, (3.0, 1.0)),
             ((48, 64), (96, 80), 0.05, 0.8, 0.8, 120.0, (0.4, 0.4))]
    for i, (ps, fs, pdx, fdx, wvl, efl, shift) in enumerate(cases):
        x = crandn(rng, ps)
        g = crandn(rng, fs)
        out[f'c{i}_x'] = x
        out[f'c{i}_g'] = g
        out[f'c{i}_ps'] = np.array(ps)
        out[f'c{i}_fs'] = np.array(fs)
        out[f'c{i}_par'] = np.array([pdx, fdx, wvl, efl, shift[0], shift[1]], dtype=np.float64)
        for kind in ('mdft', 'czt'):
            ex = propagation.prepare_executor(pdx, ps, fdx, fs, wvl, efl, focal_shift=shift, kind=kind)
            out[f'c{i}_{kind}_fwd'] = propagation.focus_dft(x, ex)
            out[f'c{i}_{kind}_adj'] = propagation.unfocus_dft(g, ex)
        cx, cy, cfx, cfy = propagation.coordinates_for_focus(pdx, ps, fdx, fs, wvl, efl, shift)
        out[f'c{i}_cx'], out[f'c{i}_cy'], out[f'c{i}_cfx'], out[f'c{i}_cfy'] = cx, cy, cfx, cfy
    out['ncases'] = np.int64(len(cases))

    # FFT-compatible grid: FFTDFT == MDFT == FFT  (tests/test_propagation.py:120-166)
    n, Q = 32, 2
    x = crandn(rng, (n, n))
    wf = propagation.Wavefront(x, HeNe, 0.1, 'pupil')
    f = wf.focus(efl=25.0, Q=Q)
    out['fftdft_x'] = x
    out['fftdft_par'] = np.array([0.1, f.dx, HeNe, 25.0], dtype=np.float64)
    out['fftdft_samples'] = np.array(f.data.shape)
    out['fftdft_fftfocus'] = f.data
    for kind in ('mdft', 'czt', 'fftdft'):
        ex = wf.prepare_executor(efl=25.0, dx=f.dx, samples=f.data.shape, kind=kind)
        out[f'fftdft_{kind}_fwd'] = wf.focus_dft(ex).data
        out[f'fftdft_{kind}_adj'] = f.unfocus_dft(ex).data
    # cropped fftdft (focal window smaller than the FFT length)
    ex = wf.prepare_executor(efl=25.0, dx=f.dx, samples=(24, 40), kind='fftdft')
    out['fftdft_crop_fwd'] = wf.focus_dft(ex).data
    gg = crandn(rng, (24, 40))
    out['fftdft_crop_g'] = gg
    out['fftdft_crop_adj'] = ex.adjoint(gg)
    np.savez_compressed(os.path.join(HERE, 'executors.npz'), **out)


def wavefront_and_physics():
    out = {}
    rng = np.random.default_rng(17)
    # config 1 (plumbing): 512^2 circular pupil would be 8 MB of fixture; use the
    # same construction at 64^2 -- the code path is identical.
    n = 64
    x, y = coordinates.make_xy_grid(n, diameter=10)
    r, t = coordinates.cart_to_polar(x, y)
    A = geometry.circle(5, r)
    dx = float(x[0, 1] - x[0, 0])
    wf = propagation.Wavefront.from_amp_and_phase(A, None, HeNe, dx)
    psf = wf.focus(100, Q=2)
    out['cfg1_amp'] = A
    out['cfg1_dx'] = np.float64(dx)
    out['cfg1_psf_dx'] = np.float64(psf.dx)
    out['cfg1_intensity'] = psf.intensity.data
    # airy-disk known answer (tests/test_physics.py:20-34)
    xx, yy = coordinates.make_xy_grid(psf.data.shape, dx=psf.dx)
    rr, _ = coordinates.cart_to_polar(xx, yy)
    out['cfg1_airy'] = ppsf.airydisk(rr, 100 / 10, HeNe)

    # from_amp_and_phase with OPD, phase_screen, thin_lens
    opd = polynomials.hopkins(0, 4, 0, r / 5, t, 1) * 500
    out['opd'] = opd
    wf2 = propagation.Wavefront.from_amp_and_phase(A, opd, 0.55, dx)
    out['fap_field'] = wf2.data
    out['phase_screen'] = propagation.Wavefront.phase_screen(opd, 0.55, dx).data
    out['thin_lens'] = propagation.Wavefront.thin_lens(250.0, 0.55, x, y).data
    out['xgrid'] = x
    out['ygrid'] = y
    f2 = wf2.focus(100, Q=2)
    out['fap_psf_intensity'] = f2.intensity.data
    gbar = rng.standard_normal(f2.data.shape)
    out['ibar'] = gbar
    out['intensity_adjoint'] = f2.intensity_adjoint(gbar).data
    wfbar = propagation.Wavefront(crandn(rng, wf2.data.shape), 0.55, dx)
    out['wfbar'] = wfbar.data
    out['fap_adjoint_phase'] = wf2.from_amp_and_phase_adjoint_phase(wfbar)
    # free_space through the object API
    fs = wf2.free_space(dz=5.0, Q=1)
    out['free_space'] = fs.data
    # polychromatic recipe (docs how-to), tiny: 5 wavelengths, MDFT 32^2
    wvls = np.linspace(0.5, 0.7, 5)
    weights = np.array([0.5, 1.0, 2.0, 1.0, 0.5])
    comps = []
    for w in wvls:
        wfl = propagation.Wavefront.from_amp_and_phase(A, opd, w, dx)
        ex = wfl.prepare_executor(100, 0.55 * 10 / 4, 32)
        comps.append(wfl.focus_dft(ex).intensity.data)
    comps = np.asarray(comps)
    out['poly_wvls'] = wvls
    out['poly_weights'] = weights
    out['poly_sum'] = polynomials.sum_of_2d_modes(comps, weights)
    out['poly_fdx'] = np.float64(0.55 * 10 / 4)
    # otf / convolution (next rows)
    psfi = psf.intensity.data
    data, df = potf.transform_psf(psfi, psf.dx)
    out['otf_transform'] = data
    mtf = potf.mtf_from_psf(psfi, psf.dx)
    out['otf_mtf'] = mtf.data
    obj = rng.standard_normal((64, 64))
    ker = rng.standard_normal((64, 64))
    out['conv_obj'] = obj
    out['conv_psf'] = ker
    out['conv_out'] = pconv.conv(obj, ker)
    np.savez_compressed(os.path.join(HERE, 'wavefront.npz'), **out)


def coronagraph():
    """to_fpm_and_back / babinet and their adjoints through the reference."""
    out = {}
    rng = np.random.default_rng(29)
    n, m = 48, 40
    x = crandn(rng, (n, n))
    ex = propagation.prepare_executor(0.1, (n, n), 1.2, (m, m), HeNe, 60.0)
    fpm = crandn(rng, (m, m))
    fpm_real = rng.random((m, m))
    lyot = (rng.random((n, n)) > 0.3).astype(float)
    g = crandn(rng, (n, n))
    out['x'], out['fpm'], out['fpm_real'], out['lyot'], out['g'] = x, fpm, fpm_real, lyot, g
    out['par'] = np.array([0.1, 1.2, HeNe, 60.0], dtype=np.float64)
    nxt, at_fpm, after = propagation.to_fpm_and_back(x, fpm, ex, return_more=True)
    out['tfab'], out['tfab_at'], out['tfab_after'] = nxt, at_fpm, after
    Ea, fbar = propagation.to_fpm_and_back_adjoint(g, fpm, ex, return_fpm_grad=True, field_at_fpm=at_fpm)
    out['tfab_adj'], out['tfab_fpmbar'] = Ea, fbar
    out['babinet'] = propagation.babinet(x, lyot, fpm_real, ex)
    out['babinet_adj'] = propagation.babinet_adjoint(g, lyot, fpm_real, ex)
    vm = propagation.vortex_phase_mask(2)
    xf, yf = np.meshgrid(np.linspace(-3, 3.1, 16), np.linspace(-2, 2.2, 12))
    out['vortex_xf'], out['vortex_yf'], out['vortex'] = xf, yf, vm(xf, yf)
    np.savez_compressed(os.path.join(HERE, 'coronagraph.npz'), **out)


def next_rows():
    """apply_transfer_functions, fourier_resample, jones_adapter."""
    from prysm.x import polarization as ppol
    out = {}
    rng = np.random.default_rng(8604)
    # apply_transfer_functions: array tfs (the DM render path, x/dm.py:254) and callable tfs, both shift modes
    obj = rng.standard_normal((24, 32))
    tf1 = crandn(rng, (24, 32))
    tf2 = rng.standard_normal((24, 32))
    out['atf_obj'] = obj
    out['atf_tf1'] = tf1
    out['atf_tf2'] = tf2
    out['atf_arrays_noshift'] = pconv.apply_transfer_functions(obj, None, [tf1, tf2], shift=False)
    out['atf_arrays_shift'] = pconv.apply_transfer_functions(obj, None, [tf1, tf2], shift=True)
    cobj = crandn(rng, (17, 20))
    out['atf_cobj'] = cobj

    def gauss(fr):
        return np.exp(-(fr / 3.0) ** 2)

    def ramp(fx, fy):
        return np.exp(-2j * np.pi * (0.01 * fx + 0.02 * fy))

    out['atf_callable_noshift'] = pconv.apply_transfer_functions(cobj, 0.05, [gauss, ramp], shift=False)
    out['atf_callable_shift'] = pconv.apply_transfer_functions(cobj, 0.05, [gauss, ramp], shift=True)
    # fourier_resample: real and complex, up / down, anisotropic zoom
    f = rng.standard_normal((16, 20))
    out['fr_f'] = f
    out['fr_up2'] = fttools.fourier_resample(f, 2)
    out['fr_up15'] = fttools.fourier_resample(f, 1.5)
    out['fr_aniso'] = fttools.fourier_resample(f, (0.75, 1.25))
    g = crandn(rng, (12, 9))
    out['fr_g'] = g
    out['fr_g_up'] = fttools.fourier_resample(g, 1.7)
    # jones_adapter on focus / angular_spectrum
    J = crandn(rng, (12, 16, 2, 2))
    out['jones_in'] = J
    out['jones_focus_Q2'] = ppol.jones_adapter(propagation.focus)(J, 2)
    out['jones_unfocus_Q1'] = ppol.jones_adapter(propagation.unfocus)(J, 1)
    out['jones_as'] = ppol.jones_adapter(propagation.angular_spectrum)(J, 0.6328, 0.01, 25.0, Q=2)
    np.savez_compressed(os.path.join(HERE, 'next_rows.npz'), **out)


def encircled():
    """otf.encircled_energy (+ adjoint) on an aberrated circular-pupil PSF and on a rectangular random PSF."""
    out = {}
    x, y = coordinates.make_xy_grid(64, diameter=10)
    r, t = coordinates.cart_to_polar(x, y)
    amp = geometry.circle(5, r)
    opd = polynomials.hopkins(0, 4, 0, r / 5, t, 1) * 150
    wf = propagation.Wavefront.from_amp_and_phase(amp, opd, HeNe, x[0, 1] - x[0, 0])
    psf = wf.focus(100, Q=2).intensity
    out['psf'] = psf.data
    out['psf_dx'] = psf.dx
    radii = np.array([1.0, 2.5, 5.0, 7.72, 10.0, 15.0, 20.0, 30.0, 40.0, 55.0])   # more than one 8-radius pass
    out['radii'] = radii
    out['ee_scalar'] = potf.encircled_energy(psf, psf.dx, 7.72)
    out['ee_many'] = potf.encircled_energy(psf.data, psf.dx, radii)
    rng = np.random.default_rng(346)
    bar = rng.standard_normal(radii.size)
    out['ee_bar'] = bar
    # the adjoint divides by |FT(psf)| (otf.py:238): a noiseless FFT PSF has exact zeros beyond the cutoff (NaN in the
    # reference), so the adjoint cases carry a small detector-like pedestal
    noisy = psf.data + 1e-4 * psf.data.max() * rng.random(psf.data.shape)
    out['psf_noisy'] = noisy
    out['ee_noisy'] = potf.encircled_energy(noisy, psf.dx, radii)
    out['ee_adj_many'] = potf.encircled_energy_adjoint(bar, psf=noisy, dx=psf.dx, radius=radii)
    out['ee_adj_scalar'] = potf.encircled_energy_adjoint(0.7, psf=noisy, dx=psf.dx, radius=12.0)
    rect = rng.random((40, 56))
    out['rect'] = rect
    out['rect_ee'] = potf.encircled_energy(rect, 0.8, radii[:3])
    out['rect_adj'] = potf.encircled_energy_adjoint(bar[:3], psf=rect, dx=0.8, radius=radii[:3])
    np.savez_compressed(os.path.join(HERE, 'encircled.npz'), **out)


def multires():
    """SURVEY 8(f) rank 2, second half: prepare_multiresolution + to_fpm_and_back_multiresolution(+adjoint), thin_lens_adjoint."""
    out = {}
    rng = np.random.default_rng(405)
    n = 40
    x = crandn(rng, (n, n))
    g = crandn(rng, (n, n))
    par = dict(pupil_dx=0.25, pupil_samples=(n, n), focal_dx=3.0, focal_samples=(24, 20), wavelength=HeNe, efl=80.0,
               num_levels=3, scaling=3.0, fine_samples=16, window=(0.25, 0.65))
    for kind in ('mdft', 'czt'):
        ex = propagation.prepare_multiresolution(kind=kind, **par)
        fpm = propagation.vortex_phase_mask(2)
        out[f'{kind}_fwd'] = propagation.to_fpm_and_back_multiresolution(x, fpm, ex)
        out[f'{kind}_adj'] = propagation.to_fpm_and_back_multiresolution_adjoint(g, fpm, ex)
    for k, w in enumerate(ex.windows):
        out[f'window{k}'] = np.asarray(w)
        out[f'xf{k}'] = ex.xf[k]
    out['x'], out['g'] = x, g
    out['par'] = np.array([0.25, 3.0, HeNe, 80.0, 3, 3.0, 16, 0.25, 0.65])
    # thin_lens_adjoint
    xx, yy = coordinates.make_xy_grid(32, diameter=8.0)
    Lbar = crandn(rng, (32, 32))
    out['tl_x'], out['tl_y'], out['tl_Lbar'] = xx, yy, Lbar
    out['tl_grad'] = np.asarray(propagation.Wavefront.thin_lens_adjoint(250.0, HeNe, xx, yy, Lbar))
    # otf adjoints (otf.py:205-316)
    psf = rng.random((20, 24)) + 0.1
    out['otf_psf'] = psf
    out['otf_mtf_bar'] = rng.standard_normal((20, 24))
    out['otf_ptf_bar'] = rng.standard_normal((20, 24))
    out['otf_otf_bar'] = crandn(rng, (20, 24))
    out['otf_mtf_adj'] = potf.mtf_from_psf_adjoint(out['otf_mtf_bar'], psf, 1.0)
    out['otf_ptf_adj'] = potf.ptf_from_psf_adjoint(out['otf_ptf_bar'], psf, 1.0)
    out['otf_otf_adj'] = potf.otf_from_psf_adjoint(out['otf_otf_bar'], psf, 1.0)
    # prepare_measured_fpm (coronagraph.py:128-200): a measured map on its own grid, resampled per level
    meas = (0.8 + 0.2 * rng.random((33, 28))) * np.exp(1j * rng.uniform(-np.pi, np.pi, (33, 28)))
    out['meas'] = meas
    xf, yf = np.meshgrid(np.linspace(-9.0, 9.5, 31), np.linspace(-11.0, 10.0, 26))
    out['meas_xf'], out['meas_yf'] = xf, yf
    for order in (0, 1):
        out[f'meas_o{order}_vortex'] = propagation.prepare_measured_fpm(meas, 0.6, center=(0.3, -0.2), charge=2, order=order)(xf, yf)
    out['meas_o1_one'] = propagation.prepare_measured_fpm(meas, 0.6)(xf, yf)
    out['meas_o1_fill'] = propagation.prepare_measured_fpm(meas, 0.6, fill=0.25 - 0.5j)(xf, yf)
    ex = propagation.prepare_multiresolution(kind='mdft', **par)
    out['meas_fwd'] = propagation.to_fpm_and_back_multiresolution(
        x, propagation.prepare_measured_fpm(meas, 0.6, center=(0.3, -0.2), charge=2), ex)
    for order in (2, 3, 4, 5):   # prefiltered spline orders (no further draws from rng: earlier entries stay as they were)
        out[f'meas_o{order}_vortex'] = propagation.prepare_measured_fpm(meas, 0.6, center=(0.3, -0.2), charge=2, order=order)(xf, yf)
    out['meas_o3_fwd'] = propagation.to_fpm_and_back_multiresolution(
        x, propagation.prepare_measured_fpm(meas, 0.6, center=(0.3, -0.2), charge=2, order=3), ex)
    np.savez_compressed(os.path.join(HERE, 'multires.npz'), **out)


def precision32():
    """fp32 path semantics (dtype propagation) on one case each."""
    out = {}
    rng = np.random.default_rng(23)
    config.precision = 32
    try:
        x = crandn(rng, (32, 32), np.complex64)
        out['x'] = x
        f = propagation.focus(x, 2)
        out['focus'] = f
        y = propagation.angular_spectrum(x, HeNe, 0.01, 10.0, Q=1)
        out['as'] = y
        ex = propagation.prepare_executor(0.1, (32, 32), 1.0, (16, 16), HeNe, 50.0)
        out['mdft'] = propagation.focus_dft(x, ex)
    finally:
        config.precision = 64
    np.savez_compressed(os.path.join(HERE, 'precision32.npz'), **out)


if __name__ == '__main__':
    if len(sys.argv) > 1:   # only the named groups
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    fft_family()
    padcrop()
    angular()
    executors()
    wavefront_and_physics()
    coronagraph()
    precision32()
    next_rows()
    multires()
    encircled()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)))
