"""GPU parity tests: the HIP path (through the C ABI, via prysm_amd) against
  (1) the golden fixtures generated from the reference itself (tests/golden/),
  (2) the CPU oracle on seeded inputs at sizes it finishes in seconds (up to 4096^2),
  (3) the identities the reference's own test-suite pins.

Tolerances (relative to the largest magnitude of the truth, always an fp64 numpy result):
  complex128 path : 1e-10     (north star: rtol 1e-5)
  complex64 path  : 5e-6 FFT family / 3e-5 matrix DFT   (north star: rtol 1e-3)
"""
import numpy as np
import pytest
import torch

from conftest import rel_max
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


def crandn(rng, shape, dtype=np.complex128):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)


# ----------------------------------------------------------------------------- golden fixtures

def test_fft_family_golden(pa, golden):
    P = pa.propagation
    g = golden('fft_family')
    for i in range(int(g['ncases'])):
        x, Q, gg = g[f'c{i}_x'], float(g[f'c{i}_Q']), g[f'c{i}_g']
        for name, arg in (('focus', x), ('unfocus', x), ('focus_adjoint', gg), ('unfocus_adjoint', gg)):
            got = tonp(getattr(P, name)(arg, Q))
            ref = g[f'c{i}_{name}']
            assert got.shape == ref.shape, (name, i)
            assert got.dtype == ref.dtype
            assert rel_max(got, ref) < TOL64, (name, i, x.shape, Q)
            got32 = tonp(getattr(P, name)(arg.astype(np.complex64), Q))
            assert got32.dtype == np.complex64
            assert rel_max(got32, ref) < TOL32, (name, i, x.shape, Q)


def test_padcrop_golden(pa, golden):
    F = pa.fttools
    g = golden('padcrop')
    cases = [((8, 8), 2, None), ((9, 9), 2, None), ((12, 12), 1.5, None), ((9, 12), 1.5, None),
             ((9, 12), None, (14, 18)), ((5, 8), None, (16, 16)), ((8, 5), 3, None)]
    for i, (shape, Q, oshape) in enumerate(cases):
        x = g[f'c{i}_x']
        p = F.pad2d(x, Q) if oshape is None else F.pad2d(x, out_shape=oshape)
        assert np.array_equal(tonp(p), g[f'c{i}_pad'])
        assert np.array_equal(tonp(F.crop_center(p, shape)), x)
    assert np.array_equal(tonp(F.pad2d(g['fill_x'], Q=2, value=1.5)), g['fill_pad'])
    assert F.pad2d(x, 1) is x


def test_angular_spectrum_golden(pa, golden):
    P = pa.propagation
    g = golden('angular_spectrum')
    for i in range(int(g['ncases'])):
        x = g[f'c{i}_x']
        Q, wvl, dx, z = (float(v) for v in g[f'c{i}_par'])
        y = tonp(P.angular_spectrum(x, wvl, dx, z, Q=Q))
        assert y.shape == g[f'c{i}_y'].shape
        assert rel_max(y, g[f'c{i}_y']) < TOL64
        tf = tonp(P.angular_spectrum_transfer_function(y.shape, wvl, dx, z))
        assert rel_max(tf, g[f'c{i}_tf']) < TOL64
        adj = tonp(P.angular_spectrum_adjoint(g[f'c{i}_g'], wvl, dx, z, Q=Q))
        assert adj.shape == g[f'c{i}_adj'].shape
        assert rel_max(adj, g[f'c{i}_adj']) < TOL64
        utf = g[f'c{i}_usertf']
        assert rel_max(tonp(P.angular_spectrum(x, wvl, dx, z, Q=Q, tf=utf)), g[f'c{i}_y_usertf']) < TOL64
        assert rel_max(tonp(P.angular_spectrum_adjoint(x, wvl, dx, z, Q=Q, tf=utf)), g[f'c{i}_adj_usertf']) < TOL64


def test_executors_golden(pa, golden):
    P = pa.propagation
    g = golden('executors')
    for i in range(int(g['ncases'])):
        x, gg = g[f'c{i}_x'], g[f'c{i}_g']
        ps, fs = tuple(int(v) for v in g[f'c{i}_ps']), tuple(int(v) for v in g[f'c{i}_fs'])
        pdx, fdx, wvl, efl, sx, sy = (float(v) for v in g[f'c{i}_par'])
        cx, cy, cfx, cfy = P.coordinates_for_focus(pdx, ps, fdx, fs, wvl, efl, (sx, sy))
        for a, b in ((cx, 'cx'), (cy, 'cy'), (cfx, 'cfx'), (cfy, 'cfy')):
            assert rel_max(tonp(a), g[f'c{i}_{b}']) < 1e-15
        for kind in ('mdft', 'czt'):
            ex = P.prepare_executor(pdx, ps, fdx, fs, wvl, efl, focal_shift=(sx, sy), kind=kind)
            assert ex.pupil_dx == pdx and ex.focal_dx == fdx
            assert rel_max(tonp(P.focus_dft(x, ex)), g[f'c{i}_{kind}_fwd']) < TOL64, (kind, i)
            assert rel_max(tonp(P.unfocus_dft(gg, ex)), g[f'c{i}_{kind}_adj']) < TOL64, (kind, i)
    x = g['fftdft_x']
    pdx, fdx, wvl, efl = (float(v) for v in g['fftdft_par'])
    samples = tuple(int(v) for v in g['fftdft_samples'])
    f = g['fftdft_fftfocus']
    for kind in ('mdft', 'czt', 'fftdft'):
        ex = P.prepare_executor(pdx, x.shape, fdx, samples, wvl, efl, kind=kind)
        assert rel_max(tonp(ex(x)), g[f'fftdft_{kind}_fwd']) < TOL64, kind
        assert rel_max(tonp(ex(x)), f) < 1e-9, kind              # FFT == MDFT == CZT == FFTDFT
        assert rel_max(tonp(ex.adjoint(f)), g[f'fftdft_{kind}_adj']) < TOL64, kind
    ex = P.prepare_executor(pdx, x.shape, fdx, (24, 40), wvl, efl, kind='fftdft')
    assert rel_max(tonp(ex(x)), g['fftdft_crop_fwd']) < TOL64
    assert rel_max(tonp(ex.adjoint(g['fftdft_crop_g'])), g['fftdft_crop_adj']) < TOL64


def test_wavefront_golden(pa, golden):
    P = pa.propagation
    W = P.Wavefront
    g = golden('wavefront')
    A, dx = g['cfg1_amp'], float(g['cfg1_dx'])
    # config 1 (plumbing): circular pupil, HeNe, focus(Q=2).intensity
    wf = W.from_amp_and_phase(A, None, O.HeNe, dx)
    psf = wf.focus(100, Q=2)
    assert psf.space == 'psf'
    assert psf.dx == float(g['cfg1_psf_dx'])
    assert rel_max(tonp(psf.intensity), g['cfg1_intensity']) < TOL64
    assert rel_max(tonp(wf.focus_intensity(100, Q=2)), g['cfg1_intensity']) < TOL64   # fused |.|^2 epilogue
    # synthesis kernels
    wf2 = W.from_amp_and_phase(A, g['opd'], 0.55, dx)
    assert rel_max(tonp(wf2.data), g['fap_field']) < TOL64
    assert rel_max(tonp(W.phase_screen(g['opd'], 0.55, dx).data), g['phase_screen']) < TOL64
    tl = W.thin_lens(250.0, 0.55, g['xgrid'], g['ygrid'])
    assert rel_max(tonp(tl.data), g['thin_lens']) < TOL64
    assert tl.dx == pytest.approx(dx)
    f2 = wf2.focus(100, Q=2)
    assert rel_max(tonp(f2.intensity), g['fap_psf_intensity']) < TOL64
    assert rel_max(tonp(f2.intensity_adjoint(g['ibar']).data), g['intensity_adjoint']) < TOL64
    wfbar = W(g['wfbar'], 0.55, dx)
    assert rel_max(tonp(wf2.from_amp_and_phase_adjoint_phase(wfbar)), g['fap_adjoint_phase']) < TOL64
    assert rel_max(tonp(wf2.free_space(dz=5.0, Q=1).data), g['free_space']) < TOL64
    # polychromatic recipe (docs how-to): per-wavelength MDFT + |.|^2 + weighted sum
    acc = None
    for w, wt in zip(g['poly_wvls'], g['poly_weights']):
        wfl = W.from_amp_and_phase(A, g['opd'], float(w), dx)
        ex = wfl.prepare_executor(100, float(g['poly_fdx']), 32)
        I = wfl.focus_dft(ex).intensity.data
        acc = I * float(wt) if acc is None else acc + I * float(wt)
    assert rel_max(tonp(acc), g['poly_sum']) < TOL64


def test_precision32_golden(pa, golden):
    """dtype propagation with config.precision = 32; truth is the fp64 oracle."""
    P = pa.propagation
    g = golden('precision32')
    x = g['x']
    x64 = x.astype(np.complex128)
    pa.config.precision = 32
    try:
        f = tonp(P.focus(x, 2))
        assert f.dtype == np.complex64
        assert rel_max(f, O.focus(x64, 2)) < TOL32
        y = tonp(P.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1))
        assert y.dtype == np.complex64
        assert rel_max(y, O.angular_spectrum(x64, O.HeNe, 0.01, 10.0, Q=1)) < TOL32
        ex = P.prepare_executor(0.1, (32, 32), 1.0, (16, 16), O.HeNe, 50.0)
        m = tonp(P.focus_dft(x, ex))
        assert m.dtype == np.complex64
        ref = O.prepare_executor(0.1, (32, 32), 1.0, (16, 16), O.HeNe, 50.0)(x64)
        assert rel_max(m, ref) < TOL32_MDFT
    finally:
        pa.config.precision = 64
    # with the default precision (64) a complex64 field is promoted, as in the reference
    assert tonp(P.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1)).dtype == np.complex128


# ----------------------------------------------------------------------------- oracle, seeded, larger sizes

@pytest.mark.parametrize('n,Q,dtype', [(512, 1, np.complex64), (512, 2, np.complex128), (1024, 1, np.complex128),
                                       (2048, 1, np.complex64), (1024, 2, np.complex64), (4096, 1, np.complex64),
                                       (4096, 1, np.complex128), (256, 1.5, np.complex128), (1000, 1, np.complex64)])
def test_focus_vs_oracle(pa, n, Q, dtype):
    """BASELINE configs 2 and the 4096^2 north-star path, plus padded / non power-of-two sizes."""
    rng = np.random.default_rng(n)
    x = crandn(rng, (n, n), dtype)
    ref = O.focus(x.astype(np.complex128), Q)
    got = tonp(pa.propagation.focus(x, Q))
    assert got.dtype == dtype and got.shape == ref.shape
    assert rel_max(got, ref) < (TOL32 if dtype == np.complex64 else TOL64)
    # fused intensity epilogue == |focus|^2
    I = tonp(pa.propagation.focus_intensity(x, Q))
    assert rel_max(I, O.intensity(ref)) < (2 * TOL32 if dtype == np.complex64 else TOL64)


@pytest.mark.parametrize('shape', [(512, 2048), (2048, 512), (64, 4096), (300, 512), (512, 300), (8192, 128), (64, 8192)])
def test_rectangular_unfocus_vs_oracle(pa, shape):
    rng = np.random.default_rng(sum(shape))
    x = crandn(rng, shape)
    assert rel_max(tonp(pa.propagation.unfocus(x, 1)), O.unfocus(x, 1)) < TOL64
    g = crandn(rng, shape)
    assert rel_max(tonp(pa.propagation.unfocus_adjoint(g, 1)), O.unfocus_adjoint(g, 1)) < TOL64


def test_angular_spectrum_config3(pa):
    """BASELINE config 3 geometry: 4096^2 complex128, HeNe, dx = 0.01 mm, z = 10 mm, Q = 1."""
    rng = np.random.default_rng(4096)
    x = crandn(rng, (4096, 4096))
    ref = O.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1)
    got = tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1))
    assert rel_max(got, ref) < TOL64
    x2 = crandn(rng, (1024, 1024))
    assert rel_max(tonp(pa.propagation.angular_spectrum(x2, 0.5, 0.005, 3.0, Q=2)),
                   O.angular_spectrum(x2, 0.5, 0.005, 3.0, Q=2)) < TOL64


def test_mdft_config4(pa):
    """BASELINE config 4: 2048^2 -> 512^2 fine-sampled PSF, complex64, vs the fp64 oracle."""
    P = pa.propagation
    rng = np.random.default_rng(2048)
    x = crandn(rng, (2048, 2048), np.complex64)
    pdx, efl, wvl = 10 / 2048, 100.0, O.HeNe
    fdx = wvl * 10 / 8
    ref_ex = O.prepare_executor(pdx, (2048, 2048), fdx, (512, 512), wvl, efl)
    ref = ref_ex(x.astype(np.complex128))
    pa.config.precision = 32
    try:
        ex = P.prepare_executor(pdx, (2048, 2048), fdx, (512, 512), wvl, efl)
        got = tonp(P.focus_dft(x, ex))
        assert got.dtype == np.complex64
        assert rel_max(got, ref) < TOL32_MDFT
        g = crandn(rng, (512, 512), np.complex64)
        adj = tonp(P.unfocus_dft(g, ex))
        assert rel_max(adj, ref_ex.adjoint(g.astype(np.complex128))) < TOL32_MDFT
        assert ex.nbytes() == 2 * 512 * 2048 * 8
    finally:
        pa.config.precision = 64
    ex64 = P.prepare_executor(pdx, (2048, 2048), fdx, (512, 512), wvl, efl)
    assert rel_max(tonp(ex64(x.astype(np.complex128))), ref) < TOL64
    # rectangular multiply-order rule (reference tests/test_fttools.py:55-85)
    xr = crandn(rng, (96, 40))
    for fs in ((20, 200), (200, 20)):
        a = P.prepare_executor(0.1, xr.shape, 1.0, fs, 0.6, 40.0, (0.2, -0.1))
        b = O.prepare_executor(0.1, xr.shape, 1.0, fs, 0.6, 40.0, (0.2, -0.1))
        assert a._forward_left_first == b._forward_left_first
        assert a._adjoint_left_first == b._adjoint_left_first
        assert rel_max(tonp(a(xr)), b(xr)) < TOL64
        gg = crandn(rng, fs)
        assert rel_max(tonp(a.adjoint(gg)), b.adjoint(gg)) < TOL64


def test_czt_fftdft_larger(pa):
    P = pa.propagation
    rng = np.random.default_rng(5)
    x = crandn(rng, (256, 200))
    for kind in ('czt',):
        a = P.prepare_executor(0.05, x.shape, 0.7, (120, 90), 0.6, 80.0, (1.0, -2.0), kind=kind)
        b = O.prepare_executor(0.05, x.shape, 0.7, (120, 90), 0.6, 80.0, (1.0, -2.0), kind='mdft')
        assert rel_max(tonp(a(x)), b(x)) < 1e-9
        g = crandn(rng, (120, 90))
        assert rel_max(tonp(a.adjoint(g)), b.adjoint(g)) < 1e-9


# ----------------------------------------------------------------------------- identities of the reference's tests

@pytest.mark.parametrize('Q', [1, 1.5, 2])
def test_adjoint_dot_products(pa, Q):
    """reference tests/test_propagation.py:32-55, 221-243."""
    P = pa.propagation
    rng = np.random.default_rng(789)
    x = crandn(rng, (9, 12))
    for fwd, adj in ((P.focus, P.focus_adjoint), (P.unfocus, P.unfocus_adjoint)):
        fx = tonp(fwd(x, Q))
        y = crandn(rng, fx.shape)
        np.testing.assert_allclose(np.vdot(fx, y), np.vdot(x, tonp(adj(y, Q))), atol=1e-11)
    fx = tonp(P.angular_spectrum(x, 0.55, 0.02, 3.0, Q=Q))
    y = crandn(rng, fx.shape)
    np.testing.assert_allclose(np.vdot(fx, y), np.vdot(x, tonp(P.angular_spectrum_adjoint(y, 0.55, 0.02, 3.0, Q=Q))),
                               atol=1e-11)


def test_roundtrip_unitarity_identity(pa):
    """reference tests/test_propagation.py:24-29, 210-218."""
    P = pa.propagation
    rng = np.random.default_rng(1)
    z = rng.random((128, 128))
    wf = P.Wavefront(dx=1, cmplx_field=z, wavelength=O.HeNe)
    wf2 = wf.focus(1, 1).unfocus(1, 1)
    assert np.allclose(tonp(wf2.data), z)
    x = crandn(rng, (4096, 4096), np.complex64)
    xd = pa.mathops.to_device(x)
    f = P.focus(xd, 1)
    e_in = float(torch.sum(pa.propagation.Wavefront(xd, 1, 1).intensity.data.double()))
    e_out = float(torch.sum(pa.propagation.Wavefront(f, 1, 1).intensity.data.double()))
    assert abs(e_out / e_in - 1) < 1e-5            # unitary at the full north-star size
    back = P.unfocus(f, 1)
    assert rel_max(tonp(back), x) < 2 * TOL32      # focus o unfocus = identity at 4096^2
    xs = crandn(rng, (16, 16))
    assert np.allclose(tonp(P.angular_spectrum(xs, 0.5, 0.01, 0.0, Q=1)), xs)


def test_fft_mdft_equivalent_wavefront(pa):
    """reference tests/test_propagation.py:98-117."""
    P = pa.propagation
    rng = np.random.default_rng(2)
    z = rng.random((32, 32))
    wf = P.Wavefront(dx=1, cmplx_field=z, wavelength=O.HeNe, space='pupil')
    focus_fft = wf.focus(Q=2, efl=1)
    mdft = wf.prepare_executor(efl=1, dx=focus_fft.dx, samples=tuple(focus_fft.data.shape))
    assert np.allclose(tonp(focus_fft.data), tonp(wf.focus_dft(mdft).data))
    wfp = P.Wavefront(dx=1, cmplx_field=rng.random((128, 128)), wavelength=O.HeNe, space='psf')
    unfocus_fft = wfp.unfocus(Q=2, efl=1)
    m2 = wfp.prepare_executor(efl=1, dx=unfocus_fft.dx, samples=tuple(unfocus_fft.data.shape))
    assert np.allclose(tonp(unfocus_fft.data), tonp(wfp.unfocus_dft(m2).data))


def test_airy_known_answer(pa):
    """reference tests/test_physics.py:20-34: FFT PSF slice == analytic Airy disk to 1e-3-class."""
    P = pa.propagation
    for efl, epd, wvl in ((10, 5, 0.55), (20, 10, 0.8), (100, 10, O.HeNe)):
        n = 256
        x, y = O.make_xy_grid(n, diameter=4 * epd)
        r, _ = O.cart_to_polar(x, y)
        amp = O.circle(epd / 2, r)
        wf = P.Wavefront.from_amp_and_phase(amp, None, wvl, float(x[0, 1] - x[0, 0]))
        psf = wf.focus(efl, Q=3)
        I = tonp(psf.intensity)
        I = I / I.max()
        xx, yy = O.make_xy_grid(I.shape, dx=psf.dx)
        rr, _ = O.cart_to_polar(xx, yy)
        airy = O.airydisk(rr, efl / epd, wvl)
        c = I.shape[0] // 2
        assert np.allclose(I[c, c:c + 40], airy[c, c:c + 40], atol=2e-3)


def test_otf_and_convolution_golden(pa, golden):
    """transform_psf / MTF / conv against the reference's outputs."""
    g = golden('wavefront')
    psf = g['cfg1_intensity']
    data, df = pa.otf.transform_psf(psf, float(g['cfg1_psf_dx']))
    assert rel_max(tonp(data), g['otf_transform']) < TOL64
    assert df == 1000 / (psf.shape[0] * float(g['cfg1_psf_dx']))
    mtf = pa.otf.mtf_from_psf(psf, float(g['cfg1_psf_dx']))
    assert rel_max(tonp(mtf), g['otf_mtf']) < TOL64
    mtf2, ptf2, otf2 = pa.otf.mtf_ptf_otf_from_psf(psf, float(g['cfg1_psf_dx']))
    assert rel_max(tonp(mtf2), g['otf_mtf']) < TOL64
    # adjoint dot-product test of the linear transform (reference tests/test_otf.py:56-100)
    rng = np.random.default_rng(3)
    y = crandn(rng, psf.shape)
    lhs = np.vdot(tonp(data), y)
    rhs = np.vdot(psf.astype(np.complex128), tonp(pa.otf.transform_psf_adjoint(y)))
    np.testing.assert_allclose(lhs, rhs, rtol=1e-11)
    out = pa.convolution.conv(g['conv_obj'], g['conv_psf'])
    assert not out.is_complex()
    assert rel_max(tonp(out), g['conv_out']) < TOL64


def test_coronagraph_golden(pa, golden):
    """to_fpm_and_back / babinet (+ adjoints), compositions of the MFMA executors."""
    P = pa.propagation
    g = golden('coronagraph')
    pdx, fdx, wvl, efl = (float(v) for v in g['par'])
    x, fpm = g['x'], g['fpm']
    ex = P.prepare_executor(pdx, x.shape, fdx, fpm.shape, wvl, efl)
    nxt, at_fpm, after = P.to_fpm_and_back(x, fpm, ex, return_more=True)
    assert rel_max(tonp(nxt), g['tfab']) < TOL64
    assert rel_max(tonp(at_fpm), g['tfab_at']) < TOL64
    assert rel_max(tonp(after), g['tfab_after']) < TOL64
    Ea, fbar = P.to_fpm_and_back_adjoint(g['g'], fpm, ex, return_fpm_grad=True, field_at_fpm=at_fpm)
    assert rel_max(tonp(Ea), g['tfab_adj']) < TOL64
    assert rel_max(tonp(fbar), g['tfab_fpmbar']) < TOL64
    assert rel_max(tonp(P.babinet(x, g['lyot'], g['fpm_real'], ex)), g['babinet']) < TOL64
    assert rel_max(tonp(P.babinet_adjoint(g['g'], g['lyot'], g['fpm_real'], ex)), g['babinet_adj']) < TOL64
    assert rel_max(tonp(P.vortex_phase_mask(2)(g['vortex_xf'], g['vortex_yf'])), g['vortex']) < 1e-12
    wf = P.Wavefront(x, wvl, pdx)
    assert rel_max(tonp(wf.babinet(g['lyot'], g['fpm_real'], ex).data), g['babinet']) < TOL64
    with pytest.raises(TypeError):
        P.vortex_phase_mask(1.5)
    with pytest.raises(ValueError):
        P.to_fpm_and_back_adjoint(g['g'], fpm, ex, return_fpm_grad=True)


def test_polychromatic_driver_single_gpu(pa):
    """BASELINE config 5 recipe at a small size on one GPU (world size 1: no process group, no reduce);
    the N > 1 sharding / reduce logic is covered on CPU by tests/test_distributed_cpu.py."""
    from prysm_amd.polychromatic import polychromatic_psf
    n = 128
    x, y = O.make_xy_grid(n, diameter=10)
    r, _ = O.cart_to_polar(x, y)
    amp = O.circle(5, r)
    opd = O.hopkins_w040(r / 5, 500.0)
    dx = float(x[0, 1] - x[0, 0])
    wvls = np.linspace(0.5, 0.7, 6)
    wts = np.array([0.5, 1.0, 1.5, 1.5, 1.0, 0.5])
    # variant M (the how-to): fixed focal grid, MDFT per wavelength
    got = tonp(polychromatic_psf(amp, opd, wvls, wts, dx, 100.0, focal_dx=0.55 * 10 / 4, samples=64, kind='mdft'))
    comps = []
    for w in wvls:
        P = O.from_amp_and_phase(amp, opd, float(w))
        comps.append(O.intensity(O.prepare_executor(dx, P.shape, 0.55 * 10 / 4, (64, 64), float(w), 100.0)(P)))
    assert rel_max(got, O.sum_of_2d_modes(np.asarray(comps), wts)) < TOL64
    # variant F (throughput): FFT focus per wavelength with the fused |.|^2 accumulate epilogue
    comps = [O.intensity(O.focus(O.from_amp_and_phase(amp, opd, float(w)), 2)) for w in wvls]
    want = O.sum_of_2d_modes(np.asarray(comps), wts)
    got = tonp(polychromatic_psf(amp, opd, wvls, wts, dx, 100.0, Q=2, batched=True))     # stacks + one weighted sum
    assert rel_max(got, want) < TOL64
    got = tonp(polychromatic_psf(amp, opd, wvls, wts, dx, 100.0, Q=2, batched=False))    # field by field, accumulate epilogue
    assert rel_max(got, want) < TOL64


def test_sum_modes_vs_oracle(pa):
    """pm_sum_modes = polynomials.sum_of_2d_modes (tensordot of weights and modes), incl. > 32 modes and accumulate."""
    from prysm_amd import _ops
    rng = np.random.default_rng(9)
    for B, dt in ((3, np.float64), (40, np.float32), (1, np.float64)):
        modes = rng.standard_normal((B, 37, 53)).astype(dt)
        w = rng.standard_normal(B)
        got = tonp(_ops.sum_modes(torch.from_numpy(modes).cuda(), w))
        want = O.sum_of_2d_modes(modes.astype(np.float64), w)
        assert rel_max(got, want) < (1e-13 if dt == np.float64 else 2e-6)
        acc = torch.ones((37, 53), dtype=torch.from_numpy(modes).dtype, device='cuda')
        got = tonp(_ops.sum_modes(torch.from_numpy(modes).cuda(), w, out=acc, accumulate=True))
        assert rel_max(got, want + 1.0) < (1e-13 if dt == np.float64 else 2e-6)


def test_errors_match_reference(pa):
    P = pa.propagation
    z = np.ones((8, 8), dtype=np.complex128)
    with pytest.raises(ValueError):
        P.Wavefront(z, 0.5, 1, space='psf').focus(1)
    with pytest.raises(ValueError):
        P.Wavefront(z, 0.5, 1, space='pupil').unfocus(1)
    with pytest.raises(ValueError):
        P.prepare_executor(0.1, 8, 1.0, 8, 0.5, 10, kind='nope')
    with pytest.raises(ValueError):
        P.Wavefront(z, 0.5, 1).free_space()
    with pytest.raises(TypeError):
        P.Wavefront(z, 0.5, 1) * 'a'
    with pytest.raises(ValueError):
        P.Wavefront(z, 0.5, 1) * P.Wavefront(z, 0.6, 1)
    with pytest.raises(ValueError):
        pa.fttools.CZT(np.arange(4.), np.arange(4.), np.arange(4.), np.arange(4.), sign=2)


# ---------------------------------------------------------------- batches of fields (one launch pair)
@pytest.mark.parametrize('n,Q,dtype,B', [(256, 1, np.complex128, 5), (512, 2, np.complex64, 3), (96, 1.5, np.complex128, 4),
                                        (1024, 1, np.complex64, 9)])
def test_batched_focus_family_vs_oracle(pa, n, Q, dtype, B):
    """A (B, n, n) stack through focus / unfocus / their adjoints / focus_intensity equals the oracle field by field
    (power-of-two sizes run grid.y = B; 96 * 1.5 = 144 goes through the direct-DFT kernels field by field)."""
    P = pa.propagation
    rng = np.random.default_rng(n + B)
    x = crandn(rng, (B, n, n), dtype)
    tol = TOL64 if dtype == np.complex128 else TOL32
    for name in ('focus', 'unfocus'):
        got = tonp(getattr(P, name)(x, Q))
        assert got.shape[0] == B and got.dtype == dtype
        for b in range(B):
            assert rel_max(got[b], getattr(O, name)(x[b], Q)) < tol, (name, b)
    M = got.shape[-1]
    g = crandn(rng, (B, M, M), dtype)
    got = tonp(P.focus_adjoint(g, Q))
    for b in range(B):
        assert rel_max(got[b], O.focus_adjoint(g[b], Q)) < tol
    got = tonp(P.focus_intensity(x, Q))
    for b in range(B):
        assert rel_max(got[b], O.intensity(O.focus(x[b], Q))) < 4 * tol


def test_batched_equals_single_bitwise_and_chunking(pa):
    """The batch is the same arithmetic as B single calls: results are bit-identical, whatever the chunking of
    the batch over the workspace budget."""
    from prysm_amd import _lib
    P = pa.propagation
    rng = np.random.default_rng(77)
    x = torch.from_numpy(crandn(rng, (7, 512, 512), np.complex64)).cuda()
    single = torch.stack([P.focus(x[b], 1) for b in range(7)])
    lib = _lib.load()
    try:
        for mib in (1, 4, 64):
            lib.pm_set_tuning(b'batch_ws_mib', mib)
            assert torch.equal(P.focus(x, 1), single), mib
    finally:
        lib.pm_set_tuning(b'batch_ws_mib', 128)
    # strided views of a bigger stack (every other field)
    assert torch.equal(P.focus(x[::2], 1), single[::2])


def test_batched_angular_spectrum_per_field_wavelengths(pa):
    """Stack of fields with one wavelength each (per-field separable transfer functions) -- the polychromatic
    free-space step -- equals the oracle's angular_spectrum wavelength by wavelength; fused 3-pass path and the
    two-call path (non power-of-two)."""
    P = pa.propagation
    rng = np.random.default_rng(5)
    wvls = [0.5, 0.6, 0.7]
    for n, Q in ((256, 1), (128, 2), (100, 1)):
        x = crandn(rng, (3, n, n))
        got = tonp(P.angular_spectrum(x, wvls, 0.01, 50.0, Q=Q))
        for b, w in enumerate(wvls):
            assert rel_max(got[b], O.angular_spectrum(x[b], w, 0.01, 50.0, Q=Q)) < TOL64
        got = tonp(P.angular_spectrum(x, 0.55, 0.01, [10.0, 20.0, 30.0], Q=Q))
        for b, z in enumerate([10.0, 20.0, 30.0]):
            assert rel_max(got[b], O.angular_spectrum(x[b], 0.55, 0.01, z, Q=Q)) < TOL64
    with pytest.raises(ValueError):
        P.angular_spectrum(x, [0.5, 0.6], 0.01, 50.0)


# ---------------------------------------------------------------- SURVEY 8(f) ranks 3-4 and the real-input path
def test_next_rows_golden(pa, golden):
    """apply_transfer_functions, fourier_resample, jones_adapter against the reference's own outputs."""
    from prysm_amd import convolution as C, fttools as F
    from prysm_amd.x.polarization import jones_adapter
    P = pa.propagation
    g = golden('next_rows')
    obj, tf1, tf2 = g['atf_obj'], g['atf_tf1'], g['atf_tf2']
    got = tonp(C.apply_transfer_functions(obj, None, [tf1, tf2], shift=False))
    assert not np.iscomplexobj(got) and rel_max(got, g['atf_arrays_noshift']) < TOL64
    assert rel_max(tonp(C.apply_transfer_functions(obj, None, [tf1, tf2], shift=True)), g['atf_arrays_shift']) < TOL64

    def gauss(fr):
        return torch.exp(-(fr / 3.0) ** 2)

    def ramp(fx, fy):
        return torch.exp(-2j * np.pi * (0.01 * fx + 0.02 * fy))

    cobj = g['atf_cobj']
    assert rel_max(tonp(C.apply_transfer_functions(cobj, 0.05, [gauss, ramp], shift=False)), g['atf_callable_noshift']) < TOL64
    assert rel_max(tonp(C.apply_transfer_functions(cobj, 0.05, [gauss, ramp], shift=True)), g['atf_callable_shift']) < TOL64
    f, gg = g['fr_f'], g['fr_g']
    for zoom, key in ((2, 'fr_up2'), (1.5, 'fr_up15'), ((0.75, 1.25), 'fr_aniso')):
        got = tonp(F.fourier_resample(f, zoom))
        assert got.shape == g[key].shape and not np.iscomplexobj(got)
        assert rel_max(got, g[key]) < TOL64
    assert rel_max(tonp(F.fourier_resample(gg, 1.7)), g['fr_g_up']) < TOL64
    assert F.fourier_resample(f, 1) is f
    with pytest.raises(ValueError):
        F.fourier_resample(f, (1, -1))
    J = g['jones_in']
    assert rel_max(tonp(jones_adapter(P.focus)(J, 2)), g['jones_focus_Q2']) < TOL64
    assert rel_max(tonp(jones_adapter(P.unfocus)(J, 1)), g['jones_unfocus_Q1']) < TOL64
    assert rel_max(tonp(jones_adapter(P.angular_spectrum)(J, 0.6328, 0.01, 25.0, Q=2)), g['jones_as']) < TOL64
    assert rel_max(tonp(jones_adapter(P.focus)(J[..., 0, 1], 2)), g['jones_focus_Q2'][..., 0, 1]) < TOL64   # 2-D passes through


@pytest.mark.parametrize('shape,dtype', [((512, 512), np.float32), ((256, 1024), np.float64), ((90, 120), np.float64),
                                         ((2048, 2048), np.float32)])
def test_real_input_paths_vs_oracle(pa, shape, dtype):
    """float32 / float64 fields are read as they are (PM_FLAG_REAL_INPUT): focus, transform_psf, conv and the batch form
    equal the oracle on the same real arrays (engine sizes and the direct-DFT sizes)."""
    from prysm_amd import otf, convolution as C
    P = pa.propagation
    rng = np.random.default_rng(shape[0] + shape[1])
    a = rng.standard_normal(shape).astype(dtype)
    b = rng.standard_normal(shape).astype(dtype)
    tol = TOL64 if dtype == np.float64 else TOL32
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    got = tonp(P.focus(a, 1))
    assert got.dtype == cdt and rel_max(got, O.focus(a, 1)) < tol
    if shape[0] <= 512:
        assert rel_max(tonp(P.focus(a, 2)), O.focus(a, 2)) < tol
        assert rel_max(tonp(P.focus(np.stack([a, b]), 1)[1]), O.focus(b, 1)) < tol
    data, df = otf.transform_psf(a, 1.0)
    assert rel_max(tonp(data), O.transform_psf(a)) < tol
    got = tonp(C.conv(a, b))
    assert got.dtype == dtype and rel_max(got, O.conv(a.astype(np.float64), b.astype(np.float64))) < 4 * tol
    assert rel_max(tonp(P.angular_spectrum(a, 0.6, 0.01, 20.0, Q=1)), O.angular_spectrum(a.astype(np.float64), 0.6, 0.01, 20.0, Q=1)) < 4 * tol


def test_hipgraph_capture_of_a_model(pa):
    """A chain (pupil synthesis -> focus Q=2 -> |.|^2 -> OTF) captured into a hipGraph replays to the same bits as
    the eager calls, on new inputs, and the library issues nothing that breaks capture (no sync, no allocation)."""
    from prysm_amd import graph, otf
    P = pa.propagation
    rng = np.random.default_rng(31)
    amp = (rng.random((256, 256)) > 0.3).astype(np.float64)

    def model(a, opd):
        wf = P.Wavefront.from_amp_and_phase(a, opd, 0.6328, 0.04)
        psf = wf.focus(100.0, Q=2).intensity
        mtf = otf.mtf_from_psf(psf)
        return psf.data, mtf.data

    opd0 = rng.standard_normal((256, 256)) * 50
    m = graph.capture(model, amp, opd0)
    for seed in (1, 2):
        opd = np.random.default_rng(seed).standard_normal((256, 256)) * 80
        psf_g, mtf_g = [t.clone() for t in m(amp, opd)]
        psf_e, mtf_e = model(torch.from_numpy(amp).cuda(), torch.from_numpy(opd).cuda())
        assert torch.equal(psf_g, psf_e) and torch.equal(mtf_g, mtf_e)
        want = O.intensity(O.focus(O.from_amp_and_phase(amp, opd, 0.6328), 2))
        assert rel_max(tonp(psf_g), want) < TOL64
    with pytest.raises(ValueError):
        m(amp)


@pytest.mark.parametrize('shape,dtype,forced', [((8192, 2048), np.complex64, False), ((4096, 2048), np.complex128, False),
                                                ((64, 2048), np.complex64, True), ((256, 4096), np.complex128, True)])
def test_folded_column_transform_vs_oracle(pa, shape, dtype, forced):
    """The radix-2 step of the column transform folded into the row pass (8192-point columns; 4096-point complex128
    columns; forced by the tuning knob elsewhere): focus / unfocus / |.|^2 / even crops equal the oracle, and the
    folded and unfolded paths agree."""
    from prysm_amd import _lib
    P = pa.propagation
    lib = _lib.load()
    rng = np.random.default_rng(shape[0] * 3 + shape[1])
    x = crandn(rng, shape, dtype)
    tol = TOL64 if dtype == np.complex128 else TOL32
    try:
        if forced:
            lib.pm_set_tuning(b'fold', 1)
        got = tonp(P.focus(x, 1))
        assert rel_max(got, O.focus(x, 1)) < tol
        assert rel_max(tonp(P.unfocus(x, 1)), O.unfocus(x, 1)) < tol
        assert rel_max(tonp(P.focus_intensity(x, 1)), O.intensity(O.focus(x, 1))) < 4 * tol
        assert rel_max(tonp(P.focus_adjoint(x, 2)), O.focus_adjoint(x, 2)) < tol      # even crop of the output
        lib.pm_set_tuning(b'fold', 0)
        unfolded = tonp(P.focus(x, 1))
        assert rel_max(got, unfolded) < 4 * tol
    finally:
        lib.pm_set_tuning(b'fold', -1)


@pytest.mark.parametrize('shape,dtype', [((64, 2048), np.complex128), ((128, 4096), np.complex64), ((4096, 2048), np.complex128)])
def test_folded_fused_angular_spectrum_vs_oracle(pa, shape, dtype):
    """The folded 3-pass chain (fold in the first row pass, M/2-point column FFT x H x IFFT per plane, unfold in the
    last row pass): angular spectrum (separable H), its adjoint (conj H), a full tf= array and conv (rotations)."""
    from prysm_amd import _lib, convolution as C
    from prysm_amd.conf import config
    P = pa.propagation
    lib = _lib.load()
    rng = np.random.default_rng(shape[0] + 7 * shape[1])
    x = crandn(rng, shape, dtype)
    tol = TOL64 if dtype == np.complex128 else 4 * TOL32
    prec = config.precision
    try:
        config.precision = 64 if dtype == np.complex128 else 32
        lib.pm_set_tuning(b'fold', 1)
        xo = x.astype(np.complex128)
        assert rel_max(tonp(P.angular_spectrum(x, 0.6, 0.01, 30.0, Q=1)), O.angular_spectrum(xo, 0.6, 0.01, 30.0, Q=1)) < tol
        assert rel_max(tonp(P.angular_spectrum_adjoint(x, 0.6, 0.01, 30.0, Q=1)),
                       O.angular_spectrum_adjoint(xo, 0.6, 0.01, 30.0, Q=1)) < tol
        if shape[0] <= 128:
            tf = crandn(rng, shape, dtype)
            assert rel_max(tonp(P.angular_spectrum(x, 0.6, 0.01, 30.0, Q=1, tf=tf)),
                           O.angular_spectrum(xo, 0.6, 0.01, 30.0, Q=1, tf=tf.astype(np.complex128))) < tol
            h = crandn(rng, shape, dtype)
            assert rel_max(tonp(C.conv(x, h)), O.conv(xo, h.astype(np.complex128))) < 4 * tol
    finally:
        lib.pm_set_tuning(b'fold', -1)
        config.precision = prec


def test_multiresolution_golden(pa, golden):
    """prepare_multiresolution + to_fpm_and_back_multiresolution(+adjoint) (mdft and czt levels), Wavefront wrappers,
    the adjoint dot-product identity and thin_lens_adjoint against the reference's outputs."""
    P = pa.propagation
    g = golden('multires')
    x, gg = g['x'], g['g']
    n = x.shape[0]
    wvl = float(g['par'][2])
    for kind in ('mdft', 'czt'):
        ex = P.prepare_multiresolution(0.25, (n, n), 3.0, (24, 20), wvl, 80.0, 3, scaling=3.0, fine_samples=16,
                                       window=(0.25, 0.65), kind=kind)
        assert len(ex) == 3
        fpm = P.vortex_phase_mask(2)
        fwd = P.to_fpm_and_back_multiresolution(x, fpm, ex)
        adj = P.to_fpm_and_back_multiresolution_adjoint(gg, fpm, ex)
        assert rel_max(tonp(fwd), g[f'{kind}_fwd']) < TOL64
        assert rel_max(tonp(adj), g[f'{kind}_adj']) < TOL64
        lhs = np.vdot(tonp(fwd), gg)
        rhs = np.vdot(x, tonp(adj))
        assert abs(lhs - rhs) < 1e-10 * abs(lhs)
    for k in range(3):
        assert rel_max(tonp(ex.windows[k]), g[f'window{k}']) < 1e-13
        assert rel_max(tonp(ex.xf[k]), g[f'xf{k}']) < 1e-13
    wf = P.Wavefront(x, wvl, 0.25)
    ex = wf.prepare_multiresolution(efl=80.0, focal_dx=3.0, focal_samples=(24, 20), num_levels=3, scaling=3.0, fine_samples=16,
                                    window=(0.25, 0.65))
    out, at_fpm, after_fpm = wf.to_fpm_and_back_multiresolution(P.vortex_phase_mask(2), ex, return_more=True)
    assert rel_max(tonp(out), g['mdft_fwd']) < TOL64 and len(at_fpm) == 3 and at_fpm[1].space == 'psf'
    Ea, fbars = P.Wavefront(gg, wvl, 0.25).to_fpm_and_back_multiresolution_adjoint(
        P.vortex_phase_mask(2), ex, return_fpm_grad=True, field_at_fpm=at_fpm)
    assert rel_max(tonp(Ea), g['mdft_adj']) < TOL64 and len(fbars) == 3
    with pytest.raises(ValueError):
        P.to_fpm_and_back_multiresolution_adjoint(gg, P.vortex_phase_mask(2), ex, return_fpm_grad=True)
    got = float(P.Wavefront.thin_lens_adjoint(250.0, wvl, g['tl_x'], g['tl_y'], g['tl_Lbar']))
    assert abs(got - float(g['tl_grad'])) < 1e-10 * abs(float(g['tl_grad']))


def test_measured_fpm_golden(pa, golden):
    """prepare_measured_fpm (coronagraph.py:128-200): pm_sample_map against scipy.ndimage.map_coordinates through the
    reference -- orders 0 / 1, vortex / constant / default continuation, fp32 coordinates, and inside a multiresolution
    round trip; orders 2 .. 5 through the prefiltered B-spline path."""
    P = pa.propagation
    g = golden('multires')
    meas, xf, yf = g['meas'], g['meas_xf'], g['meas_yf']
    for order in (0, 1, 2, 3, 4, 5):   # 2 .. 5: pm_spline_prefilter (scipy's padded recursive prefilter) + pm_sample_spline
        got = P.prepare_measured_fpm(meas, 0.6, center=(0.3, -0.2), charge=2, order=order)(xf, yf)
        assert got.dtype == torch.complex128
        assert rel_max(tonp(got), g[f'meas_o{order}_vortex']) < (1e-12 if order < 4 else 1e-11)
    assert rel_max(tonp(P.prepare_measured_fpm(meas, 0.6)(xf, yf)), g['meas_o1_one']) < 1e-12
    assert rel_max(tonp(P.prepare_measured_fpm(meas, 0.6, fill=0.25 - 0.5j)(xf, yf)), g['meas_o1_fill']) < 1e-12
    # coordinate vectors broadcast (stride 0), fp32 map
    got = P.prepare_measured_fpm(meas.astype(np.complex64), 0.6, fill=0.25 - 0.5j)(xf[:1, :], yf[:, :1])
    assert got.dtype == torch.complex64 and tuple(got.shape) == xf.shape
    assert rel_max(tonp(got), g['meas_o1_fill']) < 2e-4      # coordinates in fp32: the position error times the map slope
    n = g['x'].shape[0]
    ex = P.prepare_multiresolution(0.25, (n, n), 3.0, (24, 20), float(g['par'][2]), 80.0, 3, scaling=3.0, fine_samples=16,
                                   window=(0.25, 0.65))
    fpm = P.prepare_measured_fpm(meas, 0.6, center=(0.3, -0.2), charge=2)
    assert rel_max(tonp(P.to_fpm_and_back_multiresolution(g['x'], fpm, ex)), g['meas_fwd']) < TOL64
    fpm3 = P.prepare_measured_fpm(meas, 0.6, center=(0.3, -0.2), charge=2, order=3)
    assert rel_max(tonp(P.to_fpm_and_back_multiresolution(g['x'], fpm3, ex)), g['meas_o3_fwd']) < TOL64
    got = P.prepare_measured_fpm(meas.astype(np.complex64), 0.6, fill=0.25 - 0.5j, order=3)(xf[:1, :], yf[:, :1])
    want = O.prepare_measured_fpm(meas, 0.6, fill=0.25 - 0.5j, order=3)(xf, yf)
    assert got.dtype == torch.complex64 and rel_max(tonp(got), want) < 5e-4   # fp32 coordinates x the (steeper) cubic slope
    with pytest.raises(RuntimeError):
        P.prepare_measured_fpm(meas, 0.6, order=6)


def test_otf_adjoints_golden(pa, golden):
    """mtf / ptf / otf_from_psf_adjoint (otf.py:205-316) against the reference, with and without the reused transform."""
    from prysm_amd import otf
    g = golden('multires')
    psf = g['otf_psf']
    assert rel_max(tonp(otf.mtf_from_psf_adjoint(g['otf_mtf_bar'], psf, 1.0)), g['otf_mtf_adj']) < TOL64
    assert rel_max(tonp(otf.ptf_from_psf_adjoint(g['otf_ptf_bar'], psf, 1.0)), g['otf_ptf_adj']) < TOL64
    assert rel_max(tonp(otf.otf_from_psf_adjoint(g['otf_otf_bar'], psf, 1.0)), g['otf_otf_adj']) < TOL64
    _, data = otf.mtf_from_psf(psf, 1.0, return_more=True)
    assert rel_max(tonp(otf.mtf_from_psf_adjoint(g['otf_mtf_bar'], data=data)), g['otf_mtf_adj']) < TOL64


def test_encircled_energy_golden_and_oracle(pa, golden):
    """otf.encircled_energy (+ adjoint; pm_encircled_energy: Bessel-kernel reduction over the MTF, 8 radii per pass) against
    the reference's outputs, and against the oracle on a larger PSF in both precisions."""
    from prysm_amd import otf
    g = golden('encircled')
    dx, radii = float(g['psf_dx']), g['radii']
    assert abs(float(tonp(otf.encircled_energy(g['psf'], dx, 7.72))) - float(g['ee_scalar'])) < 1e-11
    assert rel_max(tonp(otf.encircled_energy(g['psf'], dx, radii)), g['ee_many']) < TOL64
    ee, data = otf.encircled_energy(g['psf_noisy'], dx, radii, return_more=True)
    assert rel_max(tonp(ee), g['ee_noisy']) < TOL64
    assert rel_max(tonp(otf.encircled_energy_adjoint(g['ee_bar'], psf=g['psf_noisy'], dx=dx, radius=radii)), g['ee_adj_many']) < 1e-9
    assert rel_max(tonp(otf.encircled_energy_adjoint(g['ee_bar'], dx=dx, radius=radii, data=data)), g['ee_adj_many']) < 1e-9
    assert rel_max(tonp(otf.encircled_energy_adjoint(0.7, psf=g['psf_noisy'], dx=dx, radius=12.0)), g['ee_adj_scalar']) < 1e-9
    assert rel_max(tonp(otf.encircled_energy(g['rect'], 0.8, radii[:3])), g['rect_ee']) < TOL64
    assert rel_max(tonp(otf.encircled_energy_adjoint(g['ee_bar'][:3], psf=g['rect'], dx=0.8, radius=radii[:3])), g['rect_adj']) < 1e-9
    # a 1024^2 PSF (config-1 style pupil, Q = 2), float64 and float32
    x = np.arange(-256, 256) * (10 / 512)
    r = np.hypot(*np.meshgrid(x, x))
    psf = np.abs(O.focus((r <= 5).astype(np.complex128), 2)) ** 2
    psf_dx = 100 * O.HeNe / (10 / 512 * 1024)
    rr = [2.0, 7.72, 15.0, 40.0]
    want = O.encircled_energy(psf, psf_dx, rr)
    assert rel_max(tonp(otf.encircled_energy(psf, psf_dx, rr)), want) < TOL64
    got32 = tonp(otf.encircled_energy(psf.astype(np.float32), psf_dx, rr))
    assert rel_max(got32, want) < 1e-5
    with pytest.raises(ValueError):
        otf.encircled_energy_adjoint(1.0, radius=3.0, data=data)   # dx is required with data (otf.py:451-452)


def test_fft_facade_numpy_semantics(pa):
    """The module-like fft object handed to prysm's BackendShim: numpy semantics (norm modes, n / axis, odd-length shifts)."""
    from prysm_amd.mathops import FFTFacade
    fft = FFTFacade()
    rng = np.random.default_rng(8)
    x = crandn(rng, (9, 12))
    for norm in (None, 'ortho', 'forward'):
        assert rel_max(tonp(fft.fft2(x, norm=norm)), np.fft.fft2(x, norm=norm)) < TOL64
        assert rel_max(tonp(fft.ifft2(x, norm=norm)), np.fft.ifft2(x, norm=norm)) < TOL64
    assert rel_max(tonp(fft.fft(x, 16, axis=0)), np.fft.fft(x, 16, axis=0)) < TOL64
    assert rel_max(tonp(fft.ifft(x, 8, axis=1)), np.fft.ifft(x, 8, axis=1)) < TOL64
    assert np.array_equal(tonp(fft.fftshift(x)), np.fft.fftshift(x)) and np.array_equal(tonp(fft.ifftshift(x)), np.fft.ifftshift(x))
    assert np.array_equal(tonp(fft.fftshift(x, axes=1)), np.fft.fftshift(x, axes=1))
    assert np.allclose(tonp(fft.fftfreq(9, 0.5)), np.fft.fftfreq(9, 0.5))
    # prysm's focus written against the shim: fftshift(fft2(ifftshift(x), norm='ortho'))
    got = tonp(fft.fftshift(fft.fft2(fft.ifftshift(x), norm='ortho')))
    assert rel_max(got, O.focus(x, 1)) < TOL64
    real = rng.standard_normal((16, 16))
    assert rel_max(tonp(fft.fft2(real)), np.fft.fft2(real)) < TOL64


def test_array_level_plug_on_the_device(pa):
    """The ``np`` facade with the ``fft`` facade on the MI355X: the reference's formulas written against the two shims.
    focus as fft.py:7-25 spells it; the matrix DFT and its adjoint as fttools.py:187-228 spell them with ``@`` / ``.T`` /
    ``.conj()`` -- complex 2-D products reach pm_cgemm with the views as operand flags; torch's lazy conj view handed to a
    fused-level function is resolved before the library reads the bytes."""
    from prysm_amd.mathops import FFTFacade
    from prysm_amd.npfacade import NumpyFacade, DeviceArray
    P = pa.propagation
    xp, fft = NumpyFacade(), FFTFacade()
    rng = np.random.default_rng(11)
    a = crandn(rng, (48, 40))
    padded = xp.pad(xp.asarray(a), ((24, 24), (20, 20)))
    got = fft.fftshift(fft.fft2(fft.ifftshift(padded), norm='ortho'))
    assert isinstance(got, DeviceArray) and got.is_cuda
    assert rel_max(got.get(), O.focus(a, 2)) < TOL64
    for cdt, rdt, tol in ((np.complex128, np.float64, TOL64), (np.complex64, np.float32, TOL32_MDFT)):
        x = xp.arange(-20, 20, dtype=rdt) * 0.25
        y = xp.arange(-24, 24, dtype=rdt) * 0.25
        fx = xp.arange(-8, 8, dtype=rdt) * 0.031
        fy = xp.arange(-6, 6, dtype=rdt) * 0.029
        Ex = xp.exp(-2j * xp.pi * xp.outer(fx, x)).astype(cdt)
        Ey = xp.exp(-2j * xp.pi * xp.outer(fy, y)).astype(cdt)
        ary = xp.asarray(a.astype(cdt))
        out = Ey @ ary @ Ex.T
        nEx, nEy = np.asarray(Ex).astype(np.complex128), np.asarray(Ey).astype(np.complex128)
        assert isinstance(out, DeviceArray) and tuple(out.shape) == (12, 16) and out.get().dtype == cdt
        assert rel_max(out.get(), nEy @ a @ nEx.T) < tol
        g = crandn(rng, (12, 16)).astype(cdt)
        back = Ey.conj().T @ xp.asarray(g) @ Ex.conj()
        assert rel_max(back.get(), nEy.conj().T @ g @ nEx.conj()) < tol
        assert rel_max((ary @ xp.conj(ary).T).get(), a @ a.conj().T) < tol
    # numpy operand on the left of a device array, intensity as wavefront.py:146-151 spells it
    inten = xp.real(got) ** 2 + xp.imag(got) ** 2
    assert rel_max(inten.get(), np.abs(O.focus(a, 2)) ** 2) < TOL64
    assert rel_max((np.conj(a) * xp.asarray(a)).get(), np.abs(a) ** 2) < 1e-14
    # lazy conj / neg views into the fused level
    t = torch.as_tensor(a, device='cuda')
    assert torch.conj(t).is_conj()
    assert rel_max(tonp(P.focus(torch.conj(t), 1)), O.focus(np.conj(a), 1)) < TOL64
    assert rel_max(tonp(P.focus(torch.conj(t).imag, 1)), O.focus(-a.imag, 1)) < TOL64


def test_fused_pupil_synthesis(pa):
    """Wavefront.from_amp_and_phase(...).focus / focus_intensity with the pupil synthesised inside the row pass
    (PM_FLAG_SYNTH_INPUT, complex64): equals the oracle's from_amp_and_phase -> focus, for bool / float / no amplitude,
    Q = 1 and 2, the folded (4096-row) and unfolded paths; materialising .data afterwards gives the same field as the
    separate synthesis kernel; the lazy wavefront still behaves like an array holder."""
    from prysm_amd.conf import config
    P = pa.propagation
    prec = config.precision
    try:
        config.precision = 32
        rng = np.random.default_rng(12)
        for n, Q in ((512, 1), (256, 2), (4096, 1)):
            x, y = O.make_xy_grid(n, diameter=10)
            r, _ = O.cart_to_polar(x, y)
            opd = (O.hopkins_w040(r / 5, 800.0) + 30 * rng.standard_normal((n, n))).astype(np.float32)
            for amp in (O.circle(5, r), rng.random((n, n)).astype(np.float32), None):
                wf = P.Wavefront.from_amp_and_phase(amp, opd, 0.55, 10.0 / n)
                if amp is not None:
                    assert wf._fusable(Q) is not None
                want = O.focus(O.from_amp_and_phase(np.ones((n, n)) if amp is None else amp, opd.astype(np.float64), 0.55), Q)
                got = wf.focus(100.0, Q)
                assert got.data.dtype == torch.complex64
                assert rel_max(tonp(got), want) < 2 * TOL32
                assert rel_max(tonp(wf.focus_intensity(100.0, Q)), O.intensity(want)) < 6 * TOL32
                assert wf._data is None                      # nothing materialised so far
                if n <= 512:
                    field = tonp(wf.data)                    # now it is
                    assert rel_max(field, O.from_amp_and_phase(np.ones((n, n)) if amp is None else amp, opd.astype(np.float64), 0.55)) < TOL32
                    assert rel_max(tonp(wf.focus(100.0, Q)), want) < 2 * TOL32      # unfused path on the materialised data
                    assert (wf * 2.0).data.shape == (n, n)
    finally:
        config.precision = prec


def test_numpy_operand_defers_to_wavefront_operators(pa):
    """numpy_array <op> Wavefront must be ONE device operation through the reflected operator (and np.asarray(wf) the
    field itself), never numpy's elementwise object loop."""
    P = pa.propagation
    rng = np.random.default_rng(3)
    x = crandn(rng, (32, 32))
    m = rng.standard_normal((32, 32))
    wf = P.Wavefront(x, 0.5, 1.0)
    assert rel_max(tonp((m * wf).data), m * x) < TOL64
    assert rel_max(tonp((wf * m).data), m * x) < TOL64
    assert rel_max(tonp((m + wf).data), m + x) < TOL64
    assert rel_max(np.asarray(wf), x) < TOL64
    from prysm_amd.mathops import array_to_true_numpy
    assert rel_max(array_to_true_numpy(wf), x) < TOL64


@pytest.mark.parametrize('shape,dtype', [((1000, 1000), np.complex64), ((300, 500), np.complex128), ((1000, 1024), np.complex64),
                                         ((1536, 9), np.complex128), ((97, 2000), np.complex64), ((4000, 130), np.complex128)])
def test_bluestein_lengths(pa, bluestein_route, shape, dtype):
    """Lengths that are not powers of two (96 .. 4096) run on the FFT engine through Bluestein's identity (csrc/bluestein.hip):
    against numpy, against the direct O(n^2) kernel (tuning blue_min = 0), for the focus family (pad / shift / crop / inverse),
    real input, the |.|^2 epilogue and a stack."""
    from prysm_amd import _lib, _ops
    lib = _lib.load()
    rng = np.random.default_rng(sum(shape))
    tol = TOL32 if dtype == np.complex64 else TOL64
    x = crandn(rng, shape, dtype)
    want = np.fft.fft2(x.astype(np.complex128))
    xd = torch.from_numpy(x).cuda()
    got = _ops.fft2(xd, direction=-1, scale=1.0).cpu().numpy()
    assert got.dtype == dtype and rel_max(got, want) < tol
    try:
        lib.pm_set_tuning(b'blue_min', 0)
        direct = _ops.fft2(xd, direction=-1, scale=1.0).cpu().numpy()
    finally:
        lib.pm_set_tuning(b'blue_min', 96)
    assert rel_max(got, direct) < 2 * tol
    try:   # both axes on the path: one fused convolution chain (default) or axis by axis
        lib.pm_set_tuning(b'blue_2d', 0)
        per_axis = _ops.fft2(xd, direction=-1, scale=1.0).cpu().numpy()
    finally:
        lib.pm_set_tuning(b'blue_2d', 1)
    assert rel_max(per_axis, want) < tol
    try:   # ... and with the chirp multiplies as separate kernels around the chain instead of inside its first load / last store
        lib.pm_set_tuning(b'blue_fuse', 0)
        unfused = _ops.fft2(xd, direction=-1, scale=1.0).cpu().numpy()
    finally:
        lib.pm_set_tuning(b'blue_fuse', 1)
    assert rel_max(unfused, want) < tol
    inv = _ops.fft2(torch.from_numpy(got).cuda(), direction=+1, scale=1.0 / (shape[0] * shape[1])).cpu().numpy()
    assert rel_max(inv, x) < 2 * tol
    # the focus family on these shapes: pad to Q = 2 (non power-of-two padded size), shifts, adjoint crop
    small = x[:shape[0] // 2, :shape[1] // 2]
    ref = O.focus(small.astype(np.complex128), 2)
    assert rel_max(tonp(pa.propagation.focus(small, 2)), ref) < tol
    assert rel_max(tonp(pa.propagation.focus_intensity(small, 2)), O.intensity(ref)) < 2 * tol
    g = crandn(rng, ref.shape, dtype)
    assert rel_max(tonp(pa.propagation.focus_adjoint(g, 2)), O.focus_adjoint(g.astype(np.complex128), 2)) < tol
    assert rel_max(tonp(pa.propagation.unfocus(x, 1)), O.unfocus(x.astype(np.complex128), 1)) < tol
    # real input read as it is, and a stack of fields (run field by field on this path)
    xr = np.ascontiguousarray(x.real)
    assert rel_max(_ops.fft2(torch.from_numpy(xr).cuda(), direction=-1, scale=1.0).cpu().numpy(), np.fft.fft2(xr.astype(np.float64))) < tol
    st = crandn(rng, (2,) + shape, dtype)
    gs = _ops.fft2(torch.from_numpy(st).cuda(), direction=-1, scale=1.0).cpu().numpy()
    assert max(rel_max(gs[b], np.fft.fft2(st[b].astype(np.complex128))) for b in range(2)) < tol


@pytest.mark.parametrize('dtype', [np.complex64, np.complex128])
def test_bluestein_fft1(pa, bluestein_route, dtype):
    """pm_fft1_ws: batched 1-D transforms of non power-of-two length along either axis, zero padded to n, cropped output."""
    from prysm_amd import _ops
    rng = np.random.default_rng(77)
    tol = TOL32 if dtype == np.complex64 else TOL64
    x = crandn(rng, (300, 1000), dtype)
    xd = torch.from_numpy(x).cuda()
    x128 = x.astype(np.complex128)
    assert rel_max(_ops.fft1(xd, axis=1).cpu().numpy(), np.fft.fft(x128, axis=1)) < tol
    assert rel_max(_ops.fft1(xd, axis=0).cpu().numpy(), np.fft.fft(x128, axis=0)) < tol
    assert rel_max(_ops.fft1(xd, n=1500, axis=1, direction=+1, scale=1 / 1500).cpu().numpy(), np.fft.ifft(x128, 1500, axis=1)) < tol
    assert rel_max(_ops.fft1(xd, n=777, axis=0, out_len=100, out_off=30).cpu().numpy(), np.fft.fft(x128, 777, axis=0)[30:130]) < tol
    assert rel_max(_ops.fft1(xd, n=600, axis=1).cpu().numpy(), np.fft.fft(x128, 600, axis=1)) < tol   # truncation


def test_bluestein_angular_spectrum_non_pow2(pa):
    """free space on a 1000^2 grid (two Bluestein transforms with the transfer function in between) against the oracle."""
    rng = np.random.default_rng(1000)
    x = crandn(rng, (1000, 1000))
    ref = O.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1)
    got = tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1))
    assert rel_max(got, ref) < TOL64


@pytest.mark.parametrize('shape', [(64, 128), (32, 64), (128, 16), (128, 128)])
@pytest.mark.parametrize('dtype', [np.complex64, np.complex128])
def test_big_power_of_two_path_on_small_arrays(pa, shape, dtype):
    """csrc/bigfft.hip (16384 / 32768-point axes: one radix-2 / radix-4 step around engine transforms) exercised on small arrays
    by lowering the native-length knob to 32: radix 1 / 2 / 4 per axis, windows, rotations, crops, real input, inverse, |.|^2."""
    from prysm_amd import _lib, _ops
    lib = _lib.load()
    rng = np.random.default_rng(sum(shape) + (dtype == np.complex64))
    tol = TOL32 if dtype == np.complex64 else TOL64
    M, N = shape
    try:
        lib.pm_set_tuning(b'big_native_log', 5)
        x = crandn(rng, shape, dtype)
        xd = torch.from_numpy(x).cuda()
        want = np.fft.fft2(x.astype(np.complex128))
        assert rel_max(_ops.fft2(xd, direction=-1, scale=1.0).cpu().numpy(), want) < tol
        assert rel_max(_ops.fft2(xd, direction=+1, scale=1.0 / (M * N)).cpu().numpy(), np.fft.ifft2(x.astype(np.complex128))) < tol
        # the focus family: pad window + both rotations, the adjoint's crop, real input, fused intensity
        small = x[:M // 2, :N // 2]
        ref = O.focus(small.astype(np.complex128), 2)
        assert rel_max(tonp(pa.propagation.focus(small, 2)), ref) < tol
        assert rel_max(tonp(pa.propagation.focus_intensity(small, 2)), O.intensity(ref)) < 2 * tol
        g = crandn(rng, shape, dtype)
        assert rel_max(tonp(pa.propagation.focus_adjoint(g, 2)), O.focus_adjoint(g.astype(np.complex128), 2)) < tol
        xr = np.ascontiguousarray(x.real)
        assert rel_max(_ops.fft2(torch.from_numpy(xr).cuda(), direction=-1, scale=1.0).cpu().numpy(), np.fft.fft2(xr.astype(np.float64))) < tol
        sh = (M // 2, N // 2)
        got = _ops.fft2(xd, direction=-1, scale=0.5, in_shift=(3, 5), out_shift=sh, out_shape=(M - 6, N - 2), out_off=(4, 1)).cpu().numpy()
        full = np.roll(0.5 * np.fft.fft2(np.roll(x.astype(np.complex128), (-3, -5), (0, 1))), sh, (0, 1))
        assert rel_max(got, full[4:4 + M - 6, 1:1 + N - 2]) < tol
        st = crandn(rng, (2,) + shape, dtype)
        gs = _ops.fft2(torch.from_numpy(st).cuda(), direction=-1, scale=1.0).cpu().numpy()
        assert max(rel_max(gs[b], np.fft.fft2(st[b].astype(np.complex128))) for b in range(2)) < tol
    finally:
        lib.pm_set_tuning(b'big_native_log', 13)


@pytest.mark.parametrize('shape', [(16384, 64), (64, 16384), (32768, 8)])
def test_big_power_of_two_lengths(pa, shape):
    """16384- and 32768-point axes at full length against numpy."""
    from prysm_amd import _ops
    rng = np.random.default_rng(shape[0])
    x = crandn(rng, shape)
    got = _ops.fft2(torch.from_numpy(x).cuda(), direction=-1, scale=1.0).cpu().numpy()
    assert rel_max(got, np.fft.fft2(x)) < TOL64
    assert rel_max(tonp(pa.propagation.unfocus(x, 1)), O.unfocus(x, 1)) < TOL64


def test_big_16384_squared_separable_field(pa):
    """focus of a 16384^2 complex64 field (2 GiB; the reference would take minutes): a separable input a (x) b has the separable
    transform F(a) (x) F(b), so the truth costs two 1-D numpy FFTs."""
    n = 16384
    rng = np.random.default_rng(n)
    a, b = crandn(rng, n), crandn(rng, n)
    xd = torch.outer(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).to(torch.complex64)
    got = pa.propagation.focus(xd, 1)
    fa = np.fft.fftshift(np.fft.fft(np.fft.ifftshift(a), norm='ortho'))
    fb = np.fft.fftshift(np.fft.fft(np.fft.ifftshift(b), norm='ortho'))
    want = torch.outer(torch.from_numpy(fa).cuda(), torch.from_numpy(fb).cuda())
    err = float((got.to(torch.complex128) - want).abs().max() / want.abs().max())
    assert got.dtype == torch.complex64 and err < 2 * TOL32
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    pa.propagation.focus(xd, 1)
    ev1.record()
    torch.cuda.synchronize()
    print('focus 16384^2 complex64: %.2f ms' % ev0.elapsed_time(ev1))


@pytest.mark.parametrize('shape,dtype', [((40, 24), np.complex128), ((50, 50), np.complex64), ((24, 100), np.complex128),
                                         ((40, 16), np.complex128), ((64, 24), np.complex64), ((6, 40), np.complex128)])   # mixed shapes: native / split power of two / short axis beside a long one
def test_long_bluestein_on_small_arrays(pa, bluestein_route, shape, dtype):
    """Lengths in (4096, 16384] that are not powers of two convolve at 16384 / 32768 points: chirp multiply, big transform with the
    chirp spectrum in its epilogue, big inverse with the crop, chirp multiply.  Run here on small arrays (native length 32, path
    from 20 points) against numpy: forward, inverse, windows, real input."""
    from prysm_amd import _lib, _ops
    lib = _lib.load()
    rng = np.random.default_rng(sum(shape))
    tol = TOL32 if dtype == np.complex64 else TOL64
    M, N = shape
    try:
        lib.pm_set_tuning(b'big_native_log', 5)
        lib.pm_set_tuning(b'blue_min', 20)
        x = crandn(rng, shape, dtype)
        xd = torch.from_numpy(x).cuda()
        x128 = x.astype(np.complex128)
        assert rel_max(_ops.fft2(xd, direction=-1, scale=1.0).cpu().numpy(), np.fft.fft2(x128)) < tol
        assert rel_max(_ops.fft2(xd, direction=+1, scale=1.0 / (M * N)).cpu().numpy(), np.fft.ifft2(x128)) < tol
        small = x[:M // 2, :N // 2]
        ref = O.focus(small.astype(np.complex128), 2)
        assert rel_max(tonp(pa.propagation.focus(small, 2)), ref) < tol
        assert rel_max(tonp(pa.propagation.focus_intensity(small, 2)), O.intensity(ref)) < 2 * tol
        g = crandn(rng, shape, dtype)
        assert rel_max(tonp(pa.propagation.focus_adjoint(g, 2)), O.focus_adjoint(g.astype(np.complex128), 2)) < tol
        xr = np.ascontiguousarray(x.real)
        assert rel_max(_ops.fft2(torch.from_numpy(xr).cuda(), direction=-1, scale=1.0).cpu().numpy(), np.fft.fft2(xr.astype(np.float64))) < tol
    finally:
        lib.pm_set_tuning(b'big_native_log', 13)
        lib.pm_set_tuning(b'blue_min', 96)


def test_long_bluestein_5000(pa, bluestein_route):
    """unfocus of a 5000 x 4500 complex64 field (convolution at 16384 points) against numpy; timing of 8000^2 printed."""
    rng = np.random.default_rng(5000)
    x = crandn(rng, (5000, 4500), np.complex64)
    got = tonp(pa.propagation.unfocus(x, 1))
    assert rel_max(got, O.unfocus(x.astype(np.complex128), 1)) < TOL32
    xd = torch.from_numpy(crandn(rng, (8000, 8000), np.complex64)).cuda()
    pa.propagation.focus(xd, 1)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    pa.propagation.focus(xd, 1)
    ev1.record()
    torch.cuda.synchronize()
    print('focus 8000^2 complex64 (Bluestein at 16384^2): %.2f ms' % ev0.elapsed_time(ev1))


def test_randomised_differential_fuzz(pa):
    """tools/fuzz_fft2.py: random sizes / windows / rotations / crops / input kinds / stacks / epilogues / multipliers /
    precisions / fold settings of pm_fft2 and the fused chain against numpy (a fixed seed keeps it reproducible)."""
    import importlib.util
    import os
    import sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_fft2.py')
    spec = importlib.util.spec_from_file_location('fuzz_fft2', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    try:
        sys.argv = ['fuzz_fft2.py', '60', '2026']
        assert mod.main() == 0
    finally:
        sys.argv = argv
