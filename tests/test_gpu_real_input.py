"""GPU parity tests of the real-input (Hermitian) 2-D transform, both forms (round 6):

* csrc/fft_r2c.h (round 2): half spectrum along x -- R2C row pass, column pass that stores every bin and its mirror image;
* csrc/fft_hermt.h (round 6, the planner's default where it measured faster): half spectrum along y -- real-input column
  transforms (unfolded, and folded into planes of half-height tiles), then row transforms that store each row and its image.

Every case runs on BOTH forms (knob herm_t) against numpy fp64 -- fft2 of the rotated real array (prysm/otf.py:28-33 transform_psf)
with the centre normalisation and the |.|, |.|^2, angle epilogues of the MTF / PTF / OTF routines (prysm/otf.py:62-135) -- so the
form the planner does not pick for a shape stays covered.  Tolerances (max error / max magnitude): 1e-11 fp64, 3e-5 fp32.
"""
import numpy as np
import pytest
import torch

from conftest import rel_max
from oracle import prysm_oracle as O

pytestmark = pytest.mark.gpu


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


def _ref(x, in_shift, out_shift, norm_dc):
    a = np.roll(x.astype(np.float64), (-in_shift[0], -in_shift[1]), axis=(0, 1))     # logical element i is read from position i + shift
    F = np.fft.fft2(a)
    if norm_dc:
        F = F / F[0, 0]
    return np.roll(F, out_shift, axis=(0, 1))


FORMS = {'r2c': dict(herm_t=0), 'transposed': dict(herm_t=1, herm_t_fold=0), 'transposed_fold': dict(herm_t=1, herm_t_fold=1)}


@pytest.mark.parametrize('form', list(FORMS))
@pytest.mark.parametrize('shape', [(32, 32), (64, 256), (256, 64), (512, 512), (2048, 512), (512, 4096), (4096, 2048)])
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_real_input_forms_vs_numpy(pa, shape, dtype, form):
    from prysm_amd import _lib as L, _ops
    M, N = shape
    if form == 'transposed_fold' and M < 2048:
        pytest.skip('the fold is for columns of 2048 samples and more')
    if dtype == np.float64 and N > 2048 and form != 'r2c' and M * N > 1 << 22:
        pytest.skip('not a shape the transposed form takes at this precision')
    tol = 3e-5 if dtype == np.float32 else 1e-11
    rng = np.random.default_rng(M + 3 * N)
    x = (rng.random((M, N)) + 0.05).astype(dtype)
    xd = torch.from_numpy(x).cuda()
    epis = {'none': L.PM_EPI_NONE, 'abs2': L.PM_EPI_ABS2, 'abs': L.PM_EPI_ABS, 'arg': L.PM_EPI_ARG}
    cases = [((0, 0), (0, 0)), ((M // 2, N // 2), (M // 2, N // 2)), ((M // 2, 0), (0, N // 2)), ((0, N // 2), (M // 2, 0))]
    if M * N > 1 << 21:
        cases = cases[1:3]
    with L.tuning_local(r2c=2, **FORMS[form]):
        for in_shift, out_shift in cases:
            for norm_dc in (False, True):
                F = _ref(x, in_shift, out_shift, norm_dc)
                for name, code in epis.items():
                    got = tonp(_ops.fft2(xd, direction=-1, scale=1.0, in_shift=in_shift, out_shift=out_shift, epilogue=code,
                                         flags=L.PM_FLAG_REAL_INPUT | (L.PM_FLAG_NORM_DC if norm_dc else 0)))
                    if name == 'none':
                        assert got.dtype == (np.complex64 if dtype == np.float32 else np.complex128)
                        assert rel_max(got, F) < tol, (form, in_shift, out_shift, norm_dc, name)
                    elif name == 'arg':
                        ok = np.abs(F) > 1e-3 * np.abs(F).max()          # the phase of a tiny bin amplifies rounding
                        err = np.max(np.abs(np.exp(1j * got[ok]) - np.exp(1j * np.angle(F[ok]))))
                        assert err < (2e-3 if dtype == np.float32 else 1e-8), (form, in_shift, out_shift, norm_dc, name)
                    else:
                        want = np.abs(F) if name == 'abs' else np.abs(F) ** 2
                        assert got.dtype == dtype and rel_max(got, want) < tol, (form, in_shift, out_shift, norm_dc, name)


@pytest.mark.parametrize('form', ['r2c', 'transposed'])
def test_hermitian_symmetry_is_exact(pa, form):
    """F[(M - u) mod M][(N - k) mod N] = conj F[u][k], bit for bit: both forms write a bin and its image from the same registers"""
    from prysm_amd import _lib as L, _ops
    M, N = 1024, 2048
    x = torch.rand((M, N), device='cuda', generator=torch.Generator(device='cuda').manual_seed(2), dtype=torch.float32)
    with L.tuning_local(r2c=2, **FORMS[form]):
        F = _ops.fft2(x, direction=-1, scale=1.0, flags=L.PM_FLAG_REAL_INPUT)
    Fm = torch.roll(torch.flip(F, (0, 1)), (1, 1), (0, 1)).conj()
    assert torch.equal(F, Fm.resolve_conj())
    assert float(F[0, 0].imag) == 0.0 and float(F[M // 2, 0].imag) == 0.0 and float(F[0, N // 2].imag) == 0.0


@pytest.mark.parametrize('n,rdt,tol', [(512, np.float32, 2e-5), (2048, np.float32, 2e-5), (1024, np.float64, 1e-11), (4096, np.float32, 2e-5)])
def test_mtf_ptf_otf_default_route_vs_oracle(pa, n, rdt, tol):
    """otf.mtf_from_psf / ptf_from_psf / otf_from_psf on the route the planner picks (the transposed form at these sizes), against the
    oracle's restatement of prysm/otf.py:62-135 on the same PSF"""
    from prysm_amd import otf, _lib as L
    import ctypes
    rng = np.random.default_rng(n)
    psf = (rng.random((n, n)) ** 3 + 0.01).astype(rdt)
    lib = L.load()
    d = L.pm_fft2_desc()
    d.dtype = L.PM_C64 if rdt == np.float32 else L.PM_C128
    d.direction = -1
    d.in_y = d.in_x = d.out_y = d.out_x = L.pm_axis(n, n, 0, n // 2)
    d.in_ld = d.out_ld = n
    d.flags = L.PM_FLAG_REAL_INPUT | L.PM_FLAG_NORM_DC
    d.epilogue = L.PM_EPI_ABS
    buf = ctypes.create_string_buffer(256)
    L.check(lib.pm_plan_explain(ctypes.byref(d), 0, buf, 256))
    assert 'route=hermitian-transposed' in buf.value.decode()
    mtf = tonp(otf.mtf_from_psf(psf, 1.0).data)
    want = O.mtf_from_psf(psf.astype(np.float64))
    assert mtf.dtype == rdt and rel_max(mtf, want) < tol
    F = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(psf.astype(np.float64))))
    F = F / F[n // 2, n // 2]
    o = tonp(otf.otf_from_psf(psf, 1.0).data)
    assert rel_max(o, F) < tol
    p = tonp(otf.ptf_from_psf(psf, 1.0).data)
    ok = np.abs(F) > 1e-3
    assert np.max(np.abs(np.exp(1j * p[ok]) - np.exp(1j * np.angle(F[ok])))) < (2e-3 if rdt == np.float32 else 1e-8)
