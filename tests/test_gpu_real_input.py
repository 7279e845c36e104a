"""GPU parity tests of the real-input (Hermitian) 2-D transform, both forms (round 6):

* csrc/fft_r2c.h (round 2): half spectrum along x -- R2C row pass, column pass that stores every bin and its mirror image;
* csrc/fft_hermt.h (round 6, the planner's default where it measured faster): half spectrum along y -- real-input column
  transforms (unfolded, and folded into planes of half-height tiles), then row transforms that store each row and its image.

Every case runs on BOTH forms (knob herm_t) against numpy fp64 -- fft2 of the rotated real array (prysm/otf.py:28-33 transform_psf)
with the centre normalisation and the |.|, |.|^2, angle epilogues of the MTF / PTF / OTF routines (prysm/otf.py:62-135) -- so the
form the planner does not pick for a shape stays covered.  Tolerances (max error / max magnitude): 1e-11 fp64, 3e-5 fp32.
"""
import ctypes
import json
import math
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

from conftest import ROOT, rel_max
from oracle import prysm_oracle as O
from gpu_common import (  # noqa: F401
    TOL64, TOL32, TOL32_MDFT, tonp, _real_vdot, crandn_, _np_transform_psf, _two_rank_backend, _env, _spectral_case,
    crandn, _op_np, _poly_numpy, _seven_planes, CE_LENGTHS, _ce_ref)

pytestmark = pytest.mark.gpu


def _ref(x, in_shift, out_shift, norm_dc):
    a = np.roll(x.astype(np.float64), (-in_shift[0], -in_shift[1]), axis=(0, 1))     # logical element i is read from position i + shift
    F = np.fft.fft2(a)
    if norm_dc:
        F = F / F[0, 0]
    return np.roll(F, out_shift, axis=(0, 1))


FORMS = {'r2c': dict(herm_t=0), 'transposed': dict(herm_t=1, herm_t_fold=0), 'transposed_fold': dict(herm_t=1, herm_t_fold=1)}


@pytest.mark.parametrize('form', list(FORMS))
@pytest.mark.parametrize('shape', [(32, 32), (64, 256), (256, 64), (512, 512), (2048, 512), (512, 4096), (4096, 2048), (8192, 256)])
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


# ----------------------------------------------------------------------------- rounds 2 - 5 (both forms reach these through the planner)

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


@pytest.mark.parametrize('shape,dtype,tol', [((512, 512), np.float32, 5e-6), ((256, 1024), np.float64, 1e-12), ((100, 60), np.float64, 1e-12)])
def test_mtf_ptf_otf_three_outputs(pa, shape, dtype, tol):
    """mtf_ptf_otf_from_psf (otf.py:167-203): centre-normalised OTF from the Hermitian transform pair + |.| and angle in one sweep
    (pm_abs_arg); the 100 x 60 PSF takes the complex path and the same sweep"""
    from prysm_amd import otf
    rng = np.random.default_rng(shape[0] + shape[1])
    psf = (rng.random(shape) + 0.05).astype(dtype)
    F = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(psf.astype(np.float64))))
    Fn = F / F[shape[0] // 2, shape[1] // 2]
    m, p, o = otf.mtf_ptf_otf_from_psf(psf, 1.0)
    assert rel_max(tonp(o.data), Fn) < tol and rel_max(tonp(m.data), np.abs(Fn)) < tol
    d = np.angle(np.exp(1j * (tonp(p.data).astype(np.float64) - np.angle(Fn))))     # phases compared on the circle
    sel = np.abs(Fn) > 1e-3
    assert np.max(np.abs(d[sel])) < (2e-4 if dtype == np.float32 else 1e-9)
    m2, p2, o2, raw = otf.mtf_ptf_otf_from_psf(psf, 1.0, return_more=True)
    assert rel_max(tonp(raw), F) < tol and rel_max(tonp(m2.data), np.abs(Fn)) < tol


def test_real_input_at_an_odd_float_offset(pa):
    """a real view whose base address is one float off a complex boundary must not be read with misaligned pair loads:
    the library sends it down the complex path (its workspace query covers both), results unchanged"""
    from prysm_amd import _ops, otf
    rng = np.random.default_rng(31)
    big = torch.from_numpy(rng.random((256, 258)).astype(np.float32) + 0.1).cuda()
    view = big[:, 1:257]
    assert view.data_ptr() % 8 == 4 and view.stride(0) % 2 == 0
    ref = np.fft.fft2(tonp(view).astype(np.float64))
    got = tonp(_ops.fft2(view, direction=-1, scale=1.0, epilogue=_ops.L.PM_EPI_ABS2))
    assert rel_max(got, np.abs(ref) ** 2) < 2e-5
    F = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(tonp(view).astype(np.float64))))
    assert rel_max(tonp(otf.mtf_from_psf(view, 1.0).data), np.abs(F / F[128, 128])) < 5e-6
    H = torch.from_numpy(crandn(rng, (256, 256), np.complex64)).cuda()
    conv = tonp(_ops.fft2_mul_ifft2(view, scale=1.0 / 256 ** 2, mul=H, real_out=True))
    want = np.real(np.fft.ifft2(np.fft.fft2(tonp(view).astype(np.float64)) * tonp(H).astype(np.complex128)))
    assert rel_max(conv, want) < 5e-6


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


@pytest.mark.parametrize('n', [4096, 2048, 512])
def test_real_input_rows_on_the_lean_store(pa, n):
    """the Hermitian path's row pass (fft_r2c.h) writes the tiled intermediate through the same addressing, folded from 1024 rows"""
    from prysm_amd import otf
    rng = np.random.default_rng(n)
    psf = (rng.random((n, n)) + 0.01).astype(np.float32)
    F = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(psf.astype(np.float64))))
    want = np.abs(F / F[n // 2, n // 2])
    for log_k in (-1, 0, 6):
        from prysm_amd import _lib
        with _lib.tuning_local(log_k=log_k):
            assert np.max(np.abs(tonp(otf.mtf_from_psf(psf, 1.0).data) - want)) < 5e-6
