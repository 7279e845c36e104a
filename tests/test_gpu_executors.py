"""GPU parity tests, by component: the fixed-sampling executors -- complex GEMM on MFMA (all operand forms, split K, the |.|^2 epilogue), MDFT / CZT / FFTDFT
against each other and the oracle, executors built from grid parameters (csrc/cgemm.hip, fft_conv1.h; prysm_amd/fttools.py).

All through the C ABI (ctypes -> libprysm_amd.so), against the fp64 oracle / numpy first and a second HIP route only afterwards.
Tolerances (max error / max magnitude against fp64): 1e-10 complex128, 5e-6 complex64 transforms, 3e-5 the MFMA matrix DFT.
(Regrouped in round 6 from the per-round files of rounds 2 - 5; the tests themselves are unchanged.)
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


def test_mdft_adjoint_512_to_2048_c64_vs_oracle(pa):
    """MDFT.adjoint of config 4: (Ey^H @ g @ conj(Ex)) norm, 512^2 -> 2048^2, complex64 bases (LDS-DMA GEMM kernel: transposed /
    conjugated operand forms), against the fp64 oracle"""
    P = pa.propagation
    rng = np.random.default_rng(5122048)
    pdx, efl, wvl = 10 / 2048, 100.0, O.HeNe
    fdx = wvl * 10 / 8
    g = crandn(rng, (512, 512), np.complex64)
    ref_ex = O.prepare_executor(pdx, (2048, 2048), fdx, (512, 512), wvl, efl)
    ref = ref_ex.adjoint(g.astype(np.complex128))
    prec = pa.config.precision
    pa.config.precision = 32
    try:
        ex = P.prepare_executor(pdx, (2048, 2048), fdx, (512, 512), wvl, efl)
        got = tonp(P.focus_dft_adjoint(g, ex))
    finally:
        pa.config.precision = prec
    assert got.shape == (2048, 2048) and got.dtype == np.complex64
    assert rel_max(got, ref) < TOL32_MDFT


@pytest.mark.parametrize('M,N,K', [(512, 2048, 64), (1024, 1024, 96), (512, 512, 64), (512, 1024, 192), (768, 1024, 128),
                                   (512, 512, 2048), (512, 2048, 2048)])
def test_cgemm_in_workgroup_k_split_all_ops(pa, M, N, K):
    """split-K inside the workgroup (the eight-wave 64 x 64 form config 4's first product now runs on; the 64 x 32 / 32 x 32 forms where
    the build contains them) for every transposed / conjugated operand storage, against numpy in fp64; gemm_wk = 0 (round 2's split-K
    slabs) must agree to rounding"""
    from prysm_amd import _lib, _ops
    lib = _lib.load()
    rng = np.random.default_rng(M + N + K)
    ops = [(0, 0), (3, 1), (0, 2), (2, 3)] if K >= 1024 else [(a, b) for a in range(4) for b in range(4)]
    for opA, opB in ops:
        A = crandn(rng, (K, M) if opA & 2 else (M, K), np.complex64)
        B = crandn(rng, (N, K) if opB & 2 else (K, N), np.complex64)
        ref = _op_np(A.astype(np.complex128), opA) @ _op_np(B.astype(np.complex128), opB)
        At, Bt = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
        got = tonp(_ops.cgemm(At, Bt, opA, opB, alpha=0.5))
        assert rel_max(got, 0.5 * ref) < TOL32_MDFT, (opA, opB)
        for form in (0, 2):     # round 2's slabs (0); 2 = a form that left the library in round 5 and must be refused, not run
            if lib.pm_set_tuning(b'gemm_wk', form) != 0:
                continue
            try:
                old = tonp(_ops.cgemm(At, Bt, opA, opB, alpha=0.5))
            finally:
                lib.pm_set_tuning(b'gemm_wk', 1)
            assert rel_max(got, old) < 1e-5, (opA, opB, form)
        # bitwise reproducible: the K-groups are summed in a fixed order
        assert np.array_equal(got, tonp(_ops.cgemm(At, Bt, opA, opB, alpha=0.5)))


@pytest.mark.parametrize('M,N,K', [(512, 512, 256), (512, 2048, 128), (2048, 2048, 64), (256, 256, 512), (128, 64, 1024)])
def test_cgemm_abs2_epilogue(pa, M, N, K):
    """pm_cgemm_abs2: weight |alpha A @ B^T|^2 stored / accumulated as a real image (in-kernel epilogue for the unsplit plans, the
    slab reduce's epilogue when K is split across workgroups) against numpy"""
    from prysm_amd import _ops
    rng = np.random.default_rng(M * 3 + N + K)
    A = crandn(rng, (M, K), np.complex64)
    B = crandn(rng, (N, K), np.complex64)
    ref = np.abs(0.25 * (A.astype(np.complex128) @ B.astype(np.complex128).T)) ** 2
    At, Bt = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    I = _ops.cgemm_abs2(At, Bt, 0, 2, alpha=0.25)
    assert I is not None and I.dtype == torch.float32
    assert rel_max(tonp(I), ref) < 2 * TOL32_MDFT
    base = torch.from_numpy(rng.random((M, N)).astype(np.float32)).cuda()
    want = tonp(base).astype(np.float64) + 1.5 * ref
    out = _ops.cgemm_abs2(At, Bt, 0, 2, alpha=0.25, out=base, weight=1.5)
    assert out is base and rel_max(tonp(base), want) < 2 * TOL32_MDFT
    assert _ops.cgemm_abs2(At[:, :K - 3].contiguous(), Bt[:, :K - 3].contiguous(), 0, 2) is None     # ragged K: not this kernel's


def test_mdft_intensity_matches_composed(pa):
    """MDFT.intensity (focus_dft + intensity + weighted accumulate, modulus in the second product's epilogue) on config 4's grid
    against the fp64 oracle, and its fallback for complex128 bases"""
    P = pa.propagation
    rng = np.random.default_rng(2048512)
    x = crandn(rng, (2048, 2048), np.complex64)
    pdx, efl, wvl = 10 / 2048, 100.0, O.HeNe
    fdx = wvl * 10 / 8
    ref = O.intensity(O.prepare_executor(pdx, (2048, 2048), fdx, (512, 512), wvl, efl)(x.astype(np.complex128)))
    prec = pa.config.precision
    pa.config.precision = 32
    try:
        ex = P.prepare_executor(pdx, (2048, 2048), fdx, (512, 512), wvl, efl)
        I = ex.intensity(x)
        assert I.dtype == torch.float32 and rel_max(tonp(I), ref) < 2 * TOL32_MDFT
        acc = torch.zeros((512, 512), device='cuda')
        ex.intensity(x, out=acc, weight=0.5)
        ex.intensity(x, out=acc, weight=0.25)
        assert rel_max(tonp(acc), 0.75 * ref) < 2 * TOL32_MDFT
    finally:
        pa.config.precision = prec
    ex64 = P.prepare_executor(pdx, (2048, 2048), fdx, (512, 512), wvl, efl)
    assert rel_max(tonp(ex64.intensity(x.astype(np.complex128), weight=2.0)), 2.0 * ref) < TOL64


@pytest.mark.parametrize('sign', (-1, 1))
@pytest.mark.parametrize('input_shape,output_shape,fft_shape,dys', [((7, 9), (5, 6), (16, 16), -1), ((5, 6), (7, 9), (16, 32), 1),
                                                                    ((40, 33), (21, 64), (64, 64), 1), ((200, 120), (64, 100), (256, 128), -1)])
def test_fftdft_fused_axes_match_mdft(pa, sign, input_shape, output_shape, fft_shape, dys):
    """FFTDFT on engine lengths K: one pm_fft1_ramp kernel per axis (ramp, pad, transform, crop, ramp) -- forward against the matrix
    DFT on the same grids (tests/test_fttools.py:160-184), adjoint by the dot-product identity (:187-211); dy < 0 runs the
    inverse-transform axis"""
    rng = np.random.default_rng(sum(input_shape) + sign)
    (ny, nx), (my, mx), (ky, kx) = input_shape, output_shape, fft_shape
    r = lambda n: tonp(pa.fttools.fftrange(n)).astype(float)   # noqa: E731
    dx, dy = 0.25, 0.125 * dys       # binary spacings: the reference's 32-eps spacing test (fttools.py:491,503) rejects grids whose
    x, y = r(nx) * dx + 0.375, r(ny) * dy - 0.5      # rounded coordinates miss 1 / K by more than that at K = 256
    fx, fy = (r(mx) + 0.25) / (kx * dx), (r(my) - 0.5) / (ky * abs(dy))
    inp = crandn(rng, input_shape)
    mdft = pa.fttools.MDFT(x, y, fx, fy, sign=sign, norm=0.3)
    op = pa.fttools.FFTDFT(x, y, fx, fy, sign=sign, norm=0.3)
    assert op._fused()
    want = tonp(mdft(inp))
    for x_first in (True, False):
        op._x_first = x_first
        np.testing.assert_allclose(tonp(op(inp)), want, rtol=1e-11, atol=1e-11 * np.abs(want).max())
        grad = crandn(rng, output_shape)
        lhs = np.vdot(tonp(op(inp)), grad)
        rhs = np.vdot(inp, tonp(op.adjoint(grad)))
        np.testing.assert_allclose(lhs, rhs, rtol=1e-11)
        np.testing.assert_allclose(tonp(op.adjoint(grad)), tonp(mdft.adjoint(grad)), rtol=1e-11, atol=1e-11 * np.abs(want).max())


def test_fftdft_2048_to_512_K8192_vs_oracle(pa):
    """FFTDFT on config 4's shapes (2048^2 -> 512^2) with K = 8192 per axis -- the longest engine transform, rows and columns of 8192
    points in one kernel each -- in complex128 against the oracle's matrix DFT.  The grids have binary spacings (dx = 1/256,
    dfx = 1/32): prepare_executor's decimal grids at this size trip the reference's own 32-eps spacing test (fttools.py:491,503),
    in the reference as here."""
    rng = np.random.default_rng(8192)
    a = crandn(rng, (2048, 2048))
    r = lambda n: tonp(pa.fttools.fftrange(n)).astype(float)   # noqa: E731
    x = y = r(2048) / 256.0
    fx = fy = r(512) / 32.0
    op = pa.fttools.FFTDFT(x, y, fx, fy, norm=1.0 / 8192)
    assert op._fused() and op._Kx == 8192 and op._Ky == 8192
    ref = O.MDFT(x, y, fx, fy, norm=1.0 / 8192)(a)
    assert rel_max(tonp(op(a)), ref) < 1e-9
    g = crandn(rng, (512, 512))
    assert rel_max(tonp(op.adjoint(g)), O.MDFT(x, y, fx, fy, norm=1.0 / 8192).adjoint(g)) < 1e-9


def test_wavefront_focus_dft_intensity(pa):
    """Wavefront.focus_dft_intensity == focus_dft(...).intensity for the three executor kinds (the matrix DFT with the modulus in its
    second product's epilogue), with and without a weighted accumulate"""
    P = pa.propagation
    rng = np.random.default_rng(5)
    amp = (rng.random((256, 256)) > 0.3).astype(np.float32)
    opd = (100 * rng.standard_normal((256, 256))).astype(np.float32)
    prec = pa.config.precision
    pa.config.precision = 32
    try:
        wf = P.Wavefront.from_amp_and_phase(amp, opd, 0.55, 0.04)
        for kind in ('mdft', 'czt'):
            ex = wf.prepare_executor(100.0, 1.0, 128, kind=kind)
            want = tonp(wf.focus_dft(ex).intensity.data).astype(np.float64)
            got = wf.focus_dft_intensity(ex)
            assert got.dx == ex.focal_dx and rel_max(tonp(got.data), want) < 2e-5, kind
            acc = torch.full((128, 128), 1.0, device='cuda')
            wf.focus_dft_intensity(ex, out=acc, weight=0.5)
            assert rel_max(tonp(acc), 1.0 + 0.5 * want) < 2e-5, kind
    finally:
        pa.config.precision = prec
    with pytest.raises(ValueError):
        P.Wavefront(np.ones((8, 8), complex), 0.5, 1.0, space='psf').focus_dft_intensity(None)


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
