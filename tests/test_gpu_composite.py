"""GPU parity tests, by component: composite (non-power-of-two) lengths -- the radix-R step, the LDS-resident mixed-radix kernel, the composite register engine
with its compile-time plans, real inputs of any even width, stacks (csrc/bigfft.hip, fft_mixed.h, fft_ce.h).

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


@pytest.mark.parametrize('shape,dtype', [((1000, 1000), np.complex64), ((300, 500), np.complex128), ((1000, 1024), np.complex64),
                                         ((1536, 45), np.complex128), ((77, 2000), np.complex64), ((4000, 130), np.complex128),
                                         ((1001, 143), np.complex128), ((2592, 729), np.complex64), ((3000, 36), np.complex128),
                                         ((250, 8190), np.complex64), ((6000, 40), np.complex128), ((3125, 343), np.complex128)])
def test_composite_lengths_on_the_mixed_radix_kernel(pa, shape, dtype):
    """lengths whose primes are all <= 13 (scipy.fft takes them natively: prysm/propagation/fft.py:24) run on one LDS-resident
    mixed-radix kernel per axis: against numpy, against round 2's route (Bluestein / direct, knob mix = 0), for the focus family
    (pad / shift / crop / inverse), real input, the |.|^2 epilogue and a stack"""
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
        lib.pm_set_tuning(b'mix', 0)
        old = _ops.fft2(xd, direction=-1, scale=1.0).cpu().numpy()
    finally:
        lib.pm_set_tuning(b'mix', 1)
    assert rel_max(got, old) < 2 * tol
    inv = _ops.fft2(torch.from_numpy(got).cuda(), direction=+1, scale=1.0 / (shape[0] * shape[1])).cpu().numpy()
    assert rel_max(inv, x) < 2 * tol
    if shape[0] * shape[1] <= 1100 * 1100:
        small = x[:shape[0] // 2, :shape[1] // 2]
        ref = O.focus(small.astype(np.complex128), 2)
        assert rel_max(tonp(pa.propagation.focus(small, 2)), ref) < tol
        assert rel_max(tonp(pa.propagation.focus_intensity(small, 2)), O.intensity(ref)) < 2 * tol
        g = crandn(rng, ref.shape, dtype)
        assert rel_max(tonp(pa.propagation.focus_adjoint(g, 2)), O.focus_adjoint(g.astype(np.complex128), 2)) < tol
    assert rel_max(tonp(pa.propagation.unfocus(x, 1)), O.unfocus(x.astype(np.complex128), 1)) < tol
    xr = np.ascontiguousarray(x.real)
    assert rel_max(_ops.fft2(torch.from_numpy(xr).cuda(), direction=-1, scale=1.0).cpu().numpy(), np.fft.fft2(xr.astype(np.float64))) < tol
    if shape[0] * shape[1] <= 1100 * 1100:
        st = crandn(rng, (2,) + shape, dtype)
        gs = _ops.fft2(torch.from_numpy(st).cuda(), direction=-1, scale=1.0).cpu().numpy()
        assert max(rel_max(gs[b], np.fft.fft2(st[b].astype(np.complex128))) for b in range(2)) < tol


@pytest.mark.parametrize('dtype', [np.complex64, np.complex128])
def test_composite_lengths_fft1(pa, dtype):
    """pm_fft1 on the mixed-radix kernel: both axes, zero padded to n, truncated, cropped and scaled outputs, odd batch extents"""
    from prysm_amd import _ops
    rng = np.random.default_rng(78)
    tol = TOL32 if dtype == np.complex64 else TOL64
    x = crandn(rng, (301, 1000), dtype)
    xd = torch.from_numpy(x).cuda()
    x128 = x.astype(np.complex128)
    assert rel_max(_ops.fft1(xd, axis=1).cpu().numpy(), np.fft.fft(x128, axis=1)) < tol
    assert rel_max(_ops.fft1(xd, n=315, axis=0).cpu().numpy(), np.fft.fft(x128, 315, axis=0)) < tol
    assert rel_max(_ops.fft1(xd, n=1500, axis=1, direction=+1, scale=1 / 1500).cpu().numpy(), np.fft.ifft(x128, 1500, axis=1)) < tol
    assert rel_max(_ops.fft1(xd, n=770, axis=0, out_len=100, out_off=30).cpu().numpy(), np.fft.fft(x128, 770, axis=0)[30:130]) < tol
    assert rel_max(_ops.fft1(xd, n=600, axis=1).cpu().numpy(), np.fft.fft(x128, 600, axis=1)) < tol   # truncation
    assert rel_max(_ops.fft1(xd, n=7000, axis=1, scale=0.5).cpu().numpy(), 0.5 * np.fft.fft(x128, 7000, axis=1)) < tol
    for n in (18, 20, 24, 30, 35, 48, 54, 60, 63, 72, 80, 84, 90, 99, 108, 117, 165, 169, 182, 195, 210, 1331, 2197, 2401, 4095):
        y = x128[:7, :min(n, 1000)]
        assert rel_max(_ops.fft1(torch.from_numpy(y.astype(dtype)).cuda(), n=n, axis=1).cpu().numpy(), np.fft.fft(y.astype(dtype).astype(np.complex128), n, axis=1)) < tol, n


def test_angular_spectrum_and_convolution_on_composite_grids(pa):
    """free space on a 1000 x 1500 grid and an image-chain convolution on a 600 x 1000 one: every transform of the chains on the mixed-radix kernel"""
    rng = np.random.default_rng(1001)
    x = crandn(rng, (1000, 1500))
    ref = O.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1)
    assert rel_max(tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1)), ref) < TOL64
    img = rng.standard_normal((600, 1000))
    psf = rng.random((600, 1000))
    want = np.fft.fftshift(np.fft.ifft2(np.fft.fft2(np.fft.ifftshift(img)) * np.fft.fft2(np.fft.ifftshift(psf)))).real
    got = tonp(pa.convolution.conv(img, psf))
    assert rel_max(got, want) < TOL64


def test_focus_6000_on_the_mixed_radix_kernel(pa):
    """a length in (4096, 8192]: round 2 convolved BOTH axes at 16384 points for these; now one kernel per axis (timing printed)"""
    rng = np.random.default_rng(6000)
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
    print('focus 8000^2 complex64 (mixed radix): %.2f ms' % ev0.elapsed_time(ev1))


def test_otf_and_padded_focus_on_composite_grids(pa):
    """the SURVEY 8(f) wrappers and a Q = 1.5 pad on composite grids: `mtf_from_psf` / `ptf_from_psf` of a real 600 x 1000 PSF (real input
    read as it is by the mixed-radix first stage, centre normalisation and |.| / angle by the common epilogue) and
    `Wavefront.focus(Q=1.5)` of a 1000^2 pupil (1500^2 transform with the pad in the load window)"""
    rng = np.random.default_rng(600)
    psf = rng.random((600, 1000)) + 0.01
    F = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(psf)))
    F = F / F[300, 500]
    assert rel_max(tonp(pa.otf.mtf_from_psf(psf, 1.0).data), np.abs(F)) < TOL64
    got = tonp(pa.otf.ptf_from_psf(psf, 1.0).data)
    big = np.abs(F) > 1e-3
    assert np.max(np.abs(np.angle(np.exp(1j * (got - np.angle(F))))[big])) < 1e-8
    x = crandn(rng, (1000, 1000), np.complex64)
    ref = O.focus(x.astype(np.complex128), 1.5)
    assert ref.shape == (1500, 1500)
    assert rel_max(tonp(pa.propagation.focus(x, 1.5)), ref) < TOL32
    assert rel_max(tonp(pa.propagation.focus_intensity(x, 1.5)), O.intensity(ref)) < 2 * TOL32


def test_composite_length_beside_a_split_length(pa):
    """a composite length the mixed-radix kernel owns (96 = 3 * 32) beside a power of two that needs the radix-2 step (native length
    lowered to 64: 128 splits) takes the radix-R path on BOTH axes (such shapes -- 1536 x 16384 at full size -- fell to the
    both-axes Bluestein form); against numpy, both orientations and both precisions"""
    from prysm_amd import _lib, _ops
    rng = np.random.default_rng(21)
    with _lib.tuning_local(big_native_log=6):
        for shape in ((96, 128), (128, 96), (160, 256)):
            for dtype, tol in ((np.complex64, TOL32), (np.complex128, TOL64)):
                x = crandn(rng, shape, dtype)
                got = _ops.fft2(torch.from_numpy(x).cuda(), direction=-1, scale=1.0).cpu().numpy()
                assert rel_max(got, np.fft.fft2(x.astype(np.complex128))) < tol, (shape, dtype)
                ref = O.focus(x.astype(np.complex128), 1)
                assert rel_max(tonp(pa.propagation.focus(x, 1)), ref) < tol, (shape, dtype)


@pytest.mark.parametrize('shape,dtype,tol', [((1000, 1500), np.complex128, TOL64), ((600, 750), np.complex64, TOL32),
                                             ((1000, 1024), np.complex128, TOL64), ((360, 2048), np.complex64, TOL32),
                                             ((105, 154), np.complex128, TOL64)])
def test_angular_spectrum_on_composite_grids(pa, shape, dtype, tol):
    """angular_spectrum / its adjoint / tf= on grids whose column length is composite (primes <= 13) -- three passes with the
    mixed-radix middle pass (forward stages, x H, transposed stages in LDS) -- against the oracle and against the composed route
    (two pm_fft2 calls, knob mix_fused = 0); row lengths composite and powers of two; Q = 1 and a padded Q = 2 input"""
    from prysm_amd import _lib
    rng = np.random.default_rng(shape[0] + shape[1])
    prec = pa.config.precision
    pa.config.precision = 32 if dtype == np.complex64 else 64
    try:
        x = crandn(rng, shape, dtype)
        ref = O.angular_spectrum(x.astype(np.complex128), O.HeNe, 0.01, 10.0, Q=1)
        got = tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1))
        assert got.dtype == dtype and rel_max(got, ref) < tol
        with _lib.tuning_local(mix_fused=0):
            comp = tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1))
        assert rel_max(comp, ref) < tol and rel_max(got, comp) < 2 * tol
        # padded input (only the stored rows are transformed in the first pass; the middle pass synthesises the zero rows), and
        # the adjoint: conj(H) and a crop of the rows / columns in the last pass
        small = np.ascontiguousarray(x[:shape[0] // 2, :shape[1] // 2])
        refq = O.angular_spectrum(small.astype(np.complex128), O.HeNe, 0.01, 10.0, Q=2)
        assert rel_max(tonp(pa.propagation.angular_spectrum(small, O.HeNe, 0.01, 10.0, Q=2)), refq) < tol
        refa = O.angular_spectrum_adjoint(x.astype(np.complex128), O.HeNe, 0.01, 10.0, Q=2)
        assert rel_max(tonp(pa.propagation.angular_spectrum_adjoint(x, O.HeNe, 0.01, 10.0, Q=2)), refa) < tol
        tf = O.angular_spectrum_transfer_function(shape, O.HeNe, 0.01, 7.0)
        reft = O.angular_spectrum(x.astype(np.complex128), O.HeNe, 0.01, 7.0, Q=1, tf=tf)
        assert rel_max(tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 7.0, Q=1, tf=tf.astype(dtype))), reft) < 2 * tol
    finally:
        pa.config.precision = prec


@pytest.mark.parametrize('shape', [(300, 500), (375, 250), (96, 1000)])
def test_conv_on_composite_grids(pa, shape):
    """conv / apply_transfer_functions on composite grids: both rotations ride on the chain -- the input rotation in the first two passes'
    loads, the output rotation as two runs of rows in the last pass -- complex and real objects, against the oracle"""
    from prysm_amd import convolution as C
    rng = np.random.default_rng(sum(shape))
    o = crandn(rng, shape)
    h = crandn(rng, shape)
    assert rel_max(tonp(C.conv(o, h)), O.conv(o, h)) < 1e-9
    orl = rng.standard_normal(shape)
    assert rel_max(tonp(C.conv(orl, h)), O.conv(orl, h)) < 1e-9
    tf = crandn(rng, shape)
    assert rel_max(tonp(C.apply_transfer_functions(o, 1.0, [tf])), O.apply_transfer_functions(o, 1.0, [tf])) < 1e-9
    assert rel_max(tonp(C.apply_transfer_functions(o.astype(np.complex64), 1.0, [tf.astype(np.complex64)], shift=True)),
                   O.apply_transfer_functions(o, 1.0, [tf], shift=True)) < 2e-5


@pytest.mark.parametrize('shape,Q', [((600, 750), 1), ((500, 500), 1.5), ((1000, 1536), 1), ((300, 400), 2)])
def test_pupil_synthesis_in_the_load_on_composite_grids(pa, shape, Q):
    """Wavefront.from_amp_and_phase(...).focus() / .focus_intensity() and the polychromatic driver on grids whose (padded) row length is a
    composite of primes <= 13: the pupil is synthesised by the first stage of the mixed-radix row kernel (the lazy wavefront never
    materialises it), float32 and float64 maps, float / bool / no amplitude, packed maps through the wavelength loop -- vs the oracle"""
    from prysm_amd.polychromatic import polychromatic_psf
    rng = np.random.default_rng(int(shape[0] * Q))
    ampf = (rng.random(shape) * (rng.random(shape) > 0.2)).astype(np.float32)
    ampb = rng.random(shape) > 0.3
    for rd, tol in ((np.float32, 1e-5), (np.float64, TOL64)):
        opd = (150 * rng.standard_normal(shape)).astype(rd)
        for amp in (ampf.astype(rd), ampb, None):
            a64 = np.ones(shape) if amp is None else amp.astype(np.float64)
            pref = O.focus(O.from_amp_and_phase(a64, opd.astype(np.float64), O.HeNe), Q)
            wf = pa.propagation.Wavefront.from_amp_and_phase(amp if amp is not None else np.ones(shape, dtype=rd), opd, O.HeNe, 0.04)
            assert wf._fusable(Q) is not None
            got = tonp(wf.focus(100.0, Q=Q).data)
            assert wf._data is None                    # never materialised
            assert rel_max(got, pref) < tol, (rd, None if amp is None else amp.dtype)
            I = tonp(wf.focus_intensity(100.0, Q=Q).data)
            assert rel_max(I, O.intensity(pref)) < 2 * tol
    opd = (150 * rng.standard_normal(shape)).astype(np.float32)
    wv, wt = np.linspace(0.5, 0.7, 5), np.linspace(1.0, 2.0, 5)
    poly = tonp(polychromatic_psf(ampf, opd, wv, wt, 0.04, 100.0, Q=Q))
    pw = sum(w * O.intensity(O.focus(O.from_amp_and_phase(ampf.astype(np.float64), opd.astype(np.float64), float(l)), Q)) for l, w in zip(wv, wt))
    assert poly.dtype == np.float32 and rel_max(poly, pw) < 2e-5


@pytest.mark.parametrize('shape', [(10000, 96), (96, 10000), (9000, 3000), (12000, 128), (64, 20000)])
def test_composite_lengths_above_8192(pa, shape):
    """lengths above 8192 whose cofactor of 2 .. 7 is a composite the mixed-radix kernel takes (10000 = 2 x 5000, 9000 = 2 x 4500,
    12000 = 2 x 6000, 20000 = 4 x 5000) run as one radix-R step around mixed-radix sub-transforms instead of a Bluestein convolution at
    32768 points -- beside short composite / power-of-two axes; fft2, ifft2 and the focus family (rotations, pad window) vs numpy"""
    from prysm_amd import _ops
    rng = np.random.default_rng(sum(shape))
    for dtype, tol in ((np.complex64, TOL32), (np.complex128, TOL64)):
        x = crandn(rng, shape, dtype)
        xd = torch.from_numpy(x).cuda()
        assert rel_max(_ops.fft2(xd, direction=-1, scale=1.0).cpu().numpy(), np.fft.fft2(x.astype(np.complex128))) < tol, dtype
        M, N = shape
        assert rel_max(_ops.fft2(xd, direction=+1, scale=1.0 / (M * N)).cpu().numpy(), np.fft.ifft2(x.astype(np.complex128))) < tol
        assert rel_max(tonp(pa.propagation.focus(x, 1)), O.focus(x.astype(np.complex128), 1)) < tol
    small = crandn(rng, (shape[0] // 2, shape[1] // 2), np.complex128)
    assert rel_max(tonp(pa.propagation.focus(small, 2)), O.focus(small, 2)) < TOL64
    xr = rng.standard_normal(shape)
    assert rel_max(_ops.fft2(torch.from_numpy(xr).cuda(), direction=-1, scale=1.0).cpu().numpy(), np.fft.fft2(xr)) < TOL64


@pytest.mark.parametrize('shape', [(1020, 1900), (323, 380), (272, 4913), (2048, 1020)])
def test_lengths_with_the_primes_17_and_19(pa, shape):
    """radices 17 and 19 in the mixed-radix kernel (round 4): 1020 = 6 x 10 x 17, 1900 = 10 x 10 x 19, 323 = 17 x 19, 4913 = 17^3 no longer
    convolve through Bluestein; fft2 / ifft2 / focus in both precisions and the fused chain (column lengths 1020, 323) vs numpy"""
    from prysm_amd import _ops
    rng = np.random.default_rng(sum(shape))
    M, N = shape
    for dtype, tol in ((np.complex64, TOL32), (np.complex128, TOL64)):
        x = crandn(rng, shape, dtype)
        xd = torch.from_numpy(x).cuda()
        assert rel_max(_ops.fft2(xd, direction=-1, scale=1.0).cpu().numpy(), np.fft.fft2(x.astype(np.complex128))) < tol, dtype
        assert rel_max(_ops.fft2(xd, direction=+1, scale=1.0 / (M * N)).cpu().numpy(), np.fft.ifft2(x.astype(np.complex128))) < tol
        assert rel_max(tonp(pa.propagation.focus(x, 1)), O.focus(x.astype(np.complex128), 1)) < tol
    x = crandn(rng, shape)
    assert rel_max(tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1)), O.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1)) < TOL64


@pytest.mark.parametrize('shape,rdt', [((1000, 1000), np.float32), ((3000, 3000), np.float32), ((1001, 1000), np.float64), ((300, 1536), np.float64),
                                       ((64, 64), np.float32), ((1, 30), np.float64), ((997, 2018), np.float32), ((2048, 1000), np.float64)])
def test_fft2_real_on_any_even_width_vs_numpy(pa, shape, rdt):
    """_ops.fft2_real: the real array read as complex pairs, a half-size pm_fft2 (mixed-radix, engine, Bluestein or direct -- whatever
    the lengths take) and pm_r2c_untangle; plain and centred (ifftshift in / fftshift out), complex output and the three real
    epilogues, with and without the division by the DC bin -- against numpy fp64"""
    from prysm_amd import _lib, _ops
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    x = (rng.random(shape) + 0.05).astype(rdt)
    xt = torch.from_numpy(x).cuda()
    x64 = x.astype(np.float64)
    M, N = shape
    tol = 1e-10 if rdt == np.float64 else 2e-5
    plain = np.fft.fft2(x64)
    assert rel_max(tonp(_ops.fft2_real(xt)), plain) < tol
    sh = (M // 2, N // 2)
    cen = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(x64)))
    got = tonp(_ops.fft2_real(xt, in_shift=sh, out_shift=sh))
    assert got.dtype == (np.complex128 if rdt == np.float64 else np.complex64) and rel_max(got, cen) < tol
    nrm = cen / cen[M // 2, N // 2]
    assert rel_max(tonp(_ops.fft2_real(xt, in_shift=sh, out_shift=sh, norm_dc=True)), nrm) < tol
    assert rel_max(tonp(_ops.fft2_real(xt, in_shift=sh, out_shift=sh, norm_dc=True, epilogue=_lib.PM_EPI_ABS)), np.abs(nrm)) < tol
    assert rel_max(tonp(_ops.fft2_real(xt, in_shift=sh, out_shift=sh, scale=0.5, epilogue=_lib.PM_EPI_ABS2)), np.abs(0.5 * cen) ** 2) < 2 * tol
    ang = tonp(_ops.fft2_real(xt, in_shift=sh, out_shift=sh, norm_dc=True, epilogue=_lib.PM_EPI_ARG))
    strong = np.abs(nrm) > 1e-3 * np.abs(nrm).max()          # the angle of a bin at rounding level is noise in any implementation
    dphi = np.angle(np.exp(1j * (ang - np.angle(nrm))))
    assert np.max(np.abs(dphi[strong])) < (1e-8 if rdt == np.float64 else 2e-3)
    # a view into a wider array (row pitch != width) and an odd width refused
    wide = torch.zeros((M, N + 6), dtype=xt.dtype, device='cuda')
    wide[:, 2:N + 2] = xt
    assert rel_max(tonp(_ops.fft2_real(wide[:, 2:N + 2])), plain) < tol
    if N > 2:
        assert not _ops.real_pairs_ok(xt[:, :N - 1])
        with pytest.raises(ValueError):
            _ops.fft2_real(xt[:, :N - 1])


@pytest.mark.parametrize('n,rdt', [(1000, np.float32), (3000, np.float32), (1500, np.float64)])
def test_mtf_ptf_otf_on_composite_grids(pa, n, rdt):
    """prysm/otf.py on a real PSF whose size is not a power of two: one half-size transform + the untangling sweep with the
    normalisation and |.| / angle fused (no elementwise torch sweeps), equal to the reference's formula in fp64"""
    from prysm_amd import otf
    rng = np.random.default_rng(n)
    yy, xx = np.mgrid[:n, :n] - n // 2
    psf = (np.exp(-(xx ** 2 + yy ** 2) / (2 * 9.0 ** 2)) + 0.02 * rng.random((n, n))).astype(rdt)
    F = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(psf.astype(np.float64))))
    nrm = F / F[n // 2, n // 2]
    tol = 1e-10 if rdt == np.float64 else 2e-5
    m = otf.mtf_from_psf(psf, 1.0)
    assert tuple(m.data.shape) == (n, n) and not m.data.is_complex() and abs(m.dx - 1000 / n) < 1e-12
    assert np.max(np.abs(tonp(m.data) - np.abs(nrm))) < tol
    o = otf.otf_from_psf(psf, 1.0)
    assert np.max(np.abs(tonp(o.data) - nrm)) < tol
    p = tonp(otf.ptf_from_psf(psf, 1.0).data)
    strong = np.abs(nrm) > 1e-3
    assert np.max(np.abs(np.angle(np.exp(1j * (p - np.angle(nrm))))[strong])) < (1e-8 if rdt == np.float64 else 2e-3)
    mm, pp, oo = otf.mtf_ptf_otf_from_psf(psf, 1.0)
    assert np.max(np.abs(tonp(mm.data) - np.abs(nrm))) < tol and np.max(np.abs(tonp(oo.data) - nrm)) < tol
    data, df = otf.transform_psf(psf, 1.0)
    assert rel_max(tonp(data), F) < tol and abs(df - 1000 / n) < 1e-12
    mtf2, raw = otf.mtf_from_psf(psf, 1.0, return_more=True)      # the composed route still answers return_more
    assert np.max(np.abs(tonp(mtf2.data) - np.abs(nrm))) < tol and rel_max(tonp(raw), F) < tol


@pytest.mark.parametrize('cdt', [np.complex64, np.complex128])
@pytest.mark.parametrize('n', CE_LENGTHS)
def test_composite_engine_plans_vs_numpy_and_general_kernel(pa, n, cdt):
    """Each plan as the row pass (short columns beside it) and as the column pass (ragged tiles: a column count that is no multiple of
    any tile width), plain / rotated / zero-padded / inverse, with the engine on and off (knob mix_engine)."""
    from prysm_amd import _lib, _ops
    rng = np.random.default_rng(n)
    tol = TOL32 if cdt == np.complex64 else TOL64
    other = 90 if n > 3000 else 250       # 90 and 250 run on the general kernel: one pass of each transform is the engine's
    cases = [((other, n), None, (0, 0), (0, 0), (0, 0), -1),
             ((n, other + 1), None, (0, 0), (n // 2, 3), (n // 2, 5), -1),
             ((n, other + 1), None, (0, 0), (1, 0), (0, 2), +1),
             ((other, n // 2), (other, n), (0, n // 4), (0, n // 2), (3, n // 2), -1),
             ((n // 2 + 1, other), (n, other), (n // 4, 0), (n // 2, 0), (n // 2, 0), +1)]
    if n <= 2000:
        cases.append(((n, n), None, (0, 0), (n // 2, n // 2), (n // 2, n // 2), -1))
    for xs, shape, in_off, in_shift, out_shift, direction in cases:
        shape = shape or xs
        x = (rng.standard_normal(xs) + 1j * rng.standard_normal(xs)).astype(cdt)
        want = _ce_ref(x, shape, in_off, in_shift, out_shift, direction)
        xd = torch.from_numpy(x).cuda()
        got = {}
        for eng in (1, 0):
            with _lib.tuning_local(mix_engine=eng):
                got[eng] = _ops.fft2(xd, direction=direction, scale=1.0, shape=shape, in_off=in_off, in_shift=in_shift, out_shift=out_shift).cpu().numpy()
            assert rel_max(got[eng], want) < tol, (n, cdt.__name__, xs, shape, eng)
        assert rel_max(got[1], got[0]) < tol


@pytest.mark.parametrize('n,cdt', [(1000, np.complex64), (1500, np.complex128), (3000, np.complex64)])
def test_composite_engine_intensity_epilogues(pa, n, cdt):
    """|.|^2 and weight |.|^2 accumulated (Wavefront.intensity, the polychromatic sum) in the engine's column store."""
    from prysm_amd import _lib, _ops
    rng = np.random.default_rng(n + 1)
    m = 300
    x = (rng.standard_normal((n, m)) + 1j * rng.standard_normal((n, m))).astype(cdt)
    f = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(x.astype(np.complex128)))) * 0.01
    xd = torch.from_numpy(x).cuda()
    kw = dict(direction=-1, scale=0.01, in_shift=(n // 2, m // 2), out_shift=(n // 2, m // 2))
    i1 = _ops.fft2(xd, epilogue=_lib.PM_EPI_ABS2, **kw)
    assert i1.dtype == (torch.float32 if cdt == np.complex64 else torch.float64)
    assert rel_max(i1.cpu().numpy(), np.abs(f) ** 2) < (4e-5 if cdt == np.complex64 else 1e-10)
    acc = i1.clone()
    _ops.fft2(xd, epilogue=_lib.PM_EPI_ABS2_ACCUM, out=acc, weight=0.5, **kw)
    assert rel_max(acc.cpu().numpy(), 1.5 * np.abs(f) ** 2) < (4e-5 if cdt == np.complex64 else 1e-10)


def test_composite_engine_through_the_wavefront_api(pa):
    """focus / unfocus of a 1000^2 and a 1500 x 2000 field (prysm/propagation/fft.py:7-45) against the oracle: the route the users take."""
    from prysm_amd import propagation as P
    rng = np.random.default_rng(5)
    for shape, cdt, tol in (((1000, 1000), np.complex64, TOL32), ((1500, 2000), np.complex128, TOL64)):
        x = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(cdt)
        got = tonp(P.focus(torch.from_numpy(x).cuda(), 1))
        assert rel_max(got, O.focus(x.astype(np.complex128), 1)) < tol
        back = tonp(P.unfocus(torch.from_numpy(got).cuda(), 1))
        assert rel_max(back, x) < 4 * tol


def test_composite_engine_hands_wide_arrays_to_the_general_kernel(pa):
    """The engine's kernels address with one unsigned 32-bit byte offset per lane (2 n pitch s < 2^32, csrc/pm_internal.h ce_fits32); a
    column transform of 8000 points down a 20000-wide complex128 array is past that and inside the general kernel's range: both widths
    against numpy on sampled columns."""
    from prysm_amd import _ops
    n = 8000
    for width in (16000, 20000):        # 4.1e9 and 5.1e9 bytes of 2 n pitch s
        g = torch.Generator(device='cuda').manual_seed(width)
        x = torch.randn(n, width, dtype=torch.float64, device='cuda', generator=g).to(torch.complex128)
        x += 1j * torch.randn(n, width, dtype=torch.float64, device='cuda', generator=g)
        y = _ops.fft1(x, n, axis=0)
        cols = [0, 3, width // 2 + 1, width - 1]
        want = np.fft.fft(x[:, cols].cpu().numpy(), axis=0)
        assert rel_max(y[:, cols].cpu().numpy(), want) < TOL64, width
        del x, y
        torch.cuda.empty_cache()


@pytest.mark.parametrize('n,Q', [(1000, 1), (750, 2), (1536, 1), (2000, 1)])
def test_composite_engine_synthesises_the_pupil_in_its_row_loads(pa, n, Q):
    """Wavefront.from_amp_and_phase(amp, opd, wvl).focus(efl, Q) on composite grids (prysm/propagation/wavefront.py:58-79, 478-504): the
    complex64 pupil is formed in the register engine's row loads from the OPD map + amplitude and from packed pairs, padded by Q, against
    the oracle -- and equal to the general kernel's synthesis."""
    from prysm_amd import _lib, _ops
    rng = np.random.default_rng(n + Q)
    amp = (rng.random((n, n)) > 0.25).astype(np.float32)
    opd = (150 * rng.standard_normal((n, n))).astype(np.float32)
    wvl = 0.6328
    want = O.focus(O.from_amp_and_phase(amp, opd.astype(np.float64), wvl), Q)
    N = n * Q
    off = (N - n) // 2
    k = 2 * np.pi / wvl / 1e3
    kw = dict(direction=-1, scale=1.0 / np.sqrt(N * N), shape=(N, N), in_off=(off, off), in_shift=(N // 2, N // 2), out_shift=(N // 2, N // 2))
    od, ad = torch.from_numpy(opd).cuda(), torch.from_numpy(amp).cuda()
    got = {}
    for eng in (1, 0):
        with _lib.tuning_local(mix_engine=eng):
            got[eng] = _ops.fft2(od, synth=(ad, k), **kw).cpu().numpy()
            packed = _ops.fft2(_ops.pack_amp_opd(ad, od), synth=('packed', k), **kw).cpu().numpy()
        assert got[eng].dtype == np.complex64
        assert rel_max(got[eng], want) < 2e-5 and rel_max(packed, want) < 2e-5, (n, Q, eng)
    assert rel_max(got[1], got[0]) < 2e-5


@pytest.mark.parametrize('shape,cdt,B', [((500, 768), np.complex64, 5), ((1000, 900), np.complex128, 3), ((384, 384), np.complex64, 17)])
def test_composite_engine_runs_a_stack_as_one_launch_pair(pa, shape, cdt, B):
    """(B, m, n) stacks on composite grids (the reference's multi-field batches, prysm/x/polarization.py:478-553): both passes of every
    field in ONE launch pair on the register engine (grid.y = fields), with rotations and the |.|^2 epilogue, equal to numpy per field and
    to the same stack with the engine off (field by field on the general kernel)."""
    from prysm_amd import _lib, _ops
    rng = np.random.default_rng(B)
    m, n = shape
    tol = TOL32 if cdt == np.complex64 else TOL64
    x = (rng.standard_normal((B, m, n)) + 1j * rng.standard_normal((B, m, n))).astype(cdt)
    want = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(x.astype(np.complex128), axes=(1, 2))), axes=(1, 2)) / np.sqrt(m * n)
    xd = torch.from_numpy(x).cuda()
    kw = dict(direction=-1, scale=1.0 / np.sqrt(m * n), in_shift=(m // 2, n // 2), out_shift=(m // 2, n // 2))
    got = {}
    for eng in (1, 0):
        with _lib.tuning_local(mix_engine=eng):
            got[eng] = _ops.fft2(xd, **kw).cpu().numpy()
            inten = _ops.fft2(xd, epilogue=_lib.PM_EPI_ABS2, **kw).cpu().numpy()
        assert rel_max(got[eng], want) < tol and rel_max(inten, np.abs(want) ** 2) < 8 * tol, (shape, eng)
    assert rel_max(got[1], got[0]) < tol
    # a strided stack (every other field of a larger one)
    big = torch.from_numpy(np.concatenate([x, x[::-1]], axis=0)).cuda()
    sub = _ops.fft2(big[::2], **kw).cpu().numpy()
    ref = np.concatenate([x, x[::-1]], axis=0)[::2]
    assert rel_max(sub, np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(ref.astype(np.complex128), axes=(1, 2))), axes=(1, 2)) / np.sqrt(m * n)) < tol


def test_polychromatic_psf_on_small_composite_grids_takes_stacks_by_default(pa):
    """A 500^2 pupil, 12 wavelengths (docs/source/how-tos/Polychromatic Propagation.ipynb on a decimal grid): the default now runs the
    wavelengths as stacks on the composite register engine (one launch pair per stack); same image as the per-wavelength loop and as
    the oracle's sum."""
    from prysm_amd import _ops
    from prysm_amd.polychromatic import polychromatic_psf
    rng = np.random.default_rng(12)
    n = 500
    amp = (rng.random((n, n)) > 0.3).astype(np.float32)
    opd = (120 * rng.standard_normal((n, n))).astype(np.float32)
    wv, wt = np.linspace(0.5, 0.7, 12), np.linspace(1.0, 2.0, 12)
    assert _ops.on_register_engine(n, n, torch.complex64) and not _ops.on_register_engine(600, 600, torch.complex64)
    got = tonp(polychromatic_psf(amp, opd, wv, wt, 0.04, 100.0, Q=1))
    loop = tonp(polychromatic_psf(amp, opd, wv, wt, 0.04, 100.0, Q=1, batched=False, spectral=False))
    want = sum(w * O.intensity(O.focus(O.from_amp_and_phase(amp, opd.astype(np.float64), float(l)), 1)) for l, w in zip(wv, wt))
    assert rel_max(got, want) < 2e-5 and rel_max(loop, want) < 2e-5 and rel_max(got, loop) < 2e-5
