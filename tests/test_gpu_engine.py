"""GPU parity tests, by component: the power-of-two FFT engine and the fused chains around it -- layouts of the tiled intermediate, the fold, the three-pass
fft2 x H ifft2 chain and its middle-pass forms, adjoints at BASELINE sizes, pupil synthesis in the load, padding modes, knobs and
per-thread tuning (csrc/fft_engine.h, fft_io.h, fft_kernels.h, capi.hip).

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


def test_packed_amp_opd_synthesis_equals_two_array_synthesis(pa):
    """PM_FLAG_SYNTH_PACKED: the pupil synthesised from (amplitude, OPD) pairs read as one 8-byte element is bit for bit the
    pupil synthesised from the two arrays (same arithmetic, different loads), folded and unfolded, padded and not"""
    from prysm_amd import _ops
    P = pa.propagation
    rng = np.random.default_rng(21)
    for n, Q in ((256, 1), (256, 2), (4096, 1)):
        amp = torch.from_numpy((rng.random((n, n)) > 0.3).astype(np.float32)).cuda()
        opd = torch.from_numpy((300 * rng.standard_normal((n, n))).astype(np.float32)).cuda()
        k = 2 * np.pi / 0.55 / 1e3
        a = P.focus_intensity(opd, Q, synth=(amp, k))
        b = P.focus_intensity(_ops.pack_amp_opd(amp, opd), Q, synth=('packed', k))
        assert torch.equal(a, b), (n, Q)
    ref = O.intensity(O.focus(O.from_amp_and_phase(amp.cpu().numpy().astype(np.float64), opd.cpu().numpy().astype(np.float64), 0.55), 1))
    assert rel_max(tonp(b), ref) < 2e-5


def test_array_orientation_consistency_tilt(pa):
    """arr[y, x]: a positive +y tilt in the pupil moves the PSF to +y and leaves x centred (tests/test_physics.py:56-74)"""
    P = pa.propagation
    N, wvl, Q = 128, .5, 3
    x, y = O.make_xy_grid(N, diameter=2.1)
    r, _ = O.cart_to_polar(x, y)
    amp = O.circle(1, r)
    phs = 1000 * y
    for dt in (np.float64, np.float32):
        wf = P.Wavefront.from_amp_and_phase(amp.astype(dt), phs.astype(dt), wvl, x[0, 1] - x[0, 0])
        psf = tonp(wf.focus(1, Q=Q).intensity)
        idx_y, idx_x = np.unravel_index(psf.argmax(), psf.shape)
        assert idx_x == (N * Q) // 2
        assert idx_y > (N * Q) // 2
        ref = O.intensity(O.focus(O.from_amp_and_phase(amp, phs, wvl), Q))
        assert rel_max(psf, ref) < (TOL64 if dt == np.float64 else 2e-5)


def test_thinlens_hopkins_agree(pa):
    """a weak thin lens in front of the pupil == the matching Hopkins defocus (tests/test_propagation.py:440-460)"""
    P = pa.propagation
    x, y = O.make_xy_grid(128, diameter=11)
    dx = x[0, 1] - x[0, 0]
    r = np.hypot(x, y)
    amp = O.circle(5, r)
    phs = (r / 5) ** 2 * (1.975347661 * O.HeNe * 1000)     # hopkins(0, 2, 0, rho, 0, 1) = rho^2
    psf = tonp(P.Wavefront.from_amp_and_phase(amp, phs, O.HeNe, dx).focus(efl=100, Q=2).intensity)
    no_phs_wf = P.Wavefront.from_amp_and_phase(amp, None, O.HeNe, dx)
    tl = P.Wavefront.thin_lens(10_000, O.HeNe, x, y)
    psf2 = tonp((no_phs_wf * tl).focus(efl=100, Q=2).intensity)
    assert np.allclose(psf, psf2, rtol=1e-5)
    ref = O.intensity(O.focus(O.from_amp_and_phase(amp, phs, O.HeNe), 2))
    assert rel_max(psf, ref) < TOL64


@pytest.mark.parametrize('mode', ['edge', 'reflect', 'symmetric', 'wrap'])
def test_pad2d_modes_match_numpy_pad(pa, mode):
    """fttools.pad2d(mode != 'constant') forwards to np.pad with widths (d - d // 2, d // 2) (prysm/fttools.py:79-98)"""
    from prysm_amd import fttools
    rng = np.random.default_rng(11)
    for shape, out_shape in (((9, 12), (14, 18)), ((5, 4), (17, 21)), ((1, 6), (4, 6)), ((8, 8), (8, 8))):
        for dt in (np.float32, np.complex128, np.bool_):
            a = (rng.random(shape) > 0.5) if dt is np.bool_ else rng.standard_normal(shape).astype(dt)
            diff = [o - i for o, i in zip(out_shape, shape)]
            want = np.pad(a, [(d - d // 2, d // 2) for d in diff], mode=mode)
            got = tonp(fttools.pad2d(a, out_shape=out_shape, mode=mode))
            assert got.dtype == want.dtype and np.array_equal(got, want), (shape, out_shape, dt)
    with pytest.raises(NotImplementedError):
        fttools.pad2d(np.ones((4, 4)), Q=2, mode='no_such_mode')


@pytest.mark.parametrize('mode', ['mean', 'maximum', 'minimum', 'median', 'linear_ramp'])
def test_pad2d_statistical_modes_match_numpy_pad(pa, mode):
    """np.pad's statistical modes and linear_ramp at their defaults (statistics over the whole axis, end value 0), axis by axis"""
    from prysm_amd import fttools
    rng = np.random.default_rng(12)
    for shape, out_shape in (((9, 12), (14, 18)), ((5, 4), (17, 21)), ((1, 6), (4, 6)), ((8, 8), (8, 8)), ((6, 7), (6, 12))):
        for dt in (np.float32, np.float64) + ((np.complex128,) if mode in ('mean', 'linear_ramp') else ()):
            a = rng.standard_normal(shape).astype(dt)
            if dt is np.complex128:
                a = a + 1j * rng.standard_normal(shape)
            diff = [o - i for o, i in zip(out_shape, shape)]
            want = np.pad(a, [(d - d // 2, d // 2) for d in diff], mode=mode)
            got = tonp(fttools.pad2d(a, out_shape=out_shape, mode=mode))
            assert got.dtype == want.dtype and got.shape == want.shape
            assert np.allclose(got, want, rtol=1e-6 if dt is np.float32 else 1e-13, atol=1e-6 if dt is np.float32 else 1e-13), (shape, out_shape, dt)


def test_focus_intensity_rejects_a_mismatched_accumulator(pa):
    P = pa.propagation
    x = torch.randn(64, 64, dtype=torch.complex64, device='cuda')
    good = torch.zeros(128, 128, dtype=torch.float32, device='cuda')
    P.focus_intensity(x, 2, out=good, weight=0.5)
    for bad in (torch.zeros(128, 128, dtype=torch.float64, device='cuda'),       # dtype of another precision
                torch.zeros(64, 64, dtype=torch.float32, device='cuda'),         # shape of another Q
                torch.zeros(128, 256, dtype=torch.float32, device='cuda')[:, ::2],   # last axis not contiguous
                torch.zeros(128, 128, dtype=torch.float32)):                     # host tensor
        with pytest.raises(ValueError):
            P.focus_intensity(x, 2, out=bad, weight=0.5)
    # a row-strided view is fine (only the leading dimension is read)
    wide = torch.zeros(128, 160, dtype=torch.float32, device='cuda')
    P.focus_intensity(x, 2, out=wide[:, :128], weight=0.5)
    assert rel_max(tonp(wide[:, :128]), tonp(good)) < 1e-6 and float(wide[:, 128:].abs().max()) == 0.0


def test_fused_pupil_synthesis_complex128(pa):
    """PM_FLAG_SYNTH_INPUT for float64 maps (complex128 transforms: fp64 sincospi per sample inside the row pass): the lazy wavefront
    stays lazy, the result is bit for bit the separate synthesis kernel + transform, equals the oracle, bool / float32 / float64 / no
    amplitude, folded (4096 rows) and not, packed pairs too; and the polychromatic loop on float64 maps"""
    from prysm_amd import _ops
    from prysm_amd.polychromatic import polychromatic_psf
    P = pa.propagation
    rng = np.random.default_rng(15)
    for n, Q in ((256, 1), (128, 2), (4096, 1)):
        x, y = O.make_xy_grid(n, diameter=10)
        r, _ = O.cart_to_polar(x, y)
        opd = O.hopkins_w040(r / 5, 800.0) + 30 * rng.standard_normal((n, n))
        for amp in (O.circle(5, r), rng.random((n, n)).astype(np.float32), rng.random((n, n)), None):
            wf = P.Wavefront.from_amp_and_phase(amp, opd, 0.55, 10.0 / n)
            assert wf._fusable(Q) is not None
            got = wf.focus(100.0, Q)
            assert got.data.dtype == torch.complex128 and wf._data is None
            want = O.focus(O.from_amp_and_phase(np.ones((n, n)) if amp is None else amp, opd, 0.55), Q)
            assert rel_max(tonp(got), want) < TOL64
            inten = wf.focus_intensity(100.0, Q)
            assert rel_max(tonp(inten), O.intensity(want)) < 4 * TOL64
            if n <= 256:
                field = wf.data                                       # materialised by the separate kernel
                assert torch.equal(P.focus(field, Q), got.data)       # same arithmetic, different loads
                a_dev = None if amp is None else torch.from_numpy(np.asarray(amp)).cuda()
                pk = _ops.pack_amp_opd(a_dev, torch.from_numpy(opd).cuda())
                assert pk.dtype == torch.complex128
                assert torch.equal(P.focus_intensity(pk, Q, synth=('packed', 2 * np.pi / 0.55 / 1e3)), inten.data)
    n = 256
    x, y = O.make_xy_grid(n, diameter=10)
    r, _ = O.cart_to_polar(x, y)
    amp, opd = O.circle(5, r), O.hopkins_w040(r / 5, 300.0)
    wv, wt = np.linspace(0.5, 0.7, 5), np.linspace(1.0, 2.0, 5)
    want = sum(w * O.intensity(O.focus(O.from_amp_and_phase(amp, opd, float(l)), 2)) for l, w in zip(wv, wt))
    got = polychromatic_psf(amp, opd, wv, wt, 10.0 / n, 100.0, Q=2)
    assert got.dtype == torch.float64 and rel_max(tonp(got), want) < 4 * TOL64


@pytest.mark.parametrize('mode', [0, 3])      # (1 and 2 lost their measurements and left the library in round 5: experiments/README.md)
@pytest.mark.parametrize('n,dtype,tol', [(4096, np.complex128, TOL64), (4096, np.complex64, TOL32), (2048, np.complex128, TOL64),
                                         (2048, np.complex64, TOL32)])
def test_angular_spectrum_middle_pass_forms(pa, mode, n, dtype, tol):
    """angular_spectrum(x, Q = 1) -- config 3 at 4096^2 complex128 -- with the middle pass in each of its forms; the result must
    not depend on the form beyond rounding, and all of them match the oracle"""
    from prysm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(n + mode)
    x = crandn(rng, (n, n), dtype)
    assert lib.pm_set_tuning(b'colmul_mode', mode) == 0
    prec = pa.config.precision
    pa.config.precision = 32 if dtype == np.complex64 else 64
    try:
        got = tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1))
    finally:
        lib.pm_set_tuning(b'colmul_mode', 3)
        pa.config.precision = prec
    ref = O.angular_spectrum(x.astype(np.complex128), O.HeNe, 0.01, 10.0, Q=1)
    assert got.dtype == dtype
    assert rel_max(got, ref) < tol


@pytest.mark.parametrize('mode', [0, 3])
def test_angular_spectrum_tf_and_adjoint_middle_pass_forms(pa, mode):
    """tf= (a full multiplier: the persistent form declines it and the call must still be right) and the adjoint (conj H) at 4096^2"""
    from prysm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77 + mode)
    x = crandn(rng, (4096, 4096))
    tf = O.angular_spectrum_transfer_function((4096, 4096), O.HeNe, 0.01, 10.0)
    assert lib.pm_set_tuning(b'colmul_mode', mode) == 0
    try:
        got_tf = tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1, tf=tf))
        got_adj = tonp(pa.propagation.angular_spectrum_adjoint(x, O.HeNe, 0.01, 10.0, Q=1))
    finally:
        lib.pm_set_tuning(b'colmul_mode', 3)
    assert rel_max(got_tf, O.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1, tf=tf)) < TOL64
    assert rel_max(got_adj, O.angular_spectrum_adjoint(x, O.HeNe, 0.01, 10.0, Q=1)) < TOL64


def test_focus_adjoint_4096_vs_oracle(pa):
    """focus_adjoint of a 4096^2 complex64 focal-plane gradient: Q = 2 (crop to the 2048^2 pupil in the store window) and Q = 1"""
    P = pa.propagation
    rng = np.random.default_rng(40962)
    g = crandn(rng, (4096, 4096), np.complex64)
    g64 = g.astype(np.complex128)
    for Q in (2, 1):
        got = tonp(P.focus_adjoint(g, Q))
        ref = O.focus_adjoint(g64, Q)
        assert got.shape == ref.shape and got.dtype == np.complex64
        assert rel_max(got, ref) < TOL32
    # <focus(x), g> = <x, focus_adjoint(g)> at size (x 2048^2, Q = 2)
    x = crandn(rng, (2048, 2048), np.complex64)
    lhs = np.vdot(tonp(P.focus(x, 2)).astype(np.complex128), g64)
    rhs = np.vdot(x.astype(np.complex128), tonp(P.focus_adjoint(g, 2)).astype(np.complex128))
    assert abs(lhs - rhs) / abs(lhs) < 1e-4


def test_angular_spectrum_adjoint_4096_c128_vs_oracle(pa):
    P = pa.propagation
    rng = np.random.default_rng(40963)
    g = crandn(rng, (4096, 4096))
    got = tonp(P.angular_spectrum_adjoint(g, O.HeNe, 0.01, 10.0, Q=1))
    assert rel_max(got, O.angular_spectrum_adjoint(g, O.HeNe, 0.01, 10.0, Q=1)) < TOL64
    x = crandn(rng, (4096, 4096))
    lhs = np.vdot(tonp(P.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1)), g)
    rhs = np.vdot(x, got)
    assert abs(lhs - rhs) / abs(lhs) < 1e-10


@pytest.mark.parametrize('dtype,rdtype,tol', [(np.complex64, np.float32, 1e-6), (np.complex128, np.float64, 1e-14)])
def test_intensity_adjoint_one_sweep(pa, dtype, rdtype, tol):
    """Wavefront.intensity_adjoint = 2 Ibar E (wavefront.py:282-298) through pm_rmul, at 4096^2 and on a ragged shape; other
    dtype combinations keep the composed form"""
    P = pa.propagation
    rng = np.random.default_rng(9)
    for shape in ((4096, 4096), (33, 50)):
        E = crandn(rng, shape, dtype)
        ib = rng.random(shape).astype(rdtype)
        W = P.Wavefront(E, 0.6328, 1.0, space='psf')
        got = tonp(W.intensity_adjoint(ib))
        ref = 2 * ib.astype(np.float64) * E.astype(np.complex128)
        assert got.dtype == dtype and rel_max(got, ref) < tol
    E = crandn(rng, (16, 16), np.complex64)
    ib64 = rng.random((16, 16))                      # float64 gradient on a complex64 field: numpy promotes, so do we
    got = tonp(P.Wavefront(E, 0.6328, 1.0, space='psf').intensity_adjoint(ib64))
    assert rel_max(got, 2 * ib64 * E.astype(np.complex128)) < 1e-6


def test_config_precision_16_runs_at_float32(pa):
    """config.precision = 16 (accepted as the reference accepts it): synthesised arrays are complex64 / float32"""
    P = pa.propagation
    prec = pa.config.precision
    pa.config.precision = 16
    try:
        assert pa.config.precision is np.float16 and pa.config.precision_complex is np.complex64
        tf = P.angular_spectrum_transfer_function((64, 64), 0.6328, 0.01, 5.0)
        assert tonp(tf).dtype == np.complex64
        ex = P.prepare_executor(0.05, (64, 64), 1.0, (32, 32), 0.6328, 100.0)
        assert ex.Ex.dtype == torch.complex64
        x = crandn(np.random.default_rng(1), (64, 64), np.complex64)
        ref = O.prepare_executor(0.05, (64, 64), 1.0, (32, 32), 0.6328, 100.0)(x.astype(np.complex128))
        assert rel_max(tonp(P.focus_dft(x, ex)), ref) < TOL32_MDFT
    finally:
        pa.config.precision = prec


@pytest.mark.parametrize('M,N,m_in', [(8192, 64, 8192), (8192, 32, 5000), (4096, 128, 4096), (4096, 64, 1000), (2048, 256, 777),
                                      (1024, 512, 1024)])
@pytest.mark.parametrize('dtype,tol', [(np.complex64, 2e-5), (np.complex128, 1e-11)])
def test_fused_chain_lean_middle_pass_shapes(pa, M, N, m_in, dtype, tol):
    """window(ifft2(fft2(pad(x)) H)) on tall arrays: the lean middle pass at 1024 ... 8192-point columns (512- and 1024-thread tiles),
    zero-padded input windows (rows synthesised in the load), separable and full multipliers, conj H, a cropped output -- against
    numpy in fp64"""
    from prysm_amd import _ops
    rng = np.random.default_rng(M + N + m_in)
    x = crandn(rng, (m_in, N), dtype)
    off = ((M - m_in + 1) // 2, 0)
    P = np.zeros((M, N), np.complex128)
    P[off[0]:off[0] + m_in] = x
    F = np.fft.fft2(P)
    hy, hx = np.exp(1j * rng.standard_normal(M)).astype(dtype), np.exp(1j * rng.standard_normal(N)).astype(dtype)
    H = crandn(rng, (M, N), dtype)
    xt = torch.from_numpy(x).cuda()
    sc = 1.0 / (M * N)
    # separable multiplier, full output
    got = tonp(_ops.fft2_mul_ifft2(xt, scale=sc, mul=torch.from_numpy(hy).cuda(), mul_x=torch.from_numpy(hx).cuda(), shape=(M, N), in_off=off))
    ref = np.fft.ifft2(F * np.outer(hy.astype(np.complex128), hx.astype(np.complex128)))
    assert rel_max(got, ref) < tol
    # full multiplier, conjugated, output cropped to the input window
    got = tonp(_ops.fft2_mul_ifft2(xt, scale=sc, mul=torch.from_numpy(H).cuda(), mul_conj=True, shape=(M, N), in_off=off,
                                   out_shape=(m_in, N), out_off=off))
    ref = np.fft.ifft2(F * np.conj(H.astype(np.complex128)))[off[0]:off[0] + m_in]
    assert rel_max(got, ref) < tol


def test_two_threads_with_private_tuning(pa):
    """two host threads, each on its own stream with its own route knobs (pm_set_tuning_local): thread A transforms a composite grid
    on the mixed-radix kernel with the fold off, thread B the same grid through Bluestein (mix = 0) with the fold forced -- 30 rounds
    each, interleaved by the scheduler; every result against numpy, and the process-wide values untouched afterwards"""
    from prysm_amd import _lib, _ops
    lib = _lib.load()
    rng = np.random.default_rng(11)
    xa = crandn(rng, (600, 750), np.complex128)
    xb = crandn(rng, (256, 2048), np.complex64)      # rows of 2048 samples: the forced fold is legal
    wa, wb = np.fft.fft2(xa), np.fft.fft2(xb.astype(np.complex128))
    errs, fails = {}, []

    def worker(name, knobs):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st), _lib.tuning_local(**knobs):
                da, db = torch.from_numpy(xa).cuda(), torch.from_numpy(xb).cuda()
                worst = 0.0
                for _ in range(30):
                    ga = _ops.fft2(da, direction=-1, scale=1.0)
                    gb = _ops.fft2(db, direction=-1, scale=1.0)
                    st.synchronize()
                    worst = max(worst, rel_max(ga.cpu().numpy(), wa) / TOL64, rel_max(gb.cpu().numpy(), wb) / TOL32)
                errs[name] = worst
        except Exception as exc:      # surfaced in the main thread
            fails.append((name, repr(exc)))

    ta = threading.Thread(target=worker, args=('A', dict(mix=1, fold=0)))
    tb = threading.Thread(target=worker, args=('B', dict(mix=0, fold=1)))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not fails, fails
    assert errs['A'] < 1.0 and errs['B'] < 1.0, errs
    # the main thread never took a private copy: it still plans with the process-wide defaults (mix = 1: no Bluestein scratch)
    got = _ops.fft2(torch.from_numpy(xa).cuda(), direction=-1, scale=1.0).cpu().numpy()
    assert rel_max(got, wa) < TOL64
    assert lib.pm_set_tuning_local(b'no_such_knob', 1) == 0      # unknown keys are ignored, as in pm_set_tuning
    lib.pm_reset_tuning_local()


def test_transfer_function_vectors_are_cached_per_scalars(pa):
    """angular_spectrum re-uses the two transfer-function vectors of (shape, wavelength, dx, z): same tensors on a repeat, new ones for
    another distance, results right either way; the materialised transfer function never hands cached storage out"""
    from prysm_amd import _ops
    rng = np.random.default_rng(6)
    x = crandn(rng, (256, 256))
    a = _ops.as_tf_vectors((256, 256), O.HeNe, 0.01, 10.0, torch.complex128)
    b = _ops.as_tf_vectors((256, 256), O.HeNe, 0.01, 10.0, torch.complex128)
    c = _ops.as_tf_vectors((256, 256), O.HeNe, 0.01, 11.0, torch.complex128)
    assert a[0] is b[0] and a[1] is b[1] and c[0] is not a[0]
    for z in (10.0, 11.0, 10.0):
        assert rel_max(tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, z, Q=1)), O.angular_spectrum(x, O.HeNe, 0.01, z, Q=1)) < TOL64
    tf = pa.propagation.angular_spectrum_transfer_function((256, 256), O.HeNe, 0.01, 10.0)
    assert rel_max(tonp(tf), O.angular_spectrum_transfer_function((256, 256), O.HeNe, 0.01, 10.0)) < TOL64
    assert len(_ops._AS_TF_CACHE) <= _ops._AS_TF_CACHE_MAX


@pytest.mark.gpu
def test_knobs_of_removed_variants_are_refused(pa):
    """the variants that lost their measurements left the library in round 5 (experiments/README.md): their knob values answer
    PM_ERR_UNSUPPORTED -- nothing else runs in their place -- and the shipped values are still accepted"""
    from prysm_amd import _lib
    lib = _lib.load()
    try:
        for key, v in ((b'mix_fold', 1), (b'mix_pers', 1), (b'two_units', 1), (b'engine_p8', 1), (b'spectral2', 2), (b'colmul_mode', 1),
                       (b'colmul_mode', 2), (b'gemm_wk', 2), (b'gemm_3m', 0), (b'mix_ablate', 1), (b'spectral_mode', 0)):
            assert lib.pm_set_tuning_local(key, v) == _lib.PM_ERR_UNSUPPORTED, key
        for key, v in ((b'colmul_mode', 3), (b'colmul_mode', 0), (b'gemm_wk', 1), (b'spectral_mode', 3), (b'stagger_group', 1)):
            assert lib.pm_set_tuning_local(key, v) == 0, key
    finally:
        lib.pm_reset_tuning_local()


@pytest.mark.gpu
def test_start_up_stagger_changes_timing_only(pa):
    """The start-up stagger (engine: fft_stagger / fft_stagger_col, 100 + units forces it on single-round launches; mixed-radix column
    kernel: mix_stagger) delays workgroups of the first round by a hash of their index and must not change a bit of any result."""
    from prysm_amd import _lib, _ops
    lib = _lib.load()
    g = torch.Generator(device='cuda').manual_seed(11)
    cases = [((4096, 4096), torch.complex64), ((2048, 2048), torch.complex64), ((1024, 4096), torch.complex128), ((3000, 3000), torch.complex64)]
    try:
        for shape, dt in cases:
            rdt = torch.float32 if dt == torch.complex64 else torch.float64
            x = torch.complex(torch.randn(shape, device='cuda', dtype=rdt, generator=g), torch.randn(shape, device='cuda', dtype=rdt, generator=g))
            h = (shape[0] // 2, shape[1] // 2)
            outs = []
            for r, c, m in ((0, 0, 0), (3, 5, 9), (108, 104, 1)):
                assert lib.pm_set_tuning_local(b'stagger_group', 1 if m == 9 else 0) == 0
                assert lib.pm_set_tuning_local(b'fft_stagger', r) == 0
                assert lib.pm_set_tuning_local(b'fft_stagger_col', c) == 0
                assert lib.pm_set_tuning_local(b'mix_stagger', m) == 0
                outs.append(_ops.fft2(x, direction=-1, scale=1.0, in_shift=h, out_shift=h))
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), shape
    finally:
        lib.pm_reset_tuning_local()


@pytest.mark.parametrize('shape,dtype', [((4096, 4096), np.complex64), ((4096, 2048), np.complex128), ((256, 4096), np.complex64),
                                         ((512, 256), np.complex128), ((64, 16), np.complex64), ((2048, 8192), np.complex64)])
@pytest.mark.parametrize('log_k', [-1, 0, 3, 7])
def test_tiled_intermediate_layouts_vs_numpy(pa, shape, dtype, log_k):
    """focus / unfocus with the layout tile of the intermediate from one column tile (log_k 0: narrower than every thread group) to
    1024 columns (log_k 7: wider than the 128 .. 512 threads of a row): the per-thread offset + per-slot uniform base of round 5 must
    land every element where the column pass reads it"""
    from prysm_amd import _lib
    P = pa.propagation
    rng = np.random.default_rng(shape[0] + shape[1] + log_k)
    x = crandn(rng, shape, dtype)
    tol = TOL64 if dtype == np.complex128 else TOL32
    with _lib.tuning_local(log_k=log_k):
        assert rel_max(tonp(P.focus(x, 1)), O.focus(x.astype(np.complex128), 1)) < tol
        assert rel_max(tonp(P.unfocus(x, 1)), O.unfocus(x.astype(np.complex128), 1)) < tol
        if shape[0] >= 256:     # the padded (unfolded, zero rows skipped) form
            small = x[:shape[0] // 2, :shape[1] // 2]
            assert rel_max(tonp(P.focus(small, 2)), O.focus(small.astype(np.complex128), 2)) < tol


@pytest.mark.parametrize('shape,dtype', [((4096, 4096), np.complex128), ((2048, 4096), np.complex128), ((4096, 2048), np.complex128)])
def test_fused_chain_layouts(pa, shape, dtype):
    """the folded 3-pass chain (fold store, plane column passes, unfold load -- all three on the lean addressing) at three layouts: the
    oracle's numbers, and the same bits whatever the layout"""
    from prysm_amd import _lib
    P = pa.propagation
    rng = np.random.default_rng(sum(shape))
    x = crandn(rng, shape, dtype)
    want = O.angular_spectrum(x, O.HeNe, 0.01, 25.0, Q=1)
    got = {}
    for log_k in (-1, 0, 5):
        with _lib.tuning_local(log_k=log_k):
            got[log_k] = P.angular_spectrum(x, O.HeNe, 0.01, 25.0, Q=1)
            f = P.focus(x, 1)
        assert rel_max(tonp(got[log_k]), want) < TOL64
        assert rel_max(tonp(f), O.focus(x, 1)) < TOL64
    assert all(torch.equal(got[-1], g) for g in got.values())


def test_synthesis_falls_back_when_the_planner_refuses(pa):
    """_ops.synth_supported restates the planner's test without its inputs; with mix = 0 in a tuning_local block a 1000-wide
    lazy pupil is not a row length whose kernel synthesises -- the call used to raise, now the pupil is materialised"""
    from prysm_amd import _lib
    P = pa.propagation
    rng = np.random.default_rng(12)
    amp = (rng.random((300, 1000)) > 0.3).astype(np.float32)
    opd = (50 * rng.standard_normal((300, 1000))).astype(np.float32)
    want = O.focus(O.from_amp_and_phase(amp.astype(np.float64), opd.astype(np.float64), 0.6328), 1)
    wf = P.Wavefront.from_amp_and_phase(amp, opd, 0.6328, 0.04)
    assert rel_max(tonp(wf.focus(100.0, Q=1).data), want) < 2e-5          # mixed-radix rows synthesise
    with _lib.tuning_local(mix=0):
        wf2 = P.Wavefront.from_amp_and_phase(amp, opd, 0.6328, 0.04)
        assert rel_max(tonp(wf2.focus(100.0, Q=1).data), want) < 2e-5     # Bluestein rows do not: materialised
        assert rel_max(tonp(wf2.focus_intensity(100.0, Q=1).data), O.intensity(want)) < 4e-5


def test_tuning_local_blocks_nest(pa):
    """the inner block's exit used to discard the outer block's knobs"""
    from prysm_amd import _lib
    lib = _lib.load()

    def route(n):
        d = _lib.pm_fft2_desc()
        d.dtype, d.direction = _lib.PM_C64, -1
        d.in_y = d.in_x = d.out_y = d.out_x = _lib.pm_axis(n, n, 0, 0)
        d.in_ld = d.out_ld = n
        buf = ctypes.create_string_buffer(256)
        _lib.check(lib.pm_plan_explain(ctypes.byref(d), 0, buf, 256))
        return buf.value.decode()
    assert 'mixed-radix' in route(1000) and 'engine-fold' in route(4096)
    with _lib.tuning_local(mix=0):
        assert 'mixed-radix' not in route(1000)
        with _lib.tuning_local(fold=0):
            assert 'mixed-radix' not in route(1000) and 'engine-fold' not in route(4096)
        assert 'mixed-radix' not in route(1000) and 'engine-fold' in route(4096)      # the outer block survives the inner exit
        with pytest.raises(NotImplementedError):
            with _lib.tuning_local(fold=1, spectral2=3):       # refused by the product build: nothing of the block may stick
                pass
        assert 'mixed-radix' not in route(1000) and 'engine-fold' in route(4096)
    assert 'mixed-radix' in route(1000)
