"""GPU parity tests added in round 3 (VERDICT r2 "Next round" items 2, 3 and the kernels written for them):

* the middle pass of the fused fft2 -> x H -> ifft2 chain in its three forms (tuning key colmul_mode: one tile per workgroup, the
  same under a 128-register cap, persistent prefetching workgroups with twiddles / hy in LDS) against the fp64 oracle at BASELINE
  config 3's size and below, both precisions, separable and full (tf=) multipliers;
* the gradient path AT SIZE against the oracle: focus_adjoint 4096^2 (Q = 2 crop and Q = 1), angular_spectrum_adjoint 4096^2
  complex128, MDFT.adjoint 512^2 -> 2048^2 complex64, intensity_adjoint (pm_rmul);
* adjoint dot-product identities on the same operators at size.

Tolerances as in test_gpu_parity.py: max error / max magnitude against the fp64 oracle, 1e-10 (complex128), 5e-6 (complex64
transforms), 3e-5 (K = 2048 matrix DFT in fp32).
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


# ----------------------------------------------------------------------------- fused chain: the three middle-pass forms

@pytest.mark.parametrize('mode', [0, 1, 2])
@pytest.mark.parametrize('n,dtype,tol', [(4096, np.complex128, TOL64), (4096, np.complex64, TOL32), (2048, np.complex128, TOL64),
                                         (2048, np.complex64, TOL32)])
def test_angular_spectrum_middle_pass_forms(pa, mode, n, dtype, tol):
    """angular_spectrum(x, Q = 1) -- config 3 at 4096^2 complex128 -- with the middle pass in each of its forms; the result must
    not depend on the form beyond rounding, and all of them match the oracle"""
    from prysm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(n + mode)
    x = crandn(rng, (n, n), dtype)
    prec = pa.config.precision
    pa.config.precision = 32 if dtype == np.complex64 else 64
    lib.pm_set_tuning(b'colmul_mode', mode)
    try:
        got = tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1))
    finally:
        lib.pm_set_tuning(b'colmul_mode', 2)
        pa.config.precision = prec
    ref = O.angular_spectrum(x.astype(np.complex128), O.HeNe, 0.01, 10.0, Q=1)
    assert got.dtype == dtype
    assert rel_max(got, ref) < tol


@pytest.mark.parametrize('mode', [0, 1, 2])
def test_angular_spectrum_tf_and_adjoint_middle_pass_forms(pa, mode):
    """tf= (a full multiplier: the persistent form declines it and the call must still be right) and the adjoint (conj H) at 4096^2"""
    from prysm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77 + mode)
    x = crandn(rng, (4096, 4096))
    tf = O.angular_spectrum_transfer_function((4096, 4096), O.HeNe, 0.01, 10.0)
    lib.pm_set_tuning(b'colmul_mode', mode)
    try:
        got_tf = tonp(pa.propagation.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1, tf=tf))
        got_adj = tonp(pa.propagation.angular_spectrum_adjoint(x, O.HeNe, 0.01, 10.0, Q=1))
    finally:
        lib.pm_set_tuning(b'colmul_mode', 2)
    assert rel_max(got_tf, O.angular_spectrum(x, O.HeNe, 0.01, 10.0, Q=1, tf=tf)) < TOL64
    assert rel_max(got_adj, O.angular_spectrum_adjoint(x, O.HeNe, 0.01, 10.0, Q=1)) < TOL64


# ----------------------------------------------------------------------------- the gradient path at size

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


# ----------------------------------------------------------------------------- GEMM: K split inside the workgroup, |.|^2 epilogue

def _op_np(a, op):
    if op & 1:
        a = np.conj(a)
    if op & 2:
        a = a.T
    return a


@pytest.mark.parametrize('M,N,K', [(512, 2048, 64), (1024, 1024, 96), (512, 512, 64), (512, 1024, 192), (768, 1024, 128),
                                   (512, 512, 2048), (512, 2048, 2048)])
def test_cgemm_in_workgroup_k_split_all_ops(pa, M, N, K):
    """the 64 x 32 (two K-groups) and 32 x 32 (four K-groups) forms of the LDS-DMA kernel -- what the matrix-DFT products of
    config 4 now run on -- for every transposed / conjugated operand storage, against numpy in fp64; gemm_wk = 0 (round 2's
    split-K slabs) must agree to rounding"""
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
        for form in (0, 2, 7):        # round 2's slabs; the 64 x 32 form; every in-workgroup form
            lib.pm_set_tuning(b'gemm_wk', form)
            try:
                old = tonp(_ops.cgemm(At, Bt, opA, opB, alpha=0.5))
            finally:
                lib.pm_set_tuning(b'gemm_wk', 5)
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
