"""CPU: host-side logic that needs no GPU -- config, shape algebra, sharding, the FFT engine emulation."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_precision():
    """reference tests/config/test_config.py:29-50."""
    from prysm_amd.conf import Config
    c = Config()
    assert c.precision is np.float64 and c.precision_complex is np.complex128
    for p in (32, 'float32', np.float32, np.dtype('float32')):
        c.precision = p
        assert c.precision is np.float32 and c.precision_complex is np.complex64
    c.precision = 64
    assert c.precision_complex is np.complex128 and c.compute_precision is np.float64
    # 16 is accepted as the reference accepts it (test_config.py:29-33: float16 / complex64); the device synthesises in float32
    for p in (16, np.int64(16), 'float16', np.float16):
        c.precision = p
        assert c.precision is np.float16 and c.precision_complex is np.complex64 and c.compute_precision is np.float32
    c.precision = np.int64(32)
    assert c.precision is np.float32 and c.compute_precision is np.float32
    for bad in (8, 1, 'int32', 'int16', np.int32, np.complex64, 'nope', True):
        with pytest.raises(ValueError):
            c.precision = bad


def test_shape_algebra_matches_the_reference_convention():
    """(9,12) * 1.5 -> (14,18) -> back to (9,12)."""
    from prysm_amd.propagation._kernels import _padded_shape, _shape_before_pad
    assert _padded_shape((9, 12), 1.5) == (14, 18)
    assert _shape_before_pad((14, 18), 1.5) == (9, 12)
    assert _padded_shape((7, 9), 1) == (7, 9)
    assert _padded_shape((2048, 2048), 2) == (4096, 4096)


def test_sampling_helpers():
    from prysm_amd import propagation as P
    for dzeta in (1 / 128.0, 1 / 256.0, 11.123 / 128.0, 1e10 / 2048.0):
        psf_sample = P.pupil_sample_to_psf_sample(dzeta, 128, 0.55, 10)
        assert P.psf_sample_to_pupil_sample(psf_sample, 128, 0.55, 10) == dzeta
    assert P.Q_for_sampling(10, 100, 0.5, 2.5) == pytest.approx(2.0)
    assert P.unit_cell_focal_grid(0.1, 10, 0.5, 100, Q=2) == (0.5 * 100 / 0.1 / 200, 200)


def test_shard_bounds_cover_everything_once():
    from prysm_amd.polychromatic import shard_bounds
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                lo, hi = shard_bounds(n, r, w)
                seen += list(range(lo, hi))
            assert seen == list(range(n))
    assert shard_bounds(64, 3, 8) == (24, 32)


def test_fft_engine_emulation():
    """Compile and run the CPU emulation of the HIP FFT kernels' per-thread logic (tools/emu_fft.cpp):
    same headers as the device build, every stage structure, pad / shift / crop maps, tiled intermediate."""
    exe = '/tmp/pm_emu_fft_test'
    subprocess.run(['g++', '-O2', '-std=c++17', '-I', os.path.join(ROOT, 'prysm_amd', 'csrc'),
                    os.path.join(ROOT, 'tools', 'emu_fft.cpp'), '-o', exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert 'EMU OK' in out.stdout


def test_mixed_radix_kernel_emulation():
    """tools/emu_mix.cpp: the per-thread code of the mixed-radix kernel (csrc/fft_mixed.h: planner, every small DFT from 2 to 32, first /
    middle / last stages, digit-reversed last stage, row and column layouts) run thread by thread against a long-double DFT"""
    exe = '/tmp/pm_emu_mix_test'
    subprocess.run(['g++', '-O2', '-std=c++17', '-I', os.path.join(ROOT, 'prysm_amd', 'csrc'),
                    os.path.join(ROOT, 'tools', 'emu_mix.cpp'), '-o', exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:]
    assert 'all ok' in out.stdout


def test_composite_register_engine_emulation():
    """tools/emu_ce.cpp: the per-thread code of the composite register engine (csrc/fft_ce.h: prime-factor small DFTs, stages, the
    register-index exchange, row and column global access with windows and rotations) for every shipped plan and workgroup shape
    (tools/emu_ce_plans.inc, written by tools/ce_gen.py) run thread by thread against a long-double DFT"""
    exe = '/tmp/pm_emu_ce_test'
    subprocess.run(['g++', '-O2', '-std=c++17', '-I', os.path.join(ROOT, 'prysm_amd', 'csrc'), '-I', os.path.join(ROOT, 'tools'),
                    os.path.join(ROOT, 'tools', 'emu_ce.cpp'), '-o', exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:]
    assert 'all ok' in out.stdout


def test_generated_plan_tables_match_the_generator():
    """The composite register engine's instantiation files are generated (tools/ce_gen.py): the committed csrc/fft_ce_*.hip and
    tools/emu_ce_plans.inc are what the table produces, every plan satisfies the engine's constraint (later factors divide the first),
    fits the LDS and at most kCeMaxSeqs sequences per workgroup, and the first exchange of every shipped shape is free of bank conflicts
    under the model of tools/ce_banks.py."""
    import importlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        gen = importlib.import_module('ce_gen')
        banks = importlib.import_module('ce_banks')
    finally:
        sys.path.pop(0)
    csrc = os.path.join(ROOT, 'prysm_amd', 'csrc')
    for prec in gen.TABLE:
        for col, mid, name in ((False, False, 'rows'), (True, False, 'cols'), (True, True, 'mid')):
            want = gen.emit(prec, col, False, mid=mid)
            assert open(os.path.join(csrc, 'fft_ce_%s_%s.hip' % (name, prec))).read() == want, (prec, name)
        for n, (plan, rs, cs, _, _) in gen.TABLE[prec].items():
            prod = 1
            for r in plan:
                prod *= r
                assert plan[0] % r == 0 and 2 <= r <= 48
            assert prod == n and len(plan) <= 4
    inc = open(os.path.join(ROOT, 'tools', 'emu_ce_plans.inc')).read()
    assert inc.count('CHECK(') == 2 * sum(len(t) for t in gen.TABLE.values())
    for plan, seqs, col, es in (((30, 10, 10), 1, False, 4), ((30, 10, 10), 4, True, 4), ((24, 8, 8), 1, False, 8), ((20, 20, 20), 2, True, 8)):
        (best, pad), _ = banks.pads(list(plan), seqs, col, es)[0]
        assert best <= 1.0 + 1e-9, (plan, seqs, col, es, best, pad)


REF = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'prysm')), reason='reference not present on this box')
def test_install_into_prysm_rebinds_and_restores():
    """The plug mechanism of INTEGRATION.md: names on prysm's modules are rebound to the MI355X engine and
    restored afterwards (no compute here -- there is no GPU in this container)."""
    import sys
    sys.path.insert(0, REF)
    try:
        import prysm
        import prysm.propagation as P
        import prysm.propagation.wavefront as W
        import prysm.fttools as F
    finally:
        sys.path.remove(REF)
    import prysm_amd
    from prysm_amd import mathops
    orig_focus, orig_mdft, orig_wf = P.focus, F.MDFT, P.Wavefront
    mathops.set_backend_to_mi355x(prysm)
    try:
        assert P.focus is prysm_amd.propagation.focus
        assert W.focus is prysm_amd.propagation.focus          # bound by name in wavefront.py
        assert P.prepare_executor is prysm_amd.propagation.prepare_executor
        assert F.MDFT is prysm_amd.fttools.MDFT
        assert P.Wavefront is prysm_amd.propagation.Wavefront
        import prysm.mathops as PM
        assert isinstance(PM.fft._srcmodule, mathops.FFTFacade)    # the unfused fft shim level
        assert PM.fft.fft2.__self__ is PM.fft._srcmodule
    finally:
        mathops.restore_prysm_backend()
    assert P.focus is orig_focus and F.MDFT is orig_mdft and P.Wavefront is orig_wf
    import prysm.mathops as PM
    assert not isinstance(PM.fft._srcmodule, mathops.FFTFacade)


def test_small_host_helpers():
    """Pure host arithmetic of the newer host code: norm scales of the fft facade, stack sizing of the polychromatic
    driver, the auto choice between stacks and field-by-field."""
    from prysm_amd.mathops import FFTFacade
    from prysm_amd import polychromatic as pc
    sc = FFTFacade._scale
    assert sc(None, 64, False) == 1.0 and sc(None, 64, True) == 1 / 64
    assert sc('ortho', 64, False) == sc('ortho', 64, True) == 0.125
    assert sc('forward', 64, False) == 1 / 64 and sc('forward', 64, True) == 1.0
    with pytest.raises(ValueError):
        sc('nope', 4, False)
    assert pc._fields_per_launch((512, 512), 8, 2) == 64                 # capped
    assert pc._fields_per_launch((4096, 4096), 16, 1) == 2               # 1 GiB of stack / (256 MiB + 128 MiB)
    assert pc.shard_bounds(64, 3, 8) == (24, 32) and pc.shard_bounds(7, 1, 2) == (4, 7)


def test_array_to_true_numpy_unwraps_containers():
    """Wavefront / RichData convert through their array (never as a 0-d object array); unknown objects are refused."""
    import torch
    from prysm_amd.mathops import array_to_true_numpy
    from prysm_amd._richdata import RichData
    t = torch.arange(6, dtype=torch.float64).reshape(2, 3)
    rd = RichData(t, 1.0, 0.5)
    out = array_to_true_numpy(rd)
    assert isinstance(out, np.ndarray) and out.dtype == np.float64 and out.shape == (2, 3)
    assert array_to_true_numpy(3.0) == 3.0 and array_to_true_numpy([1, 2]).tolist() == [1, 2]
    with pytest.raises(TypeError):
        array_to_true_numpy(object())


def test_numpy_facade_semantics():
    """NumpyFacade / DeviceArray: numpy spellings and dtype rules on torch tensors (CPU tensors here: plumbing only)."""
    import torch
    from prysm_amd.npfacade import NumpyFacade, DeviceArray
    xp = NumpyFacade(torch.device('cpu'))
    a = xp.arange(-3, 4, dtype=np.float32)
    assert isinstance(a, DeviceArray) and a.dtype == torch.float32 and a.get().dtype == np.float32
    assert xp.arange(5).dtype == torch.int64 and xp.zeros((2, 3)).dtype == torch.float64
    assert xp.zeros(4, dtype=np.complex64).dtype == torch.complex64
    assert a.astype(np.float64).dtype == torch.float64 and a.astype(xp.float64).get().dtype == np.float64
    z = xp.exp(1j * xp.pi * a.astype(np.float64))
    np.testing.assert_allclose(z.get(), np.exp(1j * np.pi * np.arange(-3, 4.0)), atol=1e-15)
    assert isinstance(xp.exp(1.0), float) or isinstance(xp.exp(1.0), np.floating)        # host scalars stay on the host
    p = xp.pad(xp.ones((2, 3)), ((1, 2), (3, 0)))
    np.testing.assert_array_equal(p.get(), np.pad(np.ones((2, 3)), ((1, 2), (3, 0))))
    gx, gy = xp.meshgrid(xp.arange(3), xp.arange(2))
    nx, ny = np.meshgrid(np.arange(3), np.arange(2))
    np.testing.assert_array_equal(gx.get(), nx)
    np.testing.assert_array_equal(gy.get(), ny)
    w = xp.where(a > 0, 1j, 0.5)
    np.testing.assert_array_equal(w.get(), np.where(np.arange(-3, 4.0) > 0, 1j, 0.5))
    m = np.arange(6.0).reshape(2, 3) * xp.ones((2, 3))            # numpy operand defers to the device array
    assert isinstance(m, DeviceArray)
    np.testing.assert_array_equal(np.asarray(m), np.arange(6.0).reshape(2, 3))
    assert xp.result_type(a, np.complex128) == np.complex128 and xp.finfo(a.dtype).eps == np.finfo(np.float32).eps
    assert xp.iscomplexobj(z) and not xp.iscomplexobj(a) and xp.sum(a > 0).item() == 3
    np.testing.assert_array_equal(xp.real(z).get(), z.get().real)
    with pytest.raises(AttributeError):
        xp.polyfit
    with pytest.raises(RuntimeError):
        NumpyFacade().zeros(3) if not torch.cuda.is_available() else (_ for _ in ()).throw(RuntimeError())


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'prysm')), reason='reference not present on this box')
def test_numpy_facade_runs_the_references_array_code():
    """prysm's own coordinates / geometry / Wavefront constructors / pad2d / transfer function / MDFT executor run
    unmodified on ``np._srcmodule = NumpyFacade`` and give numpy's results, dtypes included."""
    import sys
    import torch
    sys.path.insert(0, REF)
    try:
        from prysm import mathops, coordinates, geometry, fttools, propagation
        from prysm.conf import config
        from prysm.propagation import Wavefront
    finally:
        sys.path.remove(REF)
    from prysm_amd.npfacade import NumpyFacade, DeviceArray
    from prysm_amd.mathops import FFTFacade
    field = np.random.default_rng(3).standard_normal((24, 20)) + 1j * np.random.default_rng(4).standard_normal((24, 20))

    def run():
        xp = mathops.np
        out = {}
        x, y = coordinates.make_xy_grid(64, diameter=10)
        r, t = coordinates.cart_to_polar(x, y)
        A = geometry.circle(5, r)
        wf = Wavefront.from_amp_and_phase(A, 100 * (r / 5) ** 2 * xp.cos(t), 0.6328, x[0, 1] - x[0, 0])
        pad = fttools.pad2d(wf.data, Q=2)
        out.update(x=x, r=r, t=t, A=A, wf=wf.data, pad=pad, crop=fttools.crop_center(pad, (40, 30)), I=wf.intensity.data,
                   rng=fttools.fftrange(9), lens=Wavefront.thin_lens(250.0, 0.6328, x, y).data)
        out['xx'], _, out['fx'], _ = propagation.coordinates_for_focus(0.1, (64, 64), 2.0, (16, 16), 0.6328, 100.0)
        for prec in (64, 32):
            config.precision = prec
            try:
                a = xp.asarray(field.astype(np.complex64 if prec == 32 else np.complex128))
                ex = propagation.prepare_executor(0.2, (24, 20), 3.0, (10, 12), 0.6328, 50.0, kind='mdft')
                out[f'mdft{prec}'] = propagation.focus_dft(a, ex)
                out[f'mdft_adj{prec}'] = propagation.focus_dft_adjoint(out[f'mdft{prec}'], ex)
                out[f'H{prec}'] = propagation.angular_spectrum_transfer_function((8, 12), 0.6328, 0.01, 10.0)
            finally:
                config.precision = 64
        return out

    ref = run()
    cpu = torch.device('cpu')
    mathops.np._srcmodule, mathops.fft._srcmodule = NumpyFacade(cpu), FFTFacade(cpu)
    try:
        got = run()
        with pytest.raises(RuntimeError):      # the transforms themselves have no CPU path
            propagation.focus(got['wf'], 2)
    finally:
        mathops.set_backend_to_defaults()
    for k, want in ref.items():
        assert isinstance(got[k], DeviceArray), k
        have = got[k].get()
        want = np.asarray(want)
        assert have.shape == want.shape and have.dtype == want.dtype, (k, have.dtype, want.dtype)
        tol = 1e-6 if want.dtype in (np.complex64, np.float32) else 1e-14
        assert np.abs(have.astype(complex) - want.astype(complex)).max() <= tol * max(np.abs(want).max(), 1), k


def test_next_fast_len_and_czt_length():
    """fast lengths are what the device transforms without Bluestein's detour: powers of two, and from 96 the lengths 3 / 5 / 7 x 2^k
    (prysm/fttools.py:23-31 asks the FFT backend); a chirp-Z axis stays on a power of two while one fused kernel holds it (K <= 8192)"""
    from prysm_amd.fttools import next_fast_len, _czt_len
    want = {1: 1, 2: 2, 17: 32, 95: 96, 97: 112, 100: 112, 2559: 2560, 3000: 3072, 4096: 4096, 5000: 5120, 7000: 7168, 8703: 10240,
            9000: 10240, 20000: 20480, 30000: 32768}
    for n, v in want.items():
        assert next_fast_len(n) == v, n
        assert next_fast_len(n) >= n
    for n in range(1, 3000, 7):
        v = next_fast_len(n)
        r = v
        while r % 2 == 0:
            r //= 2
        assert v >= n and r in (1, 3, 5, 7) and (r == 1 or v >= 96)
    assert _czt_len(2559) == 4096 and _czt_len(4607) == 8192 and _czt_len(8703) == 10240 and _czt_len(11999) == 12288


@pytest.mark.parametrize('mode', ['mean', 'maximum', 'minimum', 'median', 'linear_ramp'])
def test_pad_statistical_modes_on_host_tensors(mode):
    """fttools._pad_stat is tensor reductions + concatenations: the same code on CPU tensors against np.pad (the GPU suite runs it on
    the device)"""
    import torch
    from prysm_amd.fttools import _pad_stat
    rng = np.random.default_rng(4)
    for shape, out_shape in (((9, 12), (14, 18)), ((5, 4), (17, 21)), ((1, 6), (4, 6)), ((6, 7), (6, 12))):
        for dt in (np.float32, np.float64, np.complex128):
            if dt is np.complex128 and mode not in ('mean', 'linear_ramp'):
                with pytest.raises(TypeError):
                    _pad_stat(torch.zeros(shape, dtype=torch.complex128), [(1, 1), (1, 1)], mode)
                continue
            a = rng.standard_normal(shape).astype(dt)
            if dt is np.complex128:
                a = a + 1j * rng.standard_normal(shape)
            widths = [(o - i - (o - i) // 2, (o - i) // 2) for o, i in zip(out_shape, shape)]
            want = np.pad(a, widths, mode=mode)
            got = _pad_stat(torch.from_numpy(a), widths, mode).numpy()
            assert got.dtype == want.dtype and got.shape == want.shape
            assert np.allclose(got, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('mode', ['mean', 'median', 'linear_ramp', 'maximum'])
def test_pad_statistical_modes_round_integer_arrays_like_numpy(mode):
    """np.pad ROUNDS the statistic / ramp of an integer array (half to even) before casting back; truncation was off by
    one.  Also the median of arrays beyond torch.quantile's 16M-element limit (by sort)."""
    import torch
    from prysm_amd.fttools import _pad_stat
    rng = np.random.default_rng(11)
    for shape, widths in (((7, 9), [(2, 3), (4, 1)]), ((4, 6), [(5, 0), (0, 7)])):
        a = rng.integers(-50, 50, size=shape).astype(np.int64)
        want = np.pad(a, widths, mode=mode)
        got = _pad_stat(torch.from_numpy(a), widths, mode).numpy()
        assert got.dtype == want.dtype and np.array_equal(got, want), (mode, shape)
    if mode == 'median':
        big = torch.from_numpy(rng.standard_normal((4100, 4100)).astype(np.float32))       # 16.8M elements
        got = _pad_stat(big, [(1, 0), (0, 1)], 'median')
        want = np.pad(big.numpy(), [(1, 0), (0, 1)], mode='median')
        assert np.allclose(got.numpy()[0, :5], want[0, :5]) and np.allclose(got.numpy()[:5, -1], want[:5, -1])


def test_hermitian_path_predicate():
    """otf._hermitian_ok mirrors the library's eligibility test (capi.hip r2c_legal) so ineligible PSFs never pay a failed call"""
    import torch
    from prysm_amd.otf import _hermitian_ok
    assert _hermitian_ok(torch.zeros(64, 128))
    assert _hermitian_ok(torch.zeros(4096, 4096, dtype=torch.float64)[:32, :32].contiguous())
    assert not _hermitian_ok(torch.zeros(1000, 1000))                       # not powers of two
    assert not _hermitian_ok(torch.zeros(16, 64))                           # too short
    assert not _hermitian_ok(torch.zeros(64, 64, dtype=torch.complex64))    # complex PSF
    assert not _hermitian_ok(torch.zeros(64, 8192, dtype=torch.float64))    # complex128 rows of 4096 complex points
    big = torch.zeros(64, 130)
    assert not _hermitian_ok(big[:, 1:65])                                  # base address one float off a complex boundary
    assert _hermitian_ok(big[:, 2:66])
    assert not _hermitian_ok(torch.zeros(64, 131)[:, :64])                  # odd leading dimension


def test_untangle_formula_of_the_real_input_route():
    """The algebra pm_r2c_untangle implements (csrc/pointwise.hip), restated in numpy: the 2-D transform of a real M x N array from
    the transform of its M x N/2 complex-pair view -- pairs of packed bins (u, k) <-> (-u, N/2 - k), the Nyquist column from k = 0,
    the other half by conjugate symmetry, the input rotation as a phase -- against numpy's fft2, on even / odd heights and widths whose
    halves are even and odd"""
    rng = np.random.default_rng(3)
    for M, N in ((8, 16), (7, 10), (5, 6), (1, 2), (12, 4), (9, 30)):
        x = rng.random((M, N))
        n2 = N // 2
        zf = np.fft.fft2(x[:, 0::2] + 1j * x[:, 1::2])
        F = np.zeros((M, N), dtype=np.complex128)
        wN = np.exp(-2j * np.pi * np.arange(N) / N)
        for u in range(M):
            um = (M - u) % M
            for k in range(n2 // 2 + 1):
                kb = (n2 - k) % n2
                a, b = zf[u, k], zf[um, kb]

                def bin_(A, Bc, w):
                    return (A + Bc) / 2 + w * (A - Bc) / 2j
                f1 = bin_(a, np.conj(b), wN[k])
                F[u, k] = f1
                if k != 0:
                    F[um, N - k] = np.conj(f1)
                if k == 0:
                    F[u, n2] = bin_(a, np.conj(b), -1.0)
                elif kb != k:
                    f2 = bin_(b, np.conj(a), -np.conj(wN[k]))
                    F[um, kb] = f2
                    F[u, N - kb] = np.conj(f2)
        assert np.allclose(F, np.fft.fft2(x), atol=1e-12 * N * M)
        # rotation of the input by (sy, sx) samples = the phase conj(W_M^(u sy) W_N^(k sx))
        sy, sx = M // 2, 3 % N
        ph = np.exp(2j * np.pi * (np.arange(M)[:, None] * sy / M + np.arange(N)[None, :] * sx / N))
        assert np.allclose(F * ph, np.fft.fft2(np.roll(x, (-sy, -sx), axis=(0, 1))), atol=1e-12 * N * M)


def test_bench_run_in_rule_and_device_state_without_a_gpu():
    """bench.py's run-in stops on NO DRIFT, not on a plateau of the clock ramp (profiles/r06/exp_warm_trajectory.log: three equal 4 ms
    batches at 120 ms were not the steady state); gpu_state() never raises; the summary carries the round-6 keys."""
    import bench
    dt = 4.0                                                       # ms per batch
    at = [dt * (i + 1) for i in range(200)]
    ramp = [98.0 - 4.0 * min(1.0, a / 200.0) for a in at]          # 98 us per step falling to 94 over 200 ms, flat afterwards
    secs = [v * 40e-6 for v in ramp]
    stops = [i for i in range(len(at)) if bench.run_in_is_steady(secs[:i + 1], at[:i + 1])]
    assert stops and at[stops[0]] >= 250.0                         # not while the look-back window (100 ms earlier) is still more than 1 % up the ramp
    # a plateau in the middle of the ramp (25 batches = 100 ms at one level, then the fall continues): never "steady" while it lasts less
    # than look-back + five batches
    stair = [98.0] * 20 + [96.0] * 20 + [94.0] * 160
    secs2 = [v * 40e-6 for v in stair]
    first = next(i for i in range(len(at)) if bench.run_in_is_steady(secs2[:i + 1], at[:i + 1]))
    assert stair[first] == 94.0 and at[first] >= 160.0 + 100.0
    # noise of +-0.3 % does not keep it from stopping; a 2 % sawtooth does
    rng = np.random.default_rng(0)
    quiet = [94.0 * (1 + 0.003 * rng.standard_normal()) * 40e-6 for _ in at]
    assert any(bench.run_in_is_steady(quiet[:i + 1], at[:i + 1]) for i in range(len(at)))
    saw = [94.0 * (1.05 if i % 2 else 1.0) * 40e-6 for i in range(len(at))]
    assert not any(bench.run_in_is_steady(saw[:i + 1], at[:i + 1]) for i in range(len(at)))
    st = bench.gpu_state(0)
    assert isinstance(st, dict)                                    # sysfs of another machine, or none at all: a dict either way
    line = {'value': 1.0, 'ms_per_step': 2.0, 'repeat_ms': [2.0, 2.1], 'repeat_spread': 0.05, 'prewarm_ms': 400.0,
            'gpu_before': {'sclk_mhz': 2350.0}, 'gpu_after': {'sclk_mhz': 2340.0, 'power_w': 1200.0},
            'roofline': {'frac': 0.7, 'row_pass_ms': 0.047, 'column_pass_ms': 0.046, 'two_plain_copies_ms': 0.087},
            'other_configs': {'model_7plane_1024': {'c64': {'eager_ms_per_wavelength': 0.127, 'graph_ms_per_wavelength': 0.135}},
                              'config4_mdft_2048_to_512_c64': {'ms': 0.141, 'frac_of_f32_mfma_peak': 0.96, 'mfma_busy': 0.68}}}
    summ = bench._summary(line)
    for key, want in (('repeat_ms', [2.0, 2.1]), ('sclk_before', 2350.0), ('power_w_after', 1200.0), ('m7_c64_eager_ms', 0.127), ('c4_mfma_busy', 0.68)):
        assert summ[key] == want, key
    assert bench.model7_flops()['total'] == 3 * 8.0 * 256 * 1024 * (1024 + 256)
