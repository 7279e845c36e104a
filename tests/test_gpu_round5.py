"""GPU parity tests added in round 5:

* the lean addressing of the tiled intermediate (fft_io.h TiledRowAddr) on every layout: layout tiles narrower AND wider than a
  thread group (knob log_k), folded and unfolded, both precisions, the real-input row pass -- against numpy fp64;
* prysm_amd.graph.sequence(): an unmodified loop of Wavefront code on alternating streams, bit-equal to the one-stream run and
  equal to the oracle; StreamRing batches that re-use inputs (results recorded on the caller's stream at join);
* pupil synthesis in the load with a planner that says no (tuning_local(mix=0)): the pupil is materialised, nothing raises;
* nested tuning_local blocks; the packed-pupil cache is per stream;
* the three root-only reduce forms on a process group of one rank (RCCL).

Tolerances as in test_gpu_parity.py (max error / max magnitude against fp64): 1e-10 complex128, 5e-6 complex64 transforms.
"""
import ctypes

import numpy as np
import pytest
import torch

from conftest import rel_max
from oracle import prysm_oracle as O

pytestmark = pytest.mark.gpu

TOL64 = 1e-10
TOL32 = 5e-6


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


# ----------------------------------------------------------------------------- lean addressing of the tiled intermediate

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


# ----------------------------------------------------------------------------- sequences of independent propagations

def _seven_planes(P, amp, opd, wvl):
    """a small relay written as plain Wavefront code: pupil -> focus -> stop -> back -> free space -> focus -> intensity"""
    wf = P.Wavefront.from_amp_and_phase(amp, opd, wvl, 0.04)
    psf = wf.focus(100.0, Q=1)
    back = psf.unfocus(100.0, Q=1)
    back = back * P.Wavefront(amp.to(back.data.dtype), wvl, back.dx)
    moved = back.free_space(dz=5.0, Q=1)
    return moved.focus(100.0, Q=1).intensity.data


@pytest.mark.parametrize('n,rdt', [(512, torch.float32), (1000, torch.float32), (256, torch.float64)])
def test_sequence_block_matches_one_stream(pa, n, rdt):
    """graph.sequence(): a loop over wavelengths of a seven-plane chain written as plain Wavefront code -- independent chains alternate
    between two streams, the dependent steps of one chain follow their producer; the block's results are, bit for bit, the
    one-stream results, the oracle's numbers, and they are safe to consume on the caller's stream after the block"""
    from prysm_amd import graph
    P = pa.propagation
    g = torch.Generator(device='cuda').manual_seed(n)
    amp = (torch.rand((n, n), device='cuda', generator=g) > 0.3).to(rdt)
    opd = torch.randn((n, n), device='cuda', generator=g, dtype=rdt) * 30
    wvls = [0.5 + 0.03 * i for i in range(6)]
    want = [_seven_planes(P, amp, opd, w).clone() for w in wvls]
    torch.cuda.synchronize()
    with graph.sequence() as seq:
        outs = [_seven_planes(P, amp, opd, w) for w in wvls]
        streams_used = {id(s) for s in seq._producer.values()}
    total = sum(o.sum() for o in outs)          # consumed on the caller's stream right after the block
    torch.cuda.synchronize()
    assert len(streams_used) == 2
    assert all(torch.equal(o, w) for o, w in zip(outs, want)) and float(total) > 0
    # and the chain itself against the oracle (first wavelength)
    a, o = amp.double().cpu().numpy(), opd.double().cpu().numpy()
    f = O.focus(O.from_amp_and_phase(a, o, wvls[0]), 1)
    b = O.unfocus(f, 1) * a
    m = O.angular_spectrum(b, wvls[0], 0.04, 5.0, Q=1)
    ref = O.intensity(O.focus(m, 1))
    assert rel_max(tonp(outs[0]), ref) < (4e-5 if rdt == torch.float32 else 1e-9)
    assert graph.active_sequence() is None
    with pytest.raises(RuntimeError):
        with graph.sequence():
            with graph.sequence():
                pass
    assert graph.active_sequence() is None


def test_sequence_accumulator_and_host_conversion(pa):
    """an `out=` accumulator makes consecutive calls dependent (they follow the accumulator's stream), and a host conversion inside
    the block joins first"""
    from prysm_amd import graph
    P = pa.propagation
    rng = np.random.default_rng(5)
    fields = [torch.from_numpy(crandn(rng, (256, 256), np.complex64)).cuda() for _ in range(5)]
    acc = torch.zeros((256, 256), dtype=torch.float32, device='cuda')
    with graph.sequence() as seq:
        for f in fields:
            P.focus_intensity(f, 1, out=acc, weight=0.5)
        assert len({id(s) for s in seq._producer.values()}) == 1
        inside = tonp(acc)          # joins, then copies
    want = sum(0.5 * O.intensity(O.focus(tonp(f).astype(np.complex128), 1)) for f in fields)
    assert rel_max(inside, want) < 4e-5 and rel_max(tonp(acc), want) < 4e-5


def test_sequence_keeps_large_fields_on_one_stream(pa):
    """two propagations whose arrays cannot share the Infinity Cache (2048^2 complex128: 3 x 64 MB each) stay on one stream of the ring
    -- two streams measured 47 -> 54 us per call there --, smaller ones alternate; the same bits either way"""
    from prysm_amd import graph
    P = pa.propagation
    g = torch.Generator(device='cuda').manual_seed(3)
    big = [torch.randn((2048, 2048), device='cuda', generator=g, dtype=torch.float64).to(torch.complex128) for _ in range(2)]
    small = [torch.randn((1024, 1024), device='cuda', generator=g, dtype=torch.float32).to(torch.complex64) for _ in range(2)]
    want = [P.focus(x, 1).clone() for x in big + small]
    with graph.sequence() as seq:
        got_big = [P.focus(x, 1) for x in big]
        n_big = len({id(s) for s in seq._producer.values()})
        got_small = [P.focus(x, 1) for x in small]
        n_all = len({id(s) for s in seq._producer.values()})
    torch.cuda.synchronize()
    assert n_big == 1 and n_all == 2
    assert all(torch.equal(a, b) for a, b in zip(got_big + got_small, want))


def test_stream_ring_batches_reuse_inputs(pa):
    """'fork once; loop {run ...; join; consume; drop}' -- the first run of every batch forks and every result is recorded
    on the caller's stream, so a dropped result's block cannot be handed to the next batch while the caller's reads are queued"""
    from prysm_amd.graph import StreamRing
    rng = np.random.default_rng(9)
    fields = [torch.from_numpy(crandn(rng, (1024, 1024), np.complex64)).cuda() for _ in range(4)]
    want = [pa.propagation.focus(f, 1).abs().sum().item() for f in fields]
    ring = StreamRing(2)
    for trip in range(6):
        outs = [ring.run(pa.propagation.focus, f, 1) for f in fields]
        ring.join()
        assert not ring._forked
        sums = [o.abs().sum() for o in outs]        # queued on the caller's stream
        del outs                                     # the blocks go back to the allocator while those reads may still be queued
        got = [s.item() for s in sums]
        assert np.allclose(got, want, rtol=1e-6)


# ----------------------------------------------------------------------------- planner / host agreements

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


def test_packed_pupil_cache_is_per_stream(pa):
    from prysm_amd import polychromatic as pc
    amp = torch.ones((64, 64), device='cuda')
    opd = torch.randn((64, 64), device='cuda')
    pc.clear_packed_pupil_cache()
    a = pc.packed_pupil(amp, opd, amp, opd, cache=True)
    assert pc.packed_pupil(amp, opd, amp, opd, cache=True) is a
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        b = pc.packed_pupil(amp, opd, amp, opd, cache=True)
        assert b is not a and pc.packed_pupil(amp, opd, amp, opd, cache=True) is b
    torch.cuda.current_stream().wait_stream(s)
    assert torch.equal(a, b)
    pc.clear_packed_pupil_cache()


def test_root_only_reduce_forms_on_a_one_rank_group(pa):
    """'reduce', 'a2a' and 'rs' on RCCL with a process group of one rank: every collective of the 8-GPU path executes here
    (all_to_all_single, reduce_scatter_tensor, gather into views of the image), twice (the receive buffers are kept)"""
    import os
    import socket
    import torch.distributed as dist
    from prysm_amd.polychromatic import _reduce_image
    if dist.is_initialized():
        pytest.skip('a process group is already up in this process')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        img = torch.rand((512, 512), device='cuda')
        for method in ('reduce', 'a2a', 'rs'):
            for _ in range(2):
                acc = img.clone()
                out = _reduce_image(acc, 1, None, False, method=method, use_dist=True)
                torch.cuda.synchronize()
                assert out is acc and torch.equal(out, img)
    finally:
        dist.destroy_process_group()


# ----------------------------------------------------------------------------- real inputs on any even width

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


# ---------------------------------------------------------------------------
# composite register engine (csrc/fft_ce.h): every built plan against numpy fp64 and against the general mixed-radix kernel
# ---------------------------------------------------------------------------
CE_LENGTHS = [384, 500, 768, 900, 1000, 1152, 1280, 1500, 1536, 1600, 1800, 2000, 2304, 2500, 2560, 3000, 3072, 3600, 4000, 4500, 5000, 5120, 6000, 6144, 8000]


def _ce_ref(x, shape, in_off, in_shift, out_shift, direction):
    M, N = shape
    full = np.zeros((M, N), dtype=np.complex128)
    full[in_off[0]:in_off[0] + x.shape[0], in_off[1]:in_off[1] + x.shape[1]] = x
    full = np.roll(full, (-in_shift[0], -in_shift[1]), (0, 1))
    f = np.fft.fft2(full) if direction < 0 else np.fft.ifft2(full) * (M * N)
    return np.roll(f, out_shift, (0, 1))


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
