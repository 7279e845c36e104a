"""CPU, world_size 2, gloo: the N > 1 path of the polychromatic driver (sharding + one sum-reduce).

The per-wavelength propagation is injected, so here it is the CPU ORACLE on CPU tensors -- the
HIP path itself is exercised by the -m gpu tests; this test covers the distributed logic that the
driver's 8-GPU run relies on (contiguous wavelength blocks, weights travel with them, one reduce).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, reduce_to_all, q, method='reduce', sub=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import prysm_oracle as O
    from prysm_amd.polychromatic import incoherent_sum, shard_bounds
    n = 32
    x, y = O.make_xy_grid(n, diameter=10)
    r, _ = O.cart_to_polar(x, y)
    amp = O.circle(5, r)
    opd = O.hopkins_w040(r / 5, 300.0)
    wvls = np.linspace(0.5, 0.7, 7)          # 7 wavelengths over 2 ranks: uneven shards
    wts = np.linspace(1.0, 2.0, 7)
    seen = []

    def propagate(wvl, w, acc):
        seen.append(wvl)
        I = torch.from_numpy(O.intensity(O.focus(O.from_amp_and_phase(amp, opd, wvl), 2)) * w)
        return I if acc is None else acc.add_(I)

    group, grank, gworld = None, rank, world
    if sub:    # a sub-group that does NOT contain global rank 0: the reduce root is the group's first rank (global rank 1)
        group = dist.new_group([1, 2])
        if rank == 0:
            q.put((rank, None))
            dist.barrier()
            dist.destroy_process_group()
            return
        grank, gworld = rank - 1, 2
    out = incoherent_sum(propagate, wvls, wts, reduce_to_all=reduce_to_all, reduce_method=method, group=group)
    lo, hi = shard_bounds(7, grank, gworld)
    assert seen == [float(w) for w in wvls[lo:hi]]
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(reduce_to_all, method='reduce', world=2, sub=False):
    sys.path.insert(0, ROOT)
    from oracle import prysm_oracle as O
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, reduce_to_all, q, method, sub)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 32
    x, y = O.make_xy_grid(n, diameter=10)
    r, _ = O.cart_to_polar(x, y)
    amp = O.circle(5, r)
    opd = O.hopkins_w040(r / 5, 300.0)
    comps = [O.intensity(O.focus(O.from_amp_and_phase(amp, opd, w), 2)) for w in np.linspace(0.5, 0.7, 7)]
    ref = O.sum_of_2d_modes(np.asarray(comps), np.linspace(1.0, 2.0, 7))
    return res, ref


def test_incoherent_sum_all_reduce_world2():
    res, ref = _run(True)
    for rank in (0, 1):
        np.testing.assert_allclose(res[rank], ref, rtol=1e-12, atol=1e-12 * ref.max())


def test_incoherent_sum_reduce_to_root_world2():
    res, ref = _run(False)
    np.testing.assert_allclose(res[0], ref, rtol=1e-12, atol=1e-12 * ref.max())


def test_incoherent_sum_all_to_all_reduce_world2():
    """root-only result through the all-to-all of slices + ordered local sum + gather (the xGMI form of SURVEY 8e)"""
    res, ref = _run(False, method='a2a')
    np.testing.assert_allclose(res[0], ref, rtol=1e-12, atol=1e-12 * ref.max())


def test_incoherent_sum_reduce_scatter_world2_and_3():
    """root-only result through reduce_scatter_tensor + the gather into the slices of the root's image (round 5): the 64 x 64 image in two
    slices over 2 ranks; over 3 ranks it does not divide and the call must fall back to the plain reduce"""
    res, ref = _run(False, method='rs')
    np.testing.assert_allclose(res[0], ref, rtol=1e-12, atol=1e-12 * ref.max())
    res, ref = _run(False, method='rs', world=3)
    np.testing.assert_allclose(res[0], ref, rtol=1e-12, atol=1e-12 * ref.max())


def test_reduce_method_names_are_checked():
    sys.path.insert(0, ROOT)
    from prysm_amd.polychromatic import _reduce_image, REDUCE_METHODS
    assert REDUCE_METHODS == ('reduce', 'a2a', 'rs')
    with pytest.raises(ValueError):
        _reduce_image(torch.zeros(4, 4), 1, None, False, method='ring')


def test_incoherent_sum_subgroup_without_global_rank0():
    """dist.reduce's dst is a GLOBAL rank; a group [1, 2] must reduce to global rank 1"""
    for method in ('reduce', 'a2a', 'rs'):
        res, ref = _run(False, method=method, world=3, sub=True)
        np.testing.assert_allclose(res[1], ref, rtol=1e-12, atol=1e-12 * ref.max())


def _pipeline_worker(rank, world, port, q, method):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import prysm_oracle as O
    from prysm_amd.polychromatic import PsfPipeline
    n = 32
    x, y = O.make_xy_grid(n, diameter=10)
    r, _ = O.cart_to_polar(x, y)
    amp = O.circle(5, r)
    wvls, wts = np.linspace(0.5, 0.7, 5), np.linspace(1.0, 2.0, 5)

    def propagate(a, o, wvl, w, acc):
        I = torch.from_numpy(O.intensity(O.focus(O.from_amp_and_phase(a, o, wvl), 2)) * w)
        return I if acc is None else acc.add_(I)

    pipe = PsfPipeline(wvls, wts, 10.0 / n, 100.0, Q=2, reduce_method=method, propagate=propagate, depth=2)
    pend = [pipe.submit(amp, O.hopkins_w040(r / 5, 100.0 * (f + 1))) for f in range(3)]    # three frames, different OPDs
    pipe.drain()
    q.put((rank, [p.result().numpy() for p in pend]))
    dist.barrier()
    dist.destroy_process_group()


def test_psf_pipeline_frames_world2():
    """PsfPipeline: a sequence of frames, each frame's reduce issued behind its own wavelength loop (synchronous on CPU tensors):
    the root receives every frame's oracle sum, in order, for both root-only reduce forms"""
    sys.path.insert(0, ROOT)
    from oracle import prysm_oracle as O
    n = 32
    x, y = O.make_xy_grid(n, diameter=10)
    r, _ = O.cart_to_polar(x, y)
    amp = O.circle(5, r)
    wvls, wts = np.linspace(0.5, 0.7, 5), np.linspace(1.0, 2.0, 5)
    for method in ('reduce', 'a2a', 'rs'):
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_pipeline_worker, args=(r_, 2, port, q, method)) for r_ in range(2)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for f in range(3):
            opd = O.hopkins_w040(r / 5, 100.0 * (f + 1))
            ref = O.sum_of_2d_modes(np.asarray([O.intensity(O.focus(O.from_amp_and_phase(amp, opd, w), 2)) for w in wvls]), wts)
            np.testing.assert_allclose(res[0][f], ref, rtol=1e-12, atol=1e-12 * ref.max())


def test_packed_pupil_cache_identity_and_version():
    """packed_pupil(cache=True): one pack per (amplitude, OPD) tensor pair; an in-place change (version bump) or a different tensor
    object -- even one that would reuse the freed tensor's address -- misses.  The cache is opt-in (a write the version
    counter cannot see would be answered with a stale map); inference tensors have no counter and are never cached."""
    sys.path.insert(0, ROOT)
    from prysm_amd import polychromatic as pc
    pc.clear_packed_pupil_cache()
    amp, opd = torch.ones(8, 8), torch.arange(64.0).reshape(8, 8)
    assert pc.packed_pupil(amp, opd, amp, opd) is not pc.packed_pupil(amp, opd, amp, opd) and not pc._PACK_CACHE   # default: no cache
    p1 = pc.packed_pupil(amp, opd, amp, opd, True)
    assert pc.packed_pupil(amp, opd, amp, opd, True) is p1
    opd.add_(1.0)                                   # rewritten in place: stale
    p2 = pc.packed_pupil(amp, opd, amp, opd, True)
    assert p2 is not p1 and torch.equal(p2.imag, opd)
    torch.autograd.graph.increment_version(opd)     # what the library's out= writes do (_ops._bump)
    assert pc.packed_pupil(amp, opd, amp, opd, True) is not p2
    opd2 = opd.clone()
    assert pc.packed_pupil(amp, opd2, amp, opd2, True) is not pc.packed_pupil(amp, opd, amp, opd, True)
    del opd2
    assert len(pc._PACK_CACHE) <= pc._PACK_CACHE_MAX
    p3 = pc.packed_pupil(None, opd, None, opd, True)      # no amplitude map: unit amplitude
    assert torch.equal(p3.real, torch.ones(8, 8)) and pc.packed_pupil(None, opd, None, opd, True) is p3
    opd.data.add_(1.0)                              # invisible to the version counter: the documented limit of the opt-in cache ...
    assert pc.packed_pupil(None, opd, None, opd, True) is p3
    pc.clear_packed_pupil_cache()                   # ... and its remedy
    p4 = pc.packed_pupil(None, opd, None, opd, True)
    assert p4 is not p3 and torch.equal(p4.imag, opd)
    with torch.inference_mode():                    # no version counter (t._version raises): packed every time, no crash
        it = torch.arange(64.0).reshape(8, 8)
        q1 = pc.packed_pupil(None, it, None, it, True)
        assert pc.packed_pupil(None, it, None, it, True) is not q1 and torch.equal(q1.imag, it)
