"""CPU, world_size 2, gloo: the N > 1 path of the polychromatic driver (sharding + one sum-reduce).

The per-wavelength propagation is injected, so here it is the CPU ORACLE on CPU tensors -- the
HIP path itself is exercised by the -m gpu tests; this test covers the distributed logic that the
driver's 8-GPU run relies on (contiguous wavelength blocks, weights travel with them, one reduce).
"""
import os
import socket
import sys

import numpy as np
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


def _worker(rank, world, port, reduce_to_all, q):
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

    out = incoherent_sum(propagate, wvls, wts, reduce_to_all=reduce_to_all)
    lo, hi = shard_bounds(7, rank, world)
    assert seen == [float(w) for w in wvls[lo:hi]]
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(reduce_to_all):
    sys.path.insert(0, ROOT)
    from oracle import prysm_oracle as O
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, reduce_to_all, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
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
