"""GPU parity tests, by component: hipGraph capture helpers, StreamRing and graph.sequence() blocks (prysm_amd/graph.py).

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


def test_stream_ring_sequence_matches_one_stream(pa):
    """prysm_amd.graph.StreamRing: a sequence of independent propagations alternating between two HIP streams gives, bit for bit, what
    one stream gives (per-stream workspaces, stream-aware allocator), and join() orders the caller's stream behind all of them"""
    from prysm_amd.graph import StreamRing
    rng = np.random.default_rng(4)
    fields = [torch.from_numpy(crandn(rng, (512, 512), np.complex64)).cuda() for _ in range(6)]
    want = [pa.propagation.focus(f, 2).clone() for f in fields]
    torch.cuda.synchronize()
    ring = StreamRing(2)
    ring.fork()
    outs = [ring.run(pa.propagation.focus, f, 2) for f in fields]
    ring.join()
    total = sum(o.abs().sum() for o in outs)        # consumed on the caller's stream, after the join
    torch.cuda.synchronize()
    assert all(torch.equal(o, w) for o, w in zip(outs, want)) and float(total) > 0
    assert StreamRing.worth_it((2048, 2048)) and not StreamRing.worth_it((4096, 4096))


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


# ----------------------------------------------------------------------------- sequence blocks and temporaries

def test_sequence_block_with_fresh_temporaries_each_iteration(pa):
    """every iteration builds its pupil amplitude with a plain torch operation on the caller's stream (different content each time),
    drops it, and the caching allocator hands the same address to the next iteration's temporary: each must be ordered behind the
    caller's stream on its own (the once-per-address shortcut of round 5 let iteration k + 1 read iteration k's bytes or
    half-written ones)"""
    from prysm_amd import graph as G
    P = pa.propagation
    n = 1024
    g = torch.Generator(device='cuda').manual_seed(11)
    base = torch.rand((n, n), device='cuda', generator=g, dtype=torch.float32)
    opd = torch.randn((n, n), device='cuda', generator=g, dtype=torch.float32) * 30
    ks = list(range(1, 13))

    def psf(k):
        amp = (base * k).sin().abs()            # a temporary made on the caller's stream: several kernels, different content per k
        amp = amp + 0.25 * (base > 0.1 * k)     # (the intermediate temporaries are dropped at once: their blocks are recycled)
        return P.Wavefront.from_amp_and_phase(amp, opd, 0.6, 0.01).focus(100.0, Q=1).intensity.data

    want = [psf(k).clone() for k in ks]
    torch.cuda.synchronize()
    for _ in range(3):       # the race is a matter of timing: a few trips
        with G.sequence():
            outs = [psf(k) for k in ks]
        torch.cuda.synchronize()
        assert all(torch.equal(o, w) for o, w in zip(outs, want))
    # ... and an input the caller REWRITES in place between two calls (same tensor, same address, new version)
    buf = torch.empty((n, n), device='cuda', dtype=torch.float32)
    want2 = []
    for k in ks[:6]:
        buf.copy_((base * k).cos().abs())
        want2.append(P.Wavefront.from_amp_and_phase(buf, opd, 0.6, 0.01).focus(100.0, Q=1).intensity.data.clone())
    torch.cuda.synchronize()
    with G.sequence() as seq:
        outs2 = []
        for k in ks[:6]:
            seq.join()                     # the caller rewrites a buffer the ring may still be reading: join first (the documented rule)
            buf.copy_((base * k).cos().abs())
            outs2.append(P.Wavefront.from_amp_and_phase(buf, opd, 0.6, 0.01).focus(100.0, Q=1).intensity.data)
    torch.cuda.synchronize()
    assert all(torch.equal(o, w) for o, w in zip(outs2, want2))


def test_sequence_block_orders_lazy_wavefronts_behind_their_maps(pa):
    """a lazy wavefront (from_amp_and_phase: no array yet, the maps held) whose OPD was summed INSIDE the block on one ring stream:
    .intensity / arithmetic on it must follow that stream, and the array it materialises inside a call belongs to that call's
    stream"""
    from prysm_amd import graph as G
    from prysm_amd import _ops
    P = pa.propagation
    n = 768      # a composite grid without a synthesising loader at this precision mix: the pupil is materialised
    g = torch.Generator(device='cuda').manual_seed(5)
    modes = torch.randn((6, n, n), device='cuda', generator=g, dtype=torch.float64)
    amp = (torch.rand((n, n), device='cuda', generator=g) > 0.2).double()
    ws = [torch.randn(6, generator=torch.Generator().manual_seed(i), dtype=torch.float64) * 20 for i in range(8)]

    def chain(w):
        opd = _ops.sum_modes(modes, w.tolist())                       # made on a ring stream inside the block
        wf = P.Wavefront.from_amp_and_phase(amp, opd, 0.6, 0.01)      # lazy
        i0 = wf.intensity.data                                        # materialises wf's array inside a sequenced call
        psf = (wf * wf).focus(100.0, Q=1).intensity.data              # reads the materialised array
        return i0, psf

    want = [tuple(t.clone() for t in chain(w)) for w in ws]
    torch.cuda.synchronize()
    for _ in range(3):
        with G.sequence():
            outs = [chain(w) for w in ws]
        torch.cuda.synchronize()
        for (a, b), (wa, wb) in zip(outs, want):
            assert torch.equal(a, wa) and torch.equal(b, wb)
    ref_opd = np.tensordot(ws[0].numpy(), tonp(modes), axes=1)
    ref = O.from_amp_and_phase(tonp(amp), ref_opd, 0.6)
    assert rel_max(tonp(want[0][1]), O.intensity(O.focus(ref * ref, 1))) < 1e-9
