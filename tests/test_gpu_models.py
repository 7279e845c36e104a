"""GPU parity tests of whole models (round 6):

* BASELINE config 1 at its stated size: a 512^2 circular pupil, HeNe, Wavefront.focus(Q=2).intensity against the oracle and against
  the Airy pattern (the reference's own physics test, tests/test_physics.py:20-34, restated);
* the seven-plane Lyot-coronagraph model at 1024^2 that bench.py times as `model_7plane_1024` (the workload BASELINE.md's published
  figures are about: docs/source/how-tos/GPU and Exascale Computing.ipynb file line 74, prysm/propagation/coronagraph.py:12-43) --
  eager, inside graph.sequence() and as a hipGraph replay, against the oracle's restatement of the same planes;
* graph.sequence() with per-iteration temporaries of DIFFERENT content made by plain torch operations on the caller's stream
  (a dropped temporary's address handed to the next iteration's temporary), and lazy wavefronts whose maps were made
  inside the block.

Tolerances (max error / max magnitude against the fp64 oracle): complex128 1e-9, complex64 3e-5 on the matrix-DFT chains (the
tolerance of config 4, tests/test_gpu_parity.py); north_star asks 1e-5 / 1e-3.
"""
import numpy as np
import pytest
import torch

import bench
from conftest import rel_max
from oracle import prysm_oracle as O

pytestmark = pytest.mark.gpu


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


# ----------------------------------------------------------------------------- config 1 at 512^2

@pytest.mark.parametrize('rdt,tol', [(np.float64, 1e-10), (np.float32, 2e-5)])
def test_config1_512_circular_pupil_hene(pa, rdt, tol):
    """BASELINE.json configs[0] at its stated size: 512^2 circular pupil, monochromatic HeNe, Wavefront.focus() PSF -- against the
    oracle on the same arrays, and against the analytic Airy pattern along the x and y slices through the peak
    (tests/test_physics.py:20-34 at 512 samples instead of 128: epd 1 mm, efl 10 mm -> f/10, padded by Q = 3, atol 1e-3)."""
    P = pa.propagation
    samples, epd, efl, wvl = 512, 1.0, 10.0, O.HeNe
    x, y = O.make_xy_grid(samples, diameter=epd)
    r, _ = O.cart_to_polar(x, y)
    amp = O.circle(epd / 2, r).astype(rdt)
    dx = float(x[0, 1] - x[0, 0])
    wf = P.Wavefront.from_amp_and_phase(amp, None, wvl, dx)
    psf = wf.focus(efl, Q=3).intensity
    got = tonp(psf.data)
    ref = O.intensity(O.focus(amp.astype(np.complex128), 3))
    assert got.shape == ref.shape == (1536, 1536) and got.dtype == rdt
    assert rel_max(got, ref) < tol
    # the Airy slices: focal-plane sample spacing of the FFT propagation (wavefront.py:478-504 -> pupil_sample_to_psf_sample); the
    # reference scales the pupil by 3 sqrt(size) / sum so that the peak is one
    psf_dx = O.pupil_sample_to_psf_sample(dx, 1536, wvl, efl)
    assert abs(psf.dx - psf_dx) < 1e-12 * psf_dx
    c = 1536 // 2
    u = (np.arange(1536) - c) * psf_dx
    k = (3 * np.sqrt(amp.size) / amp.sum()) ** 2
    analytic = O.airydisk(u, efl / epd, wvl)
    assert np.max(np.abs(k * got[c] - analytic)) < 1e-3 and np.max(np.abs(k * got[:, c] - analytic)) < 1e-3     # PRECISION of the reference's test


# ----------------------------------------------------------------------------- the seven-plane model bench.py times

def _oracle_model7(inp, w):
    """bench.model7 restated on the oracle (numpy fp64): the same seven planes"""
    n = bench.MODEL7['n']
    fdx, ddx = bench.model7_grids(w)
    exa = O.prepare_executor(inp['dx'], n, fdx, bench.MODEL7['fpm_samples'], w, bench.MODEL7['efl'])
    exb = O.prepare_executor(inp['dx'], n, ddx, bench.MODEL7['det_samples'], w, bench.MODEL7['efl'])
    E = O.from_amp_and_phase(inp['amp'], inp['opd'], w)                       # 1 entrance pupil
    E = E * O.from_amp_and_phase(np.ones_like(inp['dm']), inp['dm'], w)       # 2 deformable mirror (phase screen)
    at_lyot = O.to_fpm_and_back(E, bench.model7_fpm(w), exa)                   # 3, 4, 5
    after = at_lyot * inp['lyot']                                             # 6
    return O.intensity(O.focus_dft(after, exb))                               # 7


@pytest.mark.parametrize('prec,tol', [(32, 3e-5), (64, 1e-9)])
def test_model_7plane_1024_vs_oracle(pa, prec, tol):
    from prysm_amd import graph as G
    from prysm_amd.conf import config
    P = pa.propagation
    inp = bench.model7_inputs()
    n = bench.MODEL7['n']
    wvls = [0.55, 0.6328]
    rdt = torch.float32 if prec == 32 else torch.float64
    prec0 = config.precision
    config.precision = prec
    try:
        dev = {k: torch.from_numpy(v).to(rdt).cuda() for k, v in inp.items() if k != 'dx'}
        per = []
        for w in wvls:
            fdx, ddx = bench.model7_grids(w)
            per.append((w, torch.from_numpy(bench.model7_fpm(w)).to(rdt).cuda(),
                        P.prepare_executor(inp['dx'], n, fdx, bench.MODEL7['fpm_samples'], w, bench.MODEL7['efl']),
                        P.prepare_executor(inp['dx'], n, ddx, bench.MODEL7['det_samples'], w, bench.MODEL7['efl'])))

        def one(k):
            w, fpm, exa, exb = per[k]
            return bench.model7(P, dev['amp'], dev['opd'], dev['dm'], fpm, dev['lyot'], w, inp['dx'], exa, exb)

        eager = [one(k).clone() for k in range(len(per))]
        assert eager[0].dtype == rdt and eager[0].shape == (bench.MODEL7['det_samples'],) * 2
        for k, w in enumerate(wvls):
            ref = _oracle_model7(inp, w)
            assert rel_max(tonp(eager[k]), ref) < tol, (prec, w)
        # the same loop inside a sequence block (two streams) and as one hipGraph: the same bits
        with G.sequence():
            seq = [one(k) for k in range(len(per))]
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(seq, eager))
        g = G.capture(lambda a, o: bench.model7(P, a, o, dev['dm'], per[0][1], dev['lyot'], per[0][0], inp['dx'], per[0][2], per[0][3]),
                      dev['amp'], dev['opd'])
        assert torch.equal(g(dev['amp'], dev['opd']), eager[0])
        # a different OPD through the captured graph: the replay reads its static inputs, not the captured values
        opd2 = dev['opd'] * 0.5
        want = bench.model7(P, dev['amp'], opd2, dev['dm'], per[0][1], dev['lyot'], per[0][0], inp['dx'], per[0][2], per[0][3])
        assert torch.equal(g(dev['amp'], opd2), want)
        # both wavelengths captured INSIDE a sequence block: a hipGraph with one branch per ring stream, the same bits again

        def both(a, o):
            with G.sequence():
                imgs = [bench.model7(P, a, o, dev['dm'], p_[1], dev['lyot'], p_[0], inp['dx'], p_[2], p_[3]) for p_ in per]
            return imgs[0] + imgs[1]

        gb = G.capture(both, dev['amp'], dev['opd'])
        for _ in range(3):
            assert torch.equal(gb(dev['amp'], dev['opd']), eager[0] + eager[1])
    finally:
        config.precision = prec0


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
