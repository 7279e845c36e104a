"""GPU parity tests of whole models (round 6):

* BASELINE config 1 at its stated size: a 512^2 circular pupil, HeNe, Wavefront.focus(Q=2).intensity against the oracle and against
  the Airy pattern (the reference's own physics test, tests/test_physics.py:20-34, restated);
* the seven-plane Lyot-coronagraph model at 1024^2 that bench.py times as `model_7plane_1024` (the workload BASELINE.md's published
  figures are about: docs/source/how-tos/GPU and Exascale Computing.ipynb file line 74, prysm/propagation/coronagraph.py:12-43) --
  eager, inside graph.sequence() and as a hipGraph replay, against the oracle's restatement of the same planes.

Tolerances (max error / max magnitude against the fp64 oracle): complex128 1e-9, complex64 3e-5 on the matrix-DFT chains (the
tolerance of config 4, tests/test_gpu_parity.py); north_star asks 1e-5 / 1e-3.
"""
import numpy as np
import pytest
import torch

import bench
from conftest import rel_max
from gpu_common import tonp
from oracle import prysm_oracle as O

pytestmark = pytest.mark.gpu


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
