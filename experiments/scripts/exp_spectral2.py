"""Wavelength loop vs grouped launch pairs: round 3's kernels (fft_spectral.h, groups of 8) and round 4's four-waves-per-SIMD kernels
(fft_spectral2.h, groups of 2 .. 4, raw values re-read or kept): time per wavelength and agreement with the loop.
usage: python tools/exp_spectral2.py [n ...]        (PROF=1: one call per form at the first size, for rocprofv3)"""
import math
import os
import sys

import numpy as np
import torch

from prysm_amd import _lib as L, _ops
from prysm_amd.propagation import focus_intensity


def timeit(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    lib = L.load()
    sizes = [int(v) for v in sys.argv[1:]] or [4096, 2048, 1024]
    prof = bool(os.environ.get('PROF'))
    nl = 16
    for n in sizes:
        for Q in (1, 2) if n <= 2048 else (1,):
            g = torch.Generator(device='cuda').manual_seed(1)
            amp = (torch.rand((n, n), device='cuda', generator=g) > 0.2).float() * torch.rand((n, n), device='cuda', generator=g)
            opd = torch.randn((n, n), device='cuda', generator=g) * 50
            packed = _ops.pack_amp_opd(amp, opd)
            wl = np.linspace(0.5, 0.7, nl)
            ks = [2 * math.pi / w / 1e3 for w in wl]
            wts = list(np.linspace(0.5, 1.5, nl))
            M = n * Q
            acc = torch.zeros((M, M), device='cuda', dtype=torch.float32)

            def run(k=ks, w=wts):
                acc.zero_()
                focus_intensity(packed, Q, out=acc, synth=('packed', k[0]), spectral=(k, w))

            def setk(**kw):
                for k, v in kw.items():
                    assert lib.pm_set_tuning(k.encode(), v) == 0, k
            setk(spectral=1, spectral2=0)      # (needs the experiment build: prysm_amd/alt via PM_LIB, tools/Makefile `exp`)
            t0 = timeit(run, 3 if prof else 7)
            ref = acc.clone()
            print(f'n={n} Q={Q}: loop {t0 * 1e3 / nl:7.1f} us/wavelength', flush=True)
            forms = [('r3 groups of 8', dict(spectral=8, spectral2=0, spectral_area_log=30))]
            for grp, keep in ((2, 0), (3, 0), (4, 0), (4, 1)):
                forms.append((f'r4 groups of {grp}' + (' keep' if keep else ''), dict(spectral=8, spectral2=grp, spectral2_keep=keep, spectral_area_log=24)))
            for name, kw in forms:
                setk(**kw)
                t = timeit(run, 3 if prof else 7)
                err = float((acc - ref).abs().max() / ref.abs().max())
                print(f'    {name:22s}: {t * 1e3 / nl:7.1f} us/wavelength  ({t0 / t:4.2f}x)  max rel diff to the loop {err:.2e}', flush=True)
                # an odd count: the last group is ragged (1, 2 or 3 wavelengths)
                run(ks[:7], wts[:7])
                got = acc.clone()
                setk(spectral=1, spectral2=0)
                run(ks[:7], wts[:7])
                e7 = float((got - acc).abs().max() / acc.abs().max())
                print(f'        7 wavelengths: max rel diff {e7:.2e}', flush=True)
            setk(spectral=8, spectral2=4, spectral2_keep=0, spectral_area_log=24)
            if prof:
                return


if __name__ == '__main__':
    main()
