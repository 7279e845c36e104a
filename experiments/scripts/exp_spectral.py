"""Wavelength loop vs pm_fft2_spectral (groups of wavelengths per launch pair): time per wavelength and agreement.
usage: python tools/exp_spectral.py [n ...]"""
import math
import os
import sys

import numpy as np
import torch

from prysm_amd import _lib as L, _ops
from prysm_amd.propagation import focus_intensity


def timeit(fn, reps=5):
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
    rdt = torch.float64 if os.environ.get('PM_F64') else torch.float32
    nl = 16
    for n in sizes:
        for Q in (1, 2) if n <= 2048 else (1,):
            g = torch.Generator(device='cuda').manual_seed(1)
            amp = (torch.rand((n, n), device='cuda', generator=g) > 0.2).float()
            opd = torch.randn((n, n), device='cuda', generator=g) * 50
            packed = _ops.pack_amp_opd(amp.to(rdt), opd.to(rdt))
            wl = np.linspace(0.5, 0.7, nl)
            ks = [2 * math.pi / w / 1e3 for w in wl]
            wts = list(np.linspace(0.5, 1.5, nl))
            M = n * Q
            acc = torch.zeros((M, M), device='cuda', dtype=rdt)

            def run():
                acc.zero_()
                focus_intensity(packed, Q, out=acc, synth=('packed', ks[0]), spectral=(ks, wts))
            lib.pm_set_tuning(b'spectral', 1)
            t0 = timeit(run)
            ref = acc.clone()
            print(f'n={n} Q={Q}: loop {t0 * 1e3 / nl:7.1f} us/wavelength', flush=True)
            lib.pm_set_tuning(b'spectral_area_log', 30)      # also the sizes the library leaves to the loop
            for grp in [int(v) for v in os.environ.get('PMG', '2,4,8').split(',')]:
                for mode in [int(v) for v in os.environ.get('MODES', '0,1,2,3').split(',')]:
                    lib.pm_set_tuning(b'spectral', grp)
                    lib.pm_set_tuning(b'spectral_mode', mode)
                    t = timeit(run)
                    err = float((acc - ref).abs().max() / ref.abs().max())
                    print(f'    group {grp} mode {mode}: {t * 1e3 / nl:7.1f} us/wavelength  ({t0 / t:4.2f}x)  max rel err {err:.2e}', flush=True)
            lib.pm_set_tuning(b'spectral', 8)
            lib.pm_set_tuning(b'spectral_mode', 3)
            lib.pm_set_tuning(b'spectral_area_log', 24)


if __name__ == '__main__':
    main()
