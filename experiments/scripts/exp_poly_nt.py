"""Config 5 variant F on one GPU (64 wavelengths x 4096^2 fp32, pupil synthesised in the row load from the packed map, |.|^2 accumulated by the
column pass) with the packed map read by plain / streaming loads (nt_in) and the image accumulated with plain / streaming stores.  usage: exp_poly_nt.py"""
import time

import numpy as np
import torch

from prysm_amd import _lib as L
from prysm_amd.polychromatic import polychromatic_psf

lib = L.load()
for n in (4096, 2048):
    ax = (torch.arange(n, device='cuda', dtype=torch.float64) - n // 2) * (10.0 / n)
    r = torch.hypot(ax[None, :], ax[:, None])
    amp = (r <= 5).to(torch.float32)
    opd = (500.0 * (r / 5) ** 4).to(torch.float32)
    wvls, wts = np.linspace(0.5, 0.7, 64), np.ones(64)
    for key, vals in ((b'nt_in', (-1, 0, 1)),):
        for v in vals:
            lib.pm_set_tuning(key, v)
            fn = lambda: polychromatic_psf(amp, opd, wvls, wts, 10.0 / n, 100.0, Q=1, reduce_to_all=False)   # noqa: E731
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            print(f'n={n} {key.decode()}={v}: {sorted(ts)[2] * 1e3:.3f} ms per PSF ({sorted(ts)[2] * 1e6 / 64:.1f} us per wavelength)', flush=True)
        lib.pm_set_tuning(key, -1)
