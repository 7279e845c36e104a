"""config 5 variant F (64 wavelengths x 4096^2): the C loop inside pm_fft2_spectral against the Python loop of pm_fft2 calls."""
import numpy as np
import torch

from prysm_amd.conf import config
from prysm_amd.polychromatic import polychromatic_psf

n = 4096
ax = (torch.arange(n, device='cuda', dtype=torch.float64) - n // 2) * (10.0 / n)
r = torch.hypot(ax[None, :], ax[:, None])
amp = (r <= 5).to(torch.float32)
opd = (500.0 * (r / 5) ** 4).to(torch.float32)
wvls = np.linspace(0.5, 0.7, 64)
wts = np.ones(64)
config.precision = 32


def timed(fn, reps=5):
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
    return sorted(ts)


for rnd in range(3):
    for name, kw in (('C loop (pm_fft2_spectral)', dict()), ('python loop (pm_fft2)', dict(spectral=False))):
        ts = timed(lambda: polychromatic_psf(amp, opd, wvls, wts, 10.0 / n, 100.0, Q=1, reduce_to_all=False, **kw))
        print(f'{name:28s} median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f}', flush=True)
