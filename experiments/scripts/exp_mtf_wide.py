"""mtf_from_psf 4096^2 fp32 / fp64: 64 B-wide (default) against 128 B-wide tiles in the Hermitian column pass; value check vs numpy."""
import numpy as np
import torch

from prysm_amd import _lib as L, otf

lib = L.load()


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps)
    return float(np.median(ts)) * 1e3


for n in (4096, 2048):
    for dt in (torch.float32, torch.float64):
        psf = torch.rand(n, n, dtype=dt, device='cuda') + 0.01
        F = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(psf.cpu().numpy().astype(np.float64))))
        want = np.abs(F / F[n // 2, n // 2])
        for wide in (0, 1):
            lib.pm_set_tuning(b'herm_wide', wide)
            got = otf.mtf_from_psf(psf, 1.0).data.cpu().numpy()
            err = float(np.abs(got - want).max())
            t = timeit(lambda: otf.mtf_from_psf(psf, 1.0))
            print(f'n={n} {str(dt)[6:]} wide={wide}: {t:.1f} us  max abs err {err:.2e}', flush=True)
lib.pm_set_tuning(b'herm_wide', 0)
