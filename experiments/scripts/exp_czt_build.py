"""config-5 variant M grid by the chirp-Z executor: cost of building the executor per wavelength against applying it."""
import time

import numpy as np
import torch

from prysm_amd import propagation as P
from prysm_amd.conf import config

config.precision = 32
n = 4096
x = torch.randn(n, n, dtype=torch.complex64, device='cuda')
for kind in ('czt', 'mdft'):
    P.prepare_executor(10.0 / n, (n, n), 0.55 * 10 / 4, (512, 512), 0.55, 100.0, kind=kind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for w in np.linspace(0.5, 0.7, 16):
        ex = P.prepare_executor(10.0 / n, (n, n), 0.55 * 10 / 4, (512, 512), float(w), 100.0, kind=kind)
    torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / 16
    P.focus_dft(x, ex)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(16):
        P.focus_dft(x, ex)
    torch.cuda.synchronize()
    ta = (time.perf_counter() - t0) / 16
    print(f'{kind}: build {tb * 1e6:.0f} us, apply {ta * 1e6:.0f} us', flush=True)
