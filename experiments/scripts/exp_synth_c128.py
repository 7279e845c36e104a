"""complex128 pupil synthesis inside the row pass (fp64 sincospi per sample) against pm_pupil_synth + transform: time and error."""
import math

import numpy as np
import torch

from prysm_amd import _lib as L, _ops
from prysm_amd.propagation import focus_intensity

lib = L.load()


def timeit(fn, reps=10):
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


for n in (1024, 2048, 4096):
    amp = (torch.rand(n, n, device='cuda', dtype=torch.float64) > 0.2).double()
    opd = torch.randn(n, n, device='cuda', dtype=torch.float64) * 50
    k = 2 * math.pi / 0.55 / 1e3
    acc = torch.zeros(n, n, device='cuda', dtype=torch.float64)

    def unfused():
        P = _ops.pupil_synth(amp, opd, k, torch.complex128)
        return focus_intensity(P, 1, out=acc, weight=1.0)

    def fused():
        return focus_intensity(opd, 1, out=acc, weight=1.0, synth=(amp, k))

    def packed():
        return focus_intensity(pk, 1, out=acc, weight=1.0, synth=('packed', k))

    pk = torch.view_as_complex(torch.stack((amp, opd), dim=-1).contiguous())
    acc.zero_(); unfused(); a = acc.clone()
    acc.zero_(); fused(); b = acc.clone()
    acc.zero_(); packed(); c = acc.clone()
    print(f'n={n}: unfused {timeit(unfused):.1f} us, fused {timeit(fused):.1f} us, packed {timeit(packed):.1f} us; '
          f'max rel diff fused {float((a - b).abs().max() / a.abs().max()):.2e}, packed {float((a - c).abs().max() / a.abs().max()):.2e}', flush=True)
