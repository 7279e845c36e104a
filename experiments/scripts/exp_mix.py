"""Composite lengths: the mixed-radix kernel (csrc/fft_mixed.h) against round 2's routes (Bluestein / radix-R step), 2-D transforms."""
import sys
import numpy as np
import torch
from prysm_amd import _ops, _lib

lib = _lib.load()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


shapes = [(1000, 1000), (2000, 2000), (3000, 3000), (4000, 4000), (1536, 1536), (2560, 2560), (1001, 1001), (2592, 2592), (6000, 6000), (500, 500), (250, 250),
          (1000, 1024), (1024, 1000)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
for dt in (torch.complex64, torch.complex128):
    for shp in shapes:
        x = torch.randn(*shp, dtype=dt, device='cuda')
        ref = torch.fft.fft2(x.to(torch.complex128))
        out = []
        for mix in (0, 1, 2):
            lib.pm_set_tuning(b'mix', mix)
            try:
                y = _ops.fft2(x, direction=-1, scale=1.0)
                err = ((y.to(torch.complex128) - ref).abs().max() / ref.abs().max()).item()
                reps = 20 if shp[0] * shp[1] <= 4096 * 4096 else 5
                t = timed(lambda: _ops.fft2(x, direction=-1, scale=1.0), reps)
                out.append('mix=%d %8.1f us err %.1e' % (mix, t, err))
            except Exception as exc:
                out.append('mix=%d EXC %s' % (mix, repr(exc)[:80]))
        lib.pm_set_tuning(b'mix', 1)
        n = shp[0] * shp[1] * (8 if dt == torch.complex64 else 16)
        print('MIX %s %-12s %s   (4 x bytes / 5 TB/s = %.1f us)' % ('c64 ' if dt == torch.complex64 else 'c128', '%dx%d' % shp, ' | '.join(out), 4 * n / 5e12 * 1e6))
