"""Experiment: cost of the tiled intermediate layout.  Time the row pass with a NATURAL store (pm_fft1 axis=1) and
the column pass with a NATURAL load (pm_fft1 axis=0) against the two passes of pm_fft2 (tiled store / tiled load)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prysm_amd import _ops, _lib as L

def t(fn, reps=30):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for n in (2048, 4096):
    for dt in (torch.complex64, torch.complex128):
        x = torch.randn(n, n, dtype=dt, device='cuda')
        lib = L.load()
        def row(): return _ops.fft1(x, axis=1)
        def col(): return _ops.fft1(x, axis=0)
        def both(): return _ops.fft1(_ops.fft1(x, axis=1), axis=0)
        def f2(): return _ops.fft2(x, direction=-1, scale=1.0)
        print(f'n={n} {dt}: row-nat {t(row):.1f} us  col-nat {t(col):.1f} us  row-nat+col-nat {t(both):.1f} us  fft2(tiled) {t(f2):.1f} us', flush=True)
