"""Start-up stagger in the three passes of the fused angular-spectrum chain (4096^2 complex128 and complex64) and in the polychromatic loop
(config 5 variant F per wavelength): knobs fft_stagger (row passes), fft_stagger_col, fft_stagger_mid."""
import math
import numpy as np
import torch
from prysm_amd import _ops, _lib, propagation as P
lib = _lib.load()


def timed(fn, reps=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def setk(r, c, m):
    lib.pm_set_tuning(b'fft_stagger', r)
    lib.pm_set_tuning(b'fft_stagger_col', c)
    lib.pm_set_tuning(b'fft_stagger_mid', m)


w = torch.randn(4096, 4096, dtype=torch.complex64, device='cuda')
for _ in range(500):
    _ops.fft2(w, direction=-1, scale=1.0)
torch.cuda.synchronize()
grid = ((0, 0, 0), (-1, -1, 0), (-1, -1, 1), (-1, -1, 2), (-1, -1, 4), (-1, -1, 8), (0, 0, 1), (0, 0, 4), (8, 8, 0), (4, 4, 0), (2, 2, 0))
for dt in (torch.complex128, torch.complex64):
    x = torch.randn(4096, 4096, dtype=dt, device='cuda')
    f = lambda: P.angular_spectrum(x, 0.6328, 0.01, 10.0, Q=1)
    res = []
    for rnd in range(2):
        for r, c, m in grid:
            setk(r, c, m)
            res.append('%d/%d/%d: %.1f' % (r, c, m, timed(f)))
        res.append('|')
    print('CHAIN rows/cols/mid', 'c128' if dt == torch.complex128 else 'c64 ', ' '.join(res), flush=True)
setk(-1, -1, 0)
# config 5 variant F, 16 wavelengths at 4096^2
n, nl = 4096, 16
g = torch.Generator(device='cuda').manual_seed(1)
amp = (torch.rand((n, n), device='cuda', generator=g) > 0.2).float()
opd = torch.randn((n, n), device='cuda', generator=g) * 50
packed = _ops.pack_amp_opd(amp, opd)
ks = [2 * math.pi / wv / 1e3 for wv in np.linspace(0.5, 0.7, nl)]
acc = torch.zeros((n, n), device='cuda', dtype=torch.float32)


def poly():
    acc.zero_()
    for k in ks:
        P.focus_intensity(packed, 1, out=acc, synth=('packed', k), weight=1.0)


try:
    res = []
    for rnd in range(2):
        for r, c in ((0, 0), (-1, -1), (1, 0), (0, 1), (2, 2), (4, 4)):
            setk(r, c, 0)
            res.append('%d/%d: %.1f' % (r, c, timed(poly, 3) / nl))
        res.append('|')
    print('POLY us per wavelength rows/cols', ' '.join(res), flush=True)
except Exception as e:
    print('POLY failed', repr(e))
setk(-1, -1, 0)
