"""Config 5 variant M on one GPU (64 wavelengths x 4096^2 -> 512^2 matrix-DFT focus, executor rebuilt per wavelength) by GEMM form and with /
without the |.|^2 epilogue of the second product; the two products of one wavelength timed alone.  usage: exp_variant_m.py"""
import math
import time

import numpy as np
import torch

from prysm_amd import _lib as L, _ops
from prysm_amd import fttools, propagation as P
from prysm_amd.conf import config
from prysm_amd.polychromatic import polychromatic_psf

lib = L.load()
n = 4096
ax = (torch.arange(n, device='cuda', dtype=torch.float64) - n // 2) * (10.0 / n)
r = torch.hypot(ax[None, :], ax[:, None])
amp = (r <= 5).to(torch.float32)
opd = (500.0 * (r / 5) ** 4).to(torch.float32)
wvls, wts = np.linspace(0.5, 0.7, 64), np.ones(64)
config.precision = 32


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


def ev(fn, reps=20):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


fused = fttools.MDFT.intensity
for wk in (0, 1, 5):
    lib.pm_set_tuning(b'gemm_wk', wk)
    for name, meth in (('|.|^2 in the second product', fused), ('composed (cgemm + pm_abs2)', None)):
        if meth is None:
            del fttools.MDFT.intensity
        t = timed(lambda: polychromatic_psf(amp, opd, wvls, wts, 10.0 / n, 100.0, focal_dx=0.55 * 10 / 4, samples=512, kind='mdft', reduce_to_all=False))
        if meth is None:
            fttools.MDFT.intensity = fused
        print(f'gemm_wk={wk} variant M, {name}: {t:.2f} ms', flush=True)
    ex = P.prepare_executor(10.0 / n, (n, n), 0.55 * 10 / 4, (512, 512), 0.6, 100.0)
    x = torch.randn(n, n, dtype=torch.complex64, device='cuda')
    t1 = ev(lambda: _ops.cgemm(ex.Ey, x, 0, 0))
    y = _ops.cgemm(ex.Ey, x, 0, 0)
    t2 = ev(lambda: _ops.cgemm(y, ex.Ex, 0, 2))
    acc = torch.zeros(512, 512, device='cuda')
    t3 = ev(lambda: _ops.cgemm_abs2(y, ex.Ex, 0, 2, out=acc, weight=1.0))
    print(f'gemm_wk={wk} products of one wavelength: Ey @ a {t1:.1f} us, (.) @ Ex^T {t2:.1f} us, with the |.|^2 epilogue {t3:.1f} us', flush=True)
lib.pm_set_tuning(b'gemm_wk', 5)
