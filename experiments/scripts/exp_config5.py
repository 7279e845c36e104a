"""BASELINE config 5 on ONE GPU: 8 of the 64 wavelengths of a 4096^2 pupil (variant F: FFT focus + |.|^2 accumulate;
variant M: matrix-DFT onto a 512^2 grid), per-wavelength breakdown."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prysm_amd import propagation as P, _ops
from prysm_amd.conf import config
from prysm_amd.polychromatic import polychromatic_psf

def t(fn, reps=10):
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

n = 4096
ax = (np.arange(n) - n // 2) * (10.0 / n)          # make_xy_grid(n, diameter=10)
x, y = np.meshgrid(ax, ax)
r = np.hypot(x, y)
amp = (r <= 5).astype(np.float64)                  # circle(5, r)
opd = 500.0 * (r / 5) ** 4                          # 500 nm of Hopkins W040
dx = float(x[0, 1] - x[0, 0])
wvls = np.linspace(0.5, 0.7, 64)[:8]
wts = np.ones(8)
for prec in (32, 64):
    config.precision = prec
    rdt = torch.float32 if prec == 32 else torch.float64
    a = torch.from_numpy(amp).cuda().to(rdt)
    o = torch.from_numpy(opd).cuda().to(rdt)
    synth = t(lambda: P.Wavefront.from_amp_and_phase(a, o, 0.55, dx))
    wf = P.Wavefront.from_amp_and_phase(a, o, 0.55, dx)
    acc = torch.zeros((n, n), dtype=rdt, device='cuda')
    foc = t(lambda: P.focus_intensity(wf.data, 1, out=acc, weight=1.0))
    allF = t(lambda: polychromatic_psf(a, o, wvls, wts, dx, 100.0, Q=1), reps=3)
    ex = wf.prepare_executor(100.0, 0.55 * 10 / 4, 512)
    mdft = t(lambda: _ops.abs2(wf.focus_dft(ex).data, out=torch.zeros((512, 512), dtype=rdt, device='cuda'), weight=1.0))
    build = t(lambda: wf.prepare_executor(100.0, 0.55 * 10 / 4, 512), reps=5)
    allM = t(lambda: polychromatic_psf(a, o, wvls, wts, dx, 100.0, focal_dx=0.55 * 10 / 4, samples=512), reps=3)
    print(f'precision {prec}: pupil synthesis {synth:.1f} us; focus+|.|^2 accumulate {foc:.1f} us; 8 wavelengths variant F {allF:.1f} us '
          f'({allF / 8:.1f} us per wavelength); MDFT 4096^2->512^2 apply+|.|^2 {mdft:.1f} us, executor build {build:.1f} us; '
          f'8 wavelengths variant M {allM:.1f} us ({allM / 8:.1f} per wavelength)', flush=True)
config.precision = 64
