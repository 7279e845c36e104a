"""config 5 variant F on one GPU: GPU time against host issue time per wavelength"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prysm_amd.polychromatic import polychromatic_psf
from prysm_amd import propagation as P
n = 4096
ax = (torch.arange(n, device='cuda', dtype=torch.float64) - n // 2) * (10.0 / n)
r = torch.hypot(ax[None, :], ax[:, None])
amp = (r <= 5).to(torch.float32); opd = (500.0 * (r / 5) ** 4).to(torch.float32); del r
wvls = np.linspace(0.5, 0.7, 64); wts = np.ones(64)
f = lambda: polychromatic_psf(amp, opd, wvls, wts, 10.0 / n, 100.0, Q=1, reduce_to_all=False)
f(); torch.cuda.synchronize()
for _ in range(2):
    t0 = time.perf_counter(); f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'variant F: host issue {(t1 - t0) * 1e3:.2f} ms, wall {(t2 - t0) * 1e3:.2f} ms ({(t2 - t0) / 64 * 1e6:.1f} us per wavelength)')
acc = torch.zeros((n, n), dtype=torch.float32, device='cuda')
def gpu_only():
    for k in range(64):
        P.focus_intensity(opd, 1, out=acc, weight=1.0, synth=(amp, 2 * np.pi / wvls[k] / 1e3))
gpu_only(); torch.cuda.synchronize()
t0 = time.perf_counter(); gpu_only(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'bare loop of focus_intensity(synth): host issue {(t1 - t0) * 1e3:.2f} ms, wall {(t2 - t0) * 1e3:.2f} ms ({(t2 - t0) / 64 * 1e6:.1f} us per wavelength)')
x = torch.randn(n, n, dtype=torch.complex64, device='cuda')
def nosynth():
    for k in range(64):
        P.focus_intensity(x, 1, out=acc, weight=1.0)
nosynth(); torch.cuda.synchronize()
t0 = time.perf_counter(); nosynth(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'focus_intensity of a complex field, accumulate: {(t2 - t0) / 64 * 1e6:.1f} us per call')
pr = cProfile.Profile(); pr.enable(); f(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(8)
