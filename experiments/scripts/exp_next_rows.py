"""kernel breakdown of the SURVEY 8(f) rows at realistic sizes (rocprofv3 --kernel-trace): looking for kernels far off their roofline"""
import numpy as np
import torch

from prysm_amd import propagation as P, otf, fttools

HeNe = 0.6328
rng = np.random.default_rng(0)
n = 1024
c = (np.arange(n) - n // 2) * (10.0 / n)
xx, yy = np.meshgrid(c, c)
pupil = ((np.hypot(xx, yy) <= 5) * np.exp(1j * rng.standard_normal((n, n)) * 0.1))
ex = P.prepare_executor(10.0 / n, n, 0.5, 256, HeNe, 100.0)
fpm = rng.standard_normal((256, 256)) + 1j * rng.standard_normal((256, 256))
mr = P.prepare_multiresolution(pupil_dx=10.0 / n, pupil_samples=n, focal_dx=2.0, focal_samples=64, wavelength=HeNe, efl=100.0, num_levels=5,
                               fine_samples=64)
vort = P.vortex_phase_mask(2)
psf = np.abs(np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(pupil)))) ** 2
meas = np.exp(1j * 2 * np.arctan2(yy, xx))
for _ in range(4):
    P.to_fpm_and_back(pupil, fpm, ex)
    P.to_fpm_and_back_multiresolution(pupil, vort, mr)
    otf.encircled_energy(psf, 1.0, np.array([2.0, 5.0, 10.0, 20.0, 40.0, 80.0, 160.0, 320.0]))
    fttools.fourier_resample(psf, 1.5)
    f3 = P.prepare_measured_fpm(meas, 10.0 / n, charge=2, order=3)
    f3(torch.from_numpy(xx).cuda() * 0.7, torch.from_numpy(yy).cuda() * 0.7)
    torch.cuda.synchronize()
