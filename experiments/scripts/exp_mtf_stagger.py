"""Start-up stagger in the two kernels of the real-input transform (mtf_from_psf 4096^2 fp32): knobs fft_stagger_r2c / fft_stagger_herm."""
import torch
from prysm_amd import _ops, _lib
from prysm_amd import otf
lib = _lib.load()


def timed(fn, reps=30):
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


w = torch.randn(4096, 4096, dtype=torch.complex64, device='cuda')
for _ in range(500):
    _ops.fft2(w, direction=-1, scale=1.0)
torch.cuda.synchronize()
for hw in (1, 0):
  lib.pm_set_tuning(b'herm_wide', hw)
  for n in (4096, 8192, 2048):
      psf = torch.rand(n, n, dtype=torch.float32, device='cuda')
      f = lambda: otf.mtf_from_psf(psf, 1.0).data if hasattr(otf.mtf_from_psf(psf, 1.0), 'data') else otf.mtf_from_psf(psf, 1.0)
      g = lambda: otf.mtf_from_psf(psf, 1.0)
      res = []
      for rnd in range(2):
          for r, h in ((0, 0), (0, -1), (0, 8), (0, 16), (0, 24), (0, 32)):
              lib.pm_set_tuning(b'fft_stagger_r2c', r)
              lib.pm_set_tuning(b'fft_stagger_herm', h)
              res.append('%d/%d: %.1f' % (r, h, timed(g, 30 if n == 4096 else 8)))
          res.append('|')
      lib.pm_set_tuning(b'fft_stagger_r2c', 0)
      lib.pm_set_tuning(b'fft_stagger_herm', 0)
      print('MTF herm_wide', hw, 'r2c/herm', n, ' '.join(res), flush=True)
