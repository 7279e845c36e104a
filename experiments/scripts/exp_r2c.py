"""real-input transforms: the Hermitian path against the complex path (knob r2c = 0), transform_psf and mtf_from_psf"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prysm_amd import otf, _lib, propagation as P
lib = _lib.load()
def t(fn, reps=30):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n, dt in ((4096, torch.float32), (2048, torch.float32), (4096, torch.float64), (8192, torch.float32)):
    psf = torch.rand(n, n, dtype=dt, device='cuda') + 0.01
    z = torch.complex(psf, psf)
    row = {}
    for r2c in (2, 0):
        lib.pm_set_tuning(b'r2c', r2c)
        row[r2c] = (t(lambda: otf.transform_psf(psf, 1.0)), t(lambda: otf.mtf_from_psf(psf, 1.0)))
    lib.pm_set_tuning(b'r2c', 1)
    cplx = t(lambda: P.focus(z, 1))
    print(f'{n}^2 {str(dt)[6:]}: transform_psf hermitian {row[2][0]:7.1f} us, complex path {row[0][0]:7.1f} us | mtf_from_psf fused {row[2][1]:7.1f} us, '
          f'composed {row[0][1]:7.1f} us | complex focus of the same size {cplx:7.1f} us -> mtf / focus = {row[2][1] / cplx:.2f}')
