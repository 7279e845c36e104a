"""real convolution chain (real(ifft2(fft2(obj) H)), half spectra end to end) at 4096^2 / 2048^2: knob fold -1 / 0 / 1, log_k"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from prysm_amd import _ops, _lib as L
for n, dt, cdt in ((4096, torch.float32, torch.complex64), (2048, torch.float32, torch.complex64), (4096, torch.float64, torch.complex128)):
    obj = torch.rand(n, n, dtype=dt, device='cuda')
    Hc = torch.randn(n, n, dtype=cdt, device='cuda')
    kw = dict(scale=1.0 / n ** 2, mul=Hc, in_shift=(n // 2, n // 2), out_shift=(n // 2, n // 2))
    for kn in (dict(), dict(fold=0), dict(fold=1), dict(fold=1, log_k=3), dict(fold=0, log_k=3), dict(fold=0, log_k=1)):
        with L.tuning_local(**kn):
            t = bench._event_ms(lambda: _ops.fft2_mul_ifft2(obj, real_out=True, **kw), 30) * 1e3
        print(n, str(dt)[6:], kn, f'{t:.1f} us', flush=True)
