"""real(ifft2(fft2(obj) H)) of a real object: half-spectrum chain (folded / unfolded) against the complex chain.  usage: exp_conv.py [n ...]"""
import sys

import numpy as np
import torch

from prysm_amd import _lib as L, _ops


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps)
    return float(np.median(ts)) * 1e3


lib = L.load()
for n in [int(v) for v in sys.argv[1:]] or [1024, 2048, 4096, 8192]:
    for dt, cdt in ((torch.float32, torch.complex64), (torch.float64, torch.complex128)):
        if n == 8192 and dt == torch.float64:
            continue
        obj = torch.rand(n, n, dtype=dt, device='cuda')
        H = torch.randn(n, n, dtype=cdt, device='cuda')
        kw = dict(scale=1.0 / n ** 2, mul=H, in_shift=(n // 2, n // 2), out_shift=(n // 2, n // 2))
        res = {}
        for name, fold, real in (('half spectra, folded', 1, True), ('half spectra, unfolded', 0, True), ('complex chain', -1, False)):
            lib.pm_set_tuning(b'fold', fold)
            try:
                res[name] = timeit(lambda: _ops.fft2_mul_ifft2(obj, real_out=real, **kw))
            except Exception as exc:
                res[name] = repr(exc)[:60]
        lib.pm_set_tuning(b'fold', -1)
        print(f'n={n} {str(dt)[6:]}: ' + ', '.join(f'{k} {v:.1f} us' if isinstance(v, float) else f'{k} {v}' for k, v in res.items()), flush=True)
