"""focus_intensity 4096^2 complex64 (|.|^2 epilogue: 32 B pieces per workgroup): 64 B tiles (col_var 0) against 128 B tiles (col_var 2)."""
import numpy as np
import torch

from prysm_amd import _lib as L, _ops
from prysm_amd.propagation import focus_intensity

lib = L.load()


def timeit(fn, reps=20):
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


n = 4096
x = torch.randn(n, n, dtype=torch.complex64, device='cuda')
amp = (torch.rand(n, n, device='cuda') > 0.2).float()
opd = torch.randn(n, n, device='cuda') * 50
packed = _ops.pack_amp_opd(amp, opd)
acc = torch.zeros(n, n, device='cuda')
for var in (0, 2, 0, 2):
    lib.pm_set_tuning(b'col_var', var)
    t1 = timeit(lambda: focus_intensity(x, 1))
    t2 = timeit(lambda: focus_intensity(x, 1, out=acc, weight=1.0))
    t3 = timeit(lambda: focus_intensity(packed, 1, out=acc, weight=1.0, synth=('packed', 0.0114)))
    print(f'col_var={var}: |focus|^2 {t1:.1f} us, accumulate {t2:.1f} us, packed synthesis + accumulate {t3:.1f} us', flush=True)
lib.pm_set_tuning(b'col_var', -1)
