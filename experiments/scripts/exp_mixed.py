"""mixed-radix lengths (3 / 5 / 7 x 2^k) through the radix-R step against the Bluestein route (knob mixed_radix = 0)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prysm_amd import propagation as P, _lib
lib = _lib.load()
def t(fn, reps=20):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n in (1536, 2560, 3072, 3584, 5120, 6144):
    x = torch.randn(n, n, dtype=torch.complex64, device='cuda')
    lib.pm_set_tuning(b'mixed_radix', 1); a = t(lambda: P.focus(x, 1))
    lib.pm_set_tuning(b'mixed_radix', 0); b = t(lambda: P.focus(x, 1))
    lib.pm_set_tuning(b'mixed_radix', 1)
    print(f'focus {n}^2 complex64: radix step {a:8.1f} us ({4 * n * n * 8 / a / 1e6:5.2f} TB/s algorithmic), Bluestein {b:8.1f} us')
