"""focus(4096^2, Q=2) -> 8192^2 transform with zero-row skipping: time per propagation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prysm_amd import propagation as P

def t(fn, reps=20):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for dt in (torch.complex64,):
    x = torch.randn(4096, 4096, dtype=dt, device='cuda')
    print(f'{dt}: focus(4096^2, Q=2) {t(lambda: P.focus(x, 2)):.1f} us;  focus_intensity {t(lambda: P.focus_intensity(x, 2)):.1f} us; '
          f'focus(8192^2, Q=1) {t(lambda: P.focus(torch.randn(1, 1, dtype=dt, device="cuda").expand(8192, 8192).contiguous(), 1)):.1f} us (incl. alloc)')
    y = torch.randn(8192, 8192, dtype=dt, device='cuda')
    print(f'   focus(8192^2, Q=1) {t(lambda: P.focus(y, 1)):.1f} us')
