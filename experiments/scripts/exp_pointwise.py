"""Achieved bandwidth of the pointwise / synthesis kernels at 4096^2 (bytes moved / time)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prysm_amd import _ops
from prysm_amd import fttools

def t(fn, reps=30):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

n = 4096
for cd, rd, es in ((torch.complex64, torch.float32, 8), (torch.complex128, torch.float64, 16)):
    x = torch.randn(n, n, dtype=cd, device='cuda'); y = torch.randn(n, n, dtype=cd, device='cuda')
    r = torch.randn(n, n, dtype=rd, device='cuda'); a = torch.rand(n, n, dtype=rd, device='cuda')
    acc = torch.zeros(n, n, dtype=rd, device='cuda')
    hy = torch.randn(n, dtype=cd, device='cuda'); hx = torch.randn(n, dtype=cd, device='cuda')
    rows = [
        ('abs2', lambda: _ops.abs2(x), n * n * (es + es // 2)),
        ('abs2 accumulate', lambda: _ops.abs2(x, out=acc, weight=0.5), n * n * (es + es)),
        ('cmul', lambda: _ops.cmul(x, y), n * n * 3 * es),
        ('pupil_synth', lambda: _ops.pupil_synth(a, r, 0.01, cd), n * n * (es + es)),
        ('scale_sep', lambda: _ops.scale_sep(x, hy, hx, 1.0), n * n * 2 * es),
        ('outer', lambda: _ops.outer(hy, hx), n * n * es),
        ('pad2d Q=2 (2048 -> 4096)', lambda: fttools.pad2d(x[:2048, :2048].contiguous(), 2), n * n * es + 2048 * 2048 * es),
    ]
    for name, fn, nbytes in rows:
        try:
            us = t(fn)
            print(f'{str(cd)[6:]:10s} {name:28s} {us:8.1f} us  {nbytes / us / 1e6:7.2f} TB/s', flush=True)
        except Exception as exc:
            print(name, 'EXC', repr(exc)[:120])
