"""Stacks of small composite fields: one (B, m, n) call (one launch pair on the composite register engine) against a loop of B calls,
us per field.   python tools/exp_ce_stack.py"""
import torch

from prysm_amd import _lib, _ops

lib = _lib.load()


def timed(fn, reps=20):
    for _ in range(3):
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
for _ in range(300):
    _ops.fft2(w, direction=-1, scale=1.0)
torch.cuda.synchronize()
for n, B, dt in ((384, 16, torch.complex64), (500, 16, torch.complex64), (768, 16, torch.complex64), (1000, 8, torch.complex64), (1000, 8, torch.complex128), (1536, 4, torch.complex64), (2000, 4, torch.complex64)):
    x = torch.randn(B, n, n, dtype=dt, device='cuda')
    kw = dict(direction=-1, scale=1.0, in_shift=(n // 2, n // 2), out_shift=(n // 2, n // 2))
    loop = timed(lambda: [_ops.fft2(x[b], **kw) for b in range(B)]) / B
    stack = timed(lambda: _ops.fft2(x, **kw)) / B
    _lib.check(lib.pm_set_tuning(b'mix_engine', 0))
    general = timed(lambda: _ops.fft2(x, **kw)) / B
    _lib.check(lib.pm_set_tuning(b'mix_engine', 1))
    print('STACK %s %4d^2 x %2d: loop of calls %.1f us / field | one stacked call %.1f | stacked, general kernel (field by field inside) %.1f' % (str(dt).split('.')[-1], n, B, loop, stack, general), flush=True)
