"""planner preference: fewest stages with factors up to 32 (the default) against plans kept within a leaner kernel class"""
import torch
from prysm_amd import _ops, _lib
lib = _lib.load()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


for dt in (torch.complex64, torch.complex128):
    for n in (500, 625, 640, 768, 900, 960, 1000, 1250, 2187, 3125, 4004, 5000, 7000, 2000, 3000, 6000):
        x = torch.randn(n, n, dtype=dt, device='cuda')
        ref = torch.fft.fft2(x.to(torch.complex128))
        res = []
        for mr in (20, 16, 10):
            lib.pm_set_tuning(b'mix_maxr', mr)
            y = _ops.fft2(x, direction=-1, scale=1.0)
            err = ((y.to(torch.complex128) - ref).abs().max() / ref.abs().max()).item()
            res.append('maxr=%d %.1f (%.0e)' % (mr, timed(lambda: _ops.fft2(x, direction=-1, scale=1.0), 20 if n <= 4096 else 5), err))
        lib.pm_set_tuning(b'mix_maxr', 20)
        print('MAXR', 'c64 ' if dt == torch.complex64 else 'c128', n, ' | '.join(res))
