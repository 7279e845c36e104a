"""Start-up stagger of the engine's plain kernels (knob fft_stagger): time per transform in us, interleaved A/B, per size."""
import torch
from prysm_amd import _ops, _lib
lib = _lib.load()


def timed(fn, reps=40):
    for _ in range(5):
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


# clock pre-warm
w = torch.randn(4096, 4096, dtype=torch.complex64, device='cuda')
for _ in range(600):
    _ops.fft2(w, direction=-1, scale=1.0)
torch.cuda.synchronize()
import sys
for dt, n, grid in ((torch.complex64, 4096, ((0, 0), (1, 0), (0, 1), (1, 1), (2, 1), (1, 2), (2, 0), (0, 2), (2, 2), (3, 1), (1, 3))),
                    (torch.complex128, 4096, ((0, 0), (8, 0), (0, 8), (8, 8), (12, 12), (16, 16), (16, 8), (8, 16), (24, 24))),
                    (torch.complex64, 8192, ((0, 0), (8, 0), (0, 8), (8, 8), (12, 12), (16, 16), (24, 24), (32, 32), (16, 32), (32, 16), (48, 48))),
                    (torch.complex128, 8192, ((0, 0), (8, 8), (16, 16), (32, 32)))):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    out = torch.empty_like(x)
    f = lambda: _ops.fft2(x, direction=-1, scale=1.0, in_shift=(n // 2, n // 2), out_shift=(n // 2, n // 2), out=out)
    res = []
    for rnd in range(2):
        for r, c in grid:
            lib.pm_set_tuning(b'fft_stagger', r)
            lib.pm_set_tuning(b'fft_stagger_col', c)
            res.append('%d/%d: %.1f' % (r, c, timed(f, 40 if n <= 4096 else 10)))
        res.append('|')
    lib.pm_set_tuning(b'fft_stagger', 0)
    lib.pm_set_tuning(b'fft_stagger_col', 0)
    print('FFT_STAGGER rows/cols', 'c64 ' if dt == torch.complex64 else 'c128', n, ' '.join(res), flush=True)
