"""What a plain device-to-device copy reaches on this box, for the array sizes of the headline (torch's copy kernel and pm_scale_sep as a
read + write sweep): GB/s counted as bytes read + bytes written."""
import torch
from prysm_amd import _ops, _lib
lib = _lib.load()


def timed(fn, reps=50):
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


w = torch.randn(4096, 4096, dtype=torch.complex64, device='cuda')
for _ in range(500):
    _ops.fft2(w, direction=-1, scale=1.0)
for n, dt in ((2048, torch.complex64), (4096, torch.complex64), (4096, torch.complex128), (8192, torch.complex64), (8192, torch.complex128)):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    nbytes = x.numel() * x.element_size()
    t1 = timed(lambda: y.copy_(x), 50 if n <= 4096 else 10)
    # two dependent copies in -> ws -> out: the data movement of a two-pass transform with no arithmetic
    t2 = timed(lambda: (y.copy_(x), z.copy_(y)), 50 if n <= 4096 else 10)
    print('COPY %d %s: %.1f MB; one copy %.1f us = %.0f GB/s; in -> ws -> out %.1f us = %.0f GB/s' %
          (n, 'c64' if dt == torch.complex64 else 'c128', nbytes / 1e6, t1, 2 * nbytes / t1 / 1e3, t2, 4 * nbytes / t2 / 1e3), flush=True)
