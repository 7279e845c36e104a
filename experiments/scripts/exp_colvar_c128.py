"""Column tiling of the folded 4096^2 complex128 transform again (col_var 2 = 128 B tiles on 1024 threads, one workgroup per CU; 0 = 64 B
tiles on 512 threads, two per CU) after the stagger and twiddle changes; and log_k (row piece length of the tiled intermediate)."""
import torch
from prysm_amd import _ops, _lib, propagation as P
lib = _lib.load()


def timed(fn, reps=20):
    for _ in range(4):
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
x = torch.randn(4096, 4096, dtype=torch.complex128, device='cuda')
res = []
for rnd in range(2):
    for cv, sc in ((-1, -1), (0, -1), (0, 0), (0, 1), (0, 2), (2, 0), (2, 4), (2, 8)):
        lib.pm_set_tuning(b'col_var', cv)
        lib.pm_set_tuning(b'fft_stagger_col', sc)
        res.append('%d/%d: %.1f' % (cv, sc, timed(lambda: P.focus(x, 1))))
    res.append('|')
lib.pm_set_tuning(b'col_var', -1)
lib.pm_set_tuning(b'fft_stagger_col', -1)
print('COLVAR c128 4096 col_var/stagger_col', ' '.join(res), flush=True)
for dt, n in ((torch.complex128, 4096), (torch.complex64, 8192), (torch.complex64, 4096)):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    res = []
    for rnd in range(2):
        for lk in (-1, 0, 1, 2, 3, 4, 5, 7):
            lib.pm_set_tuning(b'log_k', lk)
            res.append('%d: %.1f' % (lk, timed(lambda: P.focus(x, 1), 20 if n == 4096 else 8)))
        res.append('|')
    lib.pm_set_tuning(b'log_k', -1)
    print('LOGK', 'c64 ' if dt == torch.complex64 else 'c128', n, ' '.join(res), flush=True)
