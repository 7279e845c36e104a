"""The engine's fold (radix-2 step of the column transform in the row pass; auto from 4096 rows) forced on at smaller sizes, after the
twiddle change: focus time in us, knob fold = -1 (auto) / 1."""
import torch
from prysm_amd import _ops, _lib, propagation as P
lib = _lib.load()


def timed(fn, reps=40):
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
for dt, n in ((torch.complex64, 2048), (torch.complex128, 2048), (torch.complex64, 1024), (torch.complex128, 1024), (torch.complex64, 512)):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    res = []
    for rnd in range(3):
        for f in (-1, 1):
            lib.pm_set_tuning(b'fold', f)
            res.append('%d: %.1f' % (f, timed(lambda: P.focus(x, 1))))
    lib.pm_set_tuning(b'fold', -1)
    print('FOLD_SMALL', 'c64 ' if dt == torch.complex64 else 'c128', n, ' | '.join(res), flush=True)
