"""Mixed-radix kernel: row pass and column pass timed alone (pm_fft1 along each axis of an n x n array) over the launch-shape knobs."""
import sys
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


def sweep(n, dt, axis, key, vals, extra=()):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    res = []
    for v in vals:
        lib.pm_set_tuning(key, v)
        for k2, v2 in extra:
            lib.pm_set_tuning(k2, v2)
        try:
            t = timed(lambda: _ops.fft1(x, n, axis=axis))
            res.append('%s=%d %.1f' % (key.decode(), v, t))
        except Exception as exc:
            res.append('%s=%d EXC %s' % (key.decode(), v, repr(exc)[:60]))
    lib.pm_set_tuning(key, 0)
    for k2, _ in extra:
        lib.pm_set_tuning(k2, 0)
    b = n * n * (8 if dt == torch.complex64 else 16)
    print('SWEEP %s n=%d axis=%d %s: %s   (2 x bytes / 5 TB/s = %.1f us)' % ('c64' if dt == torch.complex64 else 'c128', n, axis, dict(extra), ' | '.join(res), 2 * b / 5e12 * 1e6))


for dt in (torch.complex64, torch.complex128):
    for n in (1000, 3000, 4000):
        for nt in (0, 256):
            sweep(n, dt, 1, b'mix_seqs', [0, 1, 2, 4], extra=((b'mix_nt', nt),))
        for lg in (0, 1, 2, 3):
            sweep(n, dt, 0, b'mix_tc', [0, 2, 4, 8], extra=((b'mix_log_g', lg),))
lib.pm_set_tuning(b'mix_log_g', -1)
