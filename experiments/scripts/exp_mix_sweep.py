"""Mixed-radix kernel: the 2-D transform (one row-pass and one column-pass kernel) over the launch-shape knobs, one knob at a time."""
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


def sweep(n, dt, key, vals, extra=()):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    res = []
    for k2, v2 in extra:
        lib.pm_set_tuning(k2, v2)
    for v in vals:
        lib.pm_set_tuning(key, v)
        try:
            t = timed(lambda: _ops.fft2(x, direction=-1, scale=1.0))
            res.append('%s=%d %.1f' % (key.decode(), v, t))
        except Exception as exc:
            res.append('%s=%d EXC %s' % (key.decode(), v, repr(exc)[:40]))
    lib.pm_set_tuning(key, -1 if key == b'mix_log_g' else 0)
    for k2, _ in extra:
        lib.pm_set_tuning(k2, 0)
    print('SWEEP %s %d^2 %s: %s' % ('c64' if dt == torch.complex64 else 'c128', n, {k.decode(): v for k, v in extra}, ' | '.join(res)))


for dt in (torch.complex64, torch.complex128):
    for n in (1000, 2000, 3000, 4000, 6000):
        sweep(n, dt, b'mix_seqs', [0, 1, 2, 3, 4])
        sweep(n, dt, b'mix_seqs', [1, 2, 4], extra=((b'mix_nt', 128),))
        sweep(n, dt, b'mix_tc', [0, 2, 4, 8])
        sweep(n, dt, b'mix_tc', [2, 4], extra=((b'mix_ntc', 256),))
