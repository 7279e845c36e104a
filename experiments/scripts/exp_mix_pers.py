"""Column pass of the mixed-radix path as persistent workgroups with the next tile prefetched (knob mix_pers) against one tile per workgroup:
2-D transform time in us, and agreement of the two."""
import torch
from prysm_amd import _ops, _lib
lib = _lib.load()


def timed(fn, reps=10):
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


for dt, n in ((torch.complex64, 3000), (torch.complex64, 4000), (torch.complex64, 2000), (torch.complex64, 2500), (torch.complex64, 3600), (torch.complex64, 6000),
              (torch.complex128, 3000), (torch.complex128, 2000), (torch.complex128, 1500)):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    res, outs = [], {}
    for rnd in range(2):
        for pers in (0, 1):
            lib.pm_set_tuning(b'mix_pers', pers)
            outs[pers] = _ops.fft2(x, direction=-1, scale=1.0, out_shift=(n // 2, n // 2))
            res.append('%d: %.1f' % (pers, timed(lambda: _ops.fft2(x, direction=-1, scale=1.0))))
    lib.pm_set_tuning(b'mix_pers', 1)
    d = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
    print('PERS', 'c64 ' if dt == torch.complex64 else 'c128', n, ' | '.join(res), ' max rel diff %.1e' % d, flush=True)
