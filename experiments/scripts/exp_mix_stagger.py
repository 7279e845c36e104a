"""Start-up stagger of the mixed-radix workgroups (knob mix_stagger, units of 512 cycles x 0 .. 7 by a hash of the workgroup index, first
wave of the launch only): 2-D transform time per size and precision."""
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


for dt, n in ((torch.complex64, 3000), (torch.complex64, 4000), (torch.complex64, 2000), (torch.complex64, 1000), (torch.complex64, 6000),
              (torch.complex128, 3000), (torch.complex128, 2000)):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    res = []
    for rnd in range(2):
        for sg in (0, 1, 2, 4, 8, 16):
            lib.pm_set_tuning(b'mix_stagger', sg)
            res.append('%d: %.1f' % (sg, timed(lambda: _ops.fft2(x, direction=-1, scale=1.0))))
    lib.pm_set_tuning(b'mix_stagger', 0)
    print('STAGGER', 'c64 ' if dt == torch.complex64 else 'c128', n, ' | '.join(res), flush=True)
