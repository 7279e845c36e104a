"""LDS slot padding of the mixed-radix kernels (MixShape pad0 / pad1, chosen by the bank model of fft_mixed.hip): 2-D transform time with
and without, per size and precision.  PROF=1: a few transforms at 3000^2 in both precisions with the pads on (PADS=0: off), for the
rocprofv3 SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE pass."""
import os
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


if os.environ.get('PROF'):
    lib.pm_set_tuning(b'mix_pad', int(os.environ.get('PADS', '1')))
    for dt in (torch.complex64, torch.complex128):
        x = torch.randn(3000, 3000, dtype=dt, device='cuda')
        for _ in range(6):
            _ops.fft2(x, direction=-1, scale=1.0)
    torch.cuda.synchronize()
    raise SystemExit(0)

for dt, n in ((torch.complex64, 3000), (torch.complex64, 4000), (torch.complex64, 2000), (torch.complex64, 1000), (torch.complex64, 1536), (torch.complex64, 500),
              (torch.complex64, 6000), (torch.complex128, 3000), (torch.complex128, 2000), (torch.complex128, 1000)):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    res = []
    for rnd in range(2):
        for pad in (0, 1):
            lib.pm_set_tuning(b'mix_pad', pad)
            res.append('pad %d: %.1f' % (pad, timed(lambda: _ops.fft2(x, direction=-1, scale=1.0))))
    lib.pm_set_tuning(b'mix_pad', 1)
    print('PAD', 'c64 ' if dt == torch.complex64 else 'c128', n, ' | '.join(res), flush=True)
