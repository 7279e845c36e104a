"""threads per workgroup of the mixed-radix passes: lane utilisation (butterflies per stage / (rounds x threads)) against waves per CU"""
import torch
from prysm_amd import _ops, _lib
lib = _lib.load()


def timed(fn, reps=10):
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


for dt, n in ((torch.complex64, 3000), (torch.complex64, 4000), (torch.complex64, 2000), (torch.complex64, 1000), (torch.complex64, 6000), (torch.complex64, 1536),
              (torch.complex128, 3000), (torch.complex128, 2000)):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    for key in (b'mix_nt', b'mix_ntc'):
        res = []
        for nt in (0, 64, 128, 192, 256, 320, 384, 448, 512):
            lib.pm_set_tuning(key, nt)
            res.append('%d: %.1f' % (nt, timed(lambda: _ops.fft2(x, direction=-1, scale=1.0))))
        lib.pm_set_tuning(key, 0)
        print('NT', 'c64 ' if dt == torch.complex64 else 'c128', n, key.decode(), ' | '.join(res))
