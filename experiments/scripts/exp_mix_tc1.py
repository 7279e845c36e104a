"""column tiles of ONE column (16 B pieces of complex128, three workgroups per CU) against two (32 B, one workgroup per CU) at long columns"""
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


for dt, n in ((torch.complex128, 3000), (torch.complex128, 4000), (torch.complex128, 2000), (torch.complex64, 6000), (torch.complex64, 4000), (torch.complex64, 3000)):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    res = []
    for tc in (0, 1, 2, 4):
        for lg in (-1, 3):
            lib.pm_set_tuning(b'mix_tc', tc)
            lib.pm_set_tuning(b'mix_log_g', lg)
            try:
                res.append('tc=%d lg=%d %.1f' % (tc, lg, timed(lambda: _ops.fft2(x, direction=-1, scale=1.0))))
            except Exception:
                res.append('tc=%d lg=%d n/a' % (tc, lg))
    lib.pm_set_tuning(b'mix_tc', 0)
    lib.pm_set_tuning(b'mix_log_g', -1)
    print('TC1', dt, n, ' | '.join(res))
