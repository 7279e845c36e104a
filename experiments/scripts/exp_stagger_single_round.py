"""Start-up stagger forced on single-round launches with several workgroups per CU (knob value 100 + units): config 2 (2048^2 complex64:
four row workgroups per CU) and mtf_from_psf 4096^2 with 8-column Hermitian tiles (herm_wide 0: two workgroups per CU)."""
import torch
from prysm_amd import _ops, _lib, propagation as P, otf
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
for dt, n in ((torch.complex64, 2048), (torch.complex128, 2048), (torch.complex64, 1024)):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    res = []
    for rnd in range(2):
        for r, c in ((0, 0), (101, 0), (102, 0), (104, 0), (0, 101), (0, 102), (102, 102)):
            lib.pm_set_tuning(b'fft_stagger', r)
            lib.pm_set_tuning(b'fft_stagger_col', c)
            res.append('%d/%d: %.1f' % (r, c, timed(lambda: P.focus(x, 1))))
        res.append('|')
    print('SINGLE focus', 'c64 ' if dt == torch.complex64 else 'c128', n, ' '.join(res), flush=True)
lib.pm_set_tuning(b'fft_stagger', -1)
lib.pm_set_tuning(b'fft_stagger_col', -1)
psf = torch.rand(4096, 4096, dtype=torch.float32, device='cuda') + 0.01
for hw in (1, 0):
    lib.pm_set_tuning(b'herm_wide', hw)
    res = []
    for rnd in range(2):
        for r, h in ((0, 0), (101, 0), (102, 0), (104, 0), (0, 101), (0, 102), (0, 104), (0, 108), (0, 116), (102, 104)):
            lib.pm_set_tuning(b'fft_stagger_r2c', r)
            lib.pm_set_tuning(b'fft_stagger_herm', h)
            res.append('%d/%d: %.1f' % (r, h, timed(lambda: otf.mtf_from_psf(psf, 1.0), 30)))
        res.append('|')
    print('SINGLE mtf 4096 herm_wide', hw, ' '.join(res), flush=True)
