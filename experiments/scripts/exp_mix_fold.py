"""The radix-2 step of a composite column transform folded into the mixed-radix row pass (knob mix_fold): 2-D transform time with / without,
and agreement of the two -- plain transform, the focus view (both rotations), |.|^2 epilogue, conjugated (inverse) direction."""
import torch
from prysm_amd import _ops, _lib
lib = _lib.load()
L = _lib


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


for dt, n in ((torch.complex64, 3000), (torch.complex64, 2000), (torch.complex64, 2400), (torch.complex64, 3600), (torch.complex64, 4000), (torch.complex64, 5000),
              (torch.complex64, 6000), (torch.complex128, 1500), (torch.complex128, 2000), (torch.complex128, 3000), (torch.complex128, 4000)):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    h = n // 2
    forms = {
        'plain': lambda: _ops.fft2(x, direction=-1, scale=1.0),
        'focus': lambda: _ops.fft2(x, direction=-1, scale=1.0 / n, in_shift=(h, h), out_shift=(h, h)),
        'inverse': lambda: _ops.fft2(x, direction=+1, scale=1.0 / n, in_shift=(h, h), out_shift=(h, h)),
        'abs2': lambda: _ops.fft2(x, direction=-1, scale=1.0, in_shift=(h, h), out_shift=(h, h), epilogue=L.PM_EPI_ABS2),
    }
    res, worst = [], 0.0
    ref = torch.fft.fft2(x.to(torch.complex128))
    for name, f in forms.items():
        outs = {}
        for fold in (0, 1):
            lib.pm_set_tuning(b'mix_fold', fold)
            outs[fold] = f()
        worst = max(worst, float((outs[0] - outs[1]).abs().max() / outs[0].abs().max()))
    lib.pm_set_tuning(b'mix_fold', 1)
    err = float((_ops.fft2(x, direction=-1, scale=1.0).to(torch.complex128) - ref).abs().max() / ref.abs().max())
    for rnd in range(2):
        for fold in (0, 1):
            lib.pm_set_tuning(b'mix_fold', fold)
            res.append('%d: %.1f' % (fold, timed(forms['focus'])))
    lib.pm_set_tuning(b'mix_fold', 1)
    print('FOLD', 'c64 ' if dt == torch.complex64 else 'c128', n, ' | '.join(res), ' max rel diff fold vs not %.1e, err vs torch c128 %.1e' % (worst, err), flush=True)
