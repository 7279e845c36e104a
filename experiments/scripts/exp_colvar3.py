"""32 B column tiles (col_var 3: four complex64 / two complex128 columns on 256 threads) for 2048-point columns: a 2048^2 transform then
has 512 / 1024 column workgroups instead of one per CU.  focus time in us per (col_var, log_k), and agreement with the default tiling.
The variant measured no faster (profiles/r04/exp_colvar3.log) and was removed again: to repeat the measurement, re-add `VAR == 3` to
ColCfgSel / launch_fft (fft_kernels.h) and col_tile_width_for (pm_internal.h) as in commit 805cf35's successor."""
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
for dt, shape in ((torch.complex64, (2048, 2048)), (torch.complex128, (2048, 2048)), (torch.complex64, (2048, 1024)), (torch.complex64, (2048, 4096))):
    x = torch.randn(*shape, dtype=dt, device='cuda')
    ref = P.focus(x, 1).clone()
    res = []
    for rnd in range(2):
        for cv, lk in ((-1, -1), (3, -1), (3, 2), (3, 3), (3, 0)):
            lib.pm_set_tuning(b'col_var', cv)
            lib.pm_set_tuning(b'log_k', lk)
            d = float((P.focus(x, 1) - ref).abs().max() / ref.abs().max())
            res.append('%d/%d: %.1f (%.0e)' % (cv, lk, timed(lambda: P.focus(x, 1)), d))
        res.append('|')
    lib.pm_set_tuning(b'col_var', -1)
    lib.pm_set_tuning(b'log_k', -1)
    print('COLVAR3', 'c64 ' if dt == torch.complex64 else 'c128', shape, ' '.join(res), flush=True)
