"""Sweep of the composite register engine's built alternatives (experiment build: python tools/ce_gen.py exp; make EXTRA=-DPM_CE_EXP):
per precision and length the 2-D transform time (us, best of timed loops) with the general kernel, then each row shape with the shipped
column shape, then each column shape x XCD grouping with the best row shape.

    python tools/exp_ce_sweep.py [f32|f64|both] [lengths]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ce_gen import TABLE  # noqa: E402
from prysm_amd import _lib, _ops  # noqa: E402

lib = _lib.load()
which = sys.argv[1] if len(sys.argv) > 1 else 'both'
only = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else None
DEFAULTS = {'ce_rows_seqs': 0, 'ce_cols_seqs': 0, 'ce_stagger': 0, 'ce_log_g': -1, 'mix_engine': 1}


def setk(**kv):
    d = dict(DEFAULTS)
    d.update(kv)
    for k, v in d.items():
        _lib.check(lib.pm_set_tuning(k.encode(), v))


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = 1e9
    for _ in range(6):
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = min(t, e0.elapsed_time(e1) / 30 * 1e3)
    return t


w = torch.randn(4096, 4096, dtype=torch.complex64, device='cuda')
for _ in range(300):
    _ops.fft2(w, direction=-1, scale=1.0)
torch.cuda.synchronize()
del w
for prec in (['f32', 'f64'] if which == 'both' else [which]):
    dt = torch.complex64 if prec == 'f32' else torch.complex128
    for n in sorted(TABLE[prec]):
        if only and n not in only:
            continue
        plan, rs, cs, ra, ca = TABLE[prec][n]
        x = torch.randn(n, n, dtype=dt, device='cuda')
        ref = torch.fft.fftshift(torch.fft.fft2(torch.fft.ifftshift(x.to(torch.complex128))))
        fn = lambda: _ops.fft2(x, direction=-1, scale=1.0, in_shift=(n // 2, n // 2), out_shift=(n // 2, n // 2))     # focus(): the rotations of both sides folded into the passes
        setk(mix_engine=0)
        t_gen = timed(fn)
        print('SWEEP %s %d^2: general kernel %.1f us' % (prec, n, t_gen), flush=True)
        best_r, best_t = None, 1e9
        for sh in [rs] + ra:
            key = 1000 * sh[1] + sh[0]
            setk(ce_rows_seqs=key)
            err = ((fn() - ref).abs().max() / ref.abs().max()).item()
            t = timed(fn)
            print('   rows %s (cols shipped): %.1f us   err %.1e' % (sh, t, err), flush=True)
            if t < best_t:
                best_r, best_t = key, t
        rows = []
        for sh in [cs] + ca:
            key = 1000 * sh[1] + sh[0]
            for lg in (-1, 2, 3, 4, 5, 6):
                setk(ce_rows_seqs=best_r, ce_cols_seqs=key, ce_log_g=lg)
                err = ((fn() - ref).abs().max() / ref.abs().max()).item()
                rows.append((timed(fn), sh, lg, err))
        rows.sort()
        for t, sh, lg, err in rows[:6]:
            print('   rows %d, cols %s log_g %2d: %.1f us   err %.1e' % (best_r, sh, lg, t, err), flush=True)
        setk(mix_engine=0)
        print('   general again %.1f   -> best %.1f (%.2f x)' % (timed(fn), rows[0][0], t_gen / rows[0][0]), flush=True)
        del x, ref
        torch.cuda.empty_cache()
setk()
