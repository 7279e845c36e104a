"""Sweep of the column pass's layout knobs on the transforms whose column pass holds ONE workgroup per CU (round 5):
log_k (layout tile of the intermediate = 2^log_k column tiles), col_log_g (2^col_log_g adjacent tiles per XCD back to back),
stagger_group (start-up stagger hashed per sibling group).  us per call, best of 4 timed loops, two rounds interleaved.

    python tools/exp_layout_sweep.py [focus/c64/8192 focus/c128/4096 mtf/f32/8192 ...]"""
import itertools
import sys

import torch

from prysm_amd import _lib, propagation as P, otf

lib = _lib.load()


def timed(fn, reps):
    for _ in range(2):
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


def make(spec):
    kind, dt, n = spec.split('/')[:3]
    n = int(n)
    cd = {'c64': torch.complex64, 'c128': torch.complex128, 'f32': torch.float32, 'f64': torch.float64}[dt]
    reps = max(4, min(40, int(1.5e9 / (n * n * (16 if dt in ('c128', 'f64') else 8)))))
    if kind == 'focus':
        x = torch.randn(n, n, dtype=cd, device='cuda')
        return (lambda: P.focus(x, 1)), reps
    if kind == 'as':
        x = torch.randn(n, n, dtype=cd, device='cuda')
        return (lambda: P.angular_spectrum(x, 0.6328, 0.01, 10.0, Q=1)), reps
    psf = torch.rand(n, n, dtype=cd, device='cuda') + 0.01
    return (lambda: otf.mtf_from_psf(psf, 1.0)), reps


specs = [a for a in sys.argv[1:] if '/' in a] or ['focus/c64/8192', 'focus/c128/4096', 'mtf/f32/8192', 'mtf/f32/4096']
warm = torch.randn(4096, 4096, dtype=torch.complex64, device='cuda')
for _ in range(300):
    P.focus(warm, 1)
torch.cuda.synchronize()
del warm
for spec in specs:
    fn, reps = make(spec)
    combos = [(-1, -1, 0)] + [c for c in itertools.product((2, 3, 4, 5), (-1, 3, 4, 5), (0, 1)) if not (c[1] == -1 and c[2] == 0 and c[0] == 3)]
    res = {c: [] for c in combos}
    for r in range(2):
        for c in combos:
            for k, v in zip((b'log_k', b'col_log_g', b'stagger_group'), c):
                _lib.check(lib.pm_set_tuning(k, v))
            res[c].append(timed(fn, reps))
    base = min(res[(-1, -1, 0)])
    print('SWEEP %s: default (log_k auto, col_log_g auto, stagger_group 0) %.1f us' % (spec, base), flush=True)
    for c in sorted(combos, key=lambda c: min(res[c]))[:10]:
        print('   log_k %2d col_log_g %2d stagger_group %d: %s   (%.3f x default)' % (c + (' / '.join('%.1f' % t for t in res[c]), min(res[c]) / base)), flush=True)
    for k, v in ((b'log_k', -1), (b'col_log_g', -1), (b'stagger_group', 0)):
        lib.pm_set_tuning(k, v)
    del fn
    torch.cuda.empty_cache()
