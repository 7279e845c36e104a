"""Composite register engine (prysm_amd/csrc/fft_ce.h) against the general mixed-radix kernel and torch.fft: results on plain, padded,
rotated, inverse, non-square and ragged-tile views for every built plan and workgroup shape, then time per 2-D transform (us) with the
engine off / on and per built shape.

    python tools/exp_ce_check.py [c64|c128|both]"""
import itertools
import sys

import torch

from prysm_amd import _lib, _ops

lib = _lib.load()
which = sys.argv[1] if len(sys.argv) > 1 else 'c64'
DT = {'c64': [torch.complex64], 'c128': [torch.complex128], 'both': [torch.complex64, torch.complex128]}[which]
SIZES = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1000, 1500, 2000, 3000]
ROWS = {1000: [0, 10], 1500: [0], 2000: [0], 3000: [0, 4]}
COLS = {1000: [0, 4], 1500: [0], 2000: [0, 8], 3000: [0, 8]}


def knob(name, v):
    _lib.check(lib.pm_set_tuning(name.encode(), int(v)))


def ref(x, shape, in_off, in_shift, out_shift, direction):
    M, N = shape
    full = torch.zeros(M, N, dtype=torch.complex128, device=x.device)
    full[in_off[0]:in_off[0] + x.shape[0], in_off[1]:in_off[1] + x.shape[1]] = x
    full = torch.roll(full, (-in_shift[0], -in_shift[1]), (0, 1))
    f = torch.fft.fft2(full) if direction < 0 else torch.fft.ifft2(full) * (M * N)
    return torch.roll(f, (out_shift[0], out_shift[1]), (0, 1))


def timed(fn, reps):
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


bad = 0
torch.manual_seed(3)
for dt in DT:
    tol = 2e-5 if dt == torch.complex64 else 1e-12
    for n in SIZES:
        cases = [((n, n), (n, n), (0, 0), (0, 0), (0, 0), -1), ((n, n), (n, n), (0, 0), (n // 2, n // 2), (n // 2, n // 2), -1),
                 ((n, n), (n, n), (0, 0), (n // 2, n // 2), (n // 2, n // 2), +1), ((n // 2, n // 2), (n, n), (n // 4, n // 4), (n // 2, n // 2), (n // 2, n // 2), -1),
                 ((n, 1001 if n != 1000 else 1501), None, (0, 0), (3, 5), (7, 11), -1), ((700, n), None, (0, 0), (0, 0), (1, 1), +1),
                 ((n, 1024), None, (0, 0), (0, 0), (0, 0), -1)]
        for rs, cs in itertools.product(ROWS.get(n, [0]), COLS.get(n, [0])):
            knob('ce_rows_seqs', rs)
            knob('ce_cols_seqs', cs)
            worst = 0.0
            for xs, shape, in_off, in_shift, out_shift, direction in cases:
                shape = shape or xs
                x = torch.randn(*xs, dtype=dt, device='cuda')
                r = ref(x, shape, in_off, in_shift, out_shift, direction)
                outs = []
                for eng in (0, 1):
                    knob('mix_engine', eng)
                    y = _ops.fft2(x, direction=direction, scale=1.0, shape=shape, in_off=in_off, in_shift=in_shift, out_shift=out_shift)
                    outs.append(y)
                    err = ((y.to(torch.complex128) - r).abs().max() / r.abs().max()).item()
                    worst = max(worst, err)
                    if not err < tol:
                        bad += 1
                        print('  MISMATCH n %d %s engine %d rows %d cols %d case %s: err %.2e' % (n, dt, eng, rs, cs, (xs, shape, in_off, in_shift, out_shift, direction), err))
            print('CHECK %-10s n %4d rows-shape %2d cols-shape %2d: worst err %.2e %s' % (str(dt).split('.')[-1], n, rs, cs, worst, 'ok' if worst < tol else 'FAIL'), flush=True)
    knob('ce_rows_seqs', 0)
    knob('ce_cols_seqs', 0)
print('check: %d mismatches' % bad)

warm = torch.randn(4096, 4096, dtype=torch.complex64, device='cuda')
for _ in range(300):
    _ops.fft2(warm, direction=-1, scale=1.0)
torch.cuda.synchronize()
del warm
for dt in DT:
    for n in SIZES:
        x = torch.randn(n, n, dtype=dt, device='cuda')
        reps = max(10, min(200, int(4.0e9 / (n * n * x.element_size()))))
        fn = lambda: _ops.fft2(x, direction=-1, scale=1.0, in_shift=(n // 2, n // 2), out_shift=(n // 2, n // 2))
        line = []
        knob('mix_engine', 0)
        line.append('general %.1f' % timed(fn, reps))
        knob('mix_engine', 1)
        for rs, cs in itertools.product(ROWS.get(n, [0]), COLS.get(n, [0])):
            knob('ce_rows_seqs', rs)
            knob('ce_cols_seqs', cs)
            line.append('engine[rows %d cols %d] %.1f / %.1f' % (rs, cs, timed(fn, reps), timed(fn, reps)))
        knob('ce_rows_seqs', 0)
        knob('ce_cols_seqs', 0)
        knob('mix_engine', 0)
        line.append('general %.1f' % timed(fn, reps))
        knob('mix_engine', 1)
        print('TIME %-10s %4d^2: ' % (str(dt).split('.')[-1], n) + '   '.join(line), flush=True)
sys.exit(1 if bad else 0)
