"""Rows per workgroup / threads of the mixed-radix row kernel and columns per tile / threads of the column kernel, swept again after the
round-4 changes (interleaved rows, pads, twiddle powers, all loads of a trip in flight): 2-D transform time in us."""
import sys
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


def setk(**kw):
    for k, v in kw.items():
        assert lib.pm_set_tuning(k.encode(), v) == 0, k


cases = ((torch.complex64, 3000), (torch.complex64, 2000), (torch.complex64, 1000), (torch.complex64, 4000), (torch.complex128, 3000), (torch.complex128, 1000))
for dt, n in cases:
    x = torch.randn(n, n, dtype=dt, device='cuda')
    f = lambda: _ops.fft2(x, direction=-1, scale=1.0)
    tag = ('c64 ' if dt == torch.complex64 else 'c128') + ' %d' % n
    setk(mix_seqs=0, mix_nt=0, mix_tc=0, mix_ntc=0)
    print('SHAPE', tag, 'auto %.1f' % timed(f), flush=True)
    for seqs in (1, 2, 4, 8):
        if seqs * n * (8 if dt == torch.complex64 else 16) > 150 * 1024:
            continue
        row = []
        for nt in (64, 128, 192, 256, 384, 512):
            setk(mix_seqs=seqs, mix_nt=nt)
            row.append('%d: %.1f' % (nt, timed(f)))
        print('SHAPE', tag, 'rows per workgroup %d, threads' % seqs, ' | '.join(row), flush=True)
    setk(mix_seqs=0, mix_nt=0)
    for tc in (2, 4, 8):
        if tc * n * (8 if dt == torch.complex64 else 16) > 150 * 1024:
            continue
        row = []
        for nt in (256, 512, 768, 1024):
            setk(mix_tc=tc, mix_ntc=nt)
            row.append('%d: %.1f' % (nt, timed(f)))
        print('SHAPE', tag, 'columns per tile %d, threads' % tc, ' | '.join(row), flush=True)
    setk(mix_tc=0, mix_ntc=0)
