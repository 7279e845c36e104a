"""Workload for rocprofv3 --kernel-trace --stats (tools/kstats.sh) or plain timing: built shapes / ablations of the composite register
engine, one after the other (experiment build: make EXTRA=-DPM_CE_EXP).

    python tools/exp_ce_variants.py <n> <c64|c128> <mode: prof|time> <setting> [<setting> ...]
    setting: knob=value[,knob=value...]   e.g.  ce_rows_seqs=212,ce_stagger=4     (knobs not named are reset to their defaults)"""
import sys

import torch

from prysm_amd import _lib, _ops

lib = _lib.load()
n = int(sys.argv[1])
dt = torch.complex128 if sys.argv[2] == 'c128' else torch.complex64
mode = sys.argv[3]
DEFAULTS = {'ce_rows_seqs': 0, 'ce_cols_seqs': 0, 'ce_stagger': 0, 'ce_log_g': -1, 'mix_engine': 1}
x = torch.randn(n, n, dtype=dt, device='cuda')
w = torch.randn(4096, 4096, dtype=torch.complex64, device='cuda')
for _ in range(200):
    _ops.fft2(w, direction=-1, scale=1.0)
torch.cuda.synchronize()


def fn():
    return _ops.fft2(x, direction=-1, scale=1.0, in_shift=(n // 2, n // 2), out_shift=(n // 2, n // 2))


ref = None
for setting in sys.argv[4:]:
    kv = dict(DEFAULTS)
    for item in setting.split(','):
        k, v = item.split('=')
        kv[k] = int(v)
    for k, v in kv.items():
        _lib.check(lib.pm_set_tuning(k.encode(), v))
    y = fn()
    if ref is None:
        ref = torch.fft.fftshift(torch.fft.fft2(torch.fft.ifftshift(x.to(torch.complex128))))
    err = ((y - ref).abs().max() / ref.abs().max()).item()
    if mode == 'prof':
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        print('%-60s err %.1e' % (setting, err), flush=True)
    else:
        best = []
        for _ in range(2):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t = 1e9
            for _ in range(5):
                e0.record()
                for _ in range(40):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                t = min(t, e0.elapsed_time(e1) / 40 * 1e3)
            best.append(t)
        print('%-60s %6.1f / %6.1f us   err %.1e' % (setting, best[0], best[1], err), flush=True)
