"""Interleaved A/B of ONE tuning knob on a list of workloads (us per call, best of a few timed loops per setting per round):

    python tools/exp_knob_ab.py stagger_group 0,1 focus/c64/8192 mtf/f32/4096 as/c128/4096 [rounds]

workloads: focus/<c64|c128>/<n>[/Q]  unfocus/...  as/<c64|c128>/<n>  mtf/<f32|f64>/<n>  mdft/<c64|c128>/<n>  synth/f32/<n>[/Q]"""
import sys

import torch

from prysm_amd import _lib, propagation as P, otf

lib = _lib.load()


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


def make(spec):
    parts = spec.split('/')
    kind, dt, n = parts[0], parts[1], int(parts[2])
    cd = {'c64': torch.complex64, 'c128': torch.complex128, 'f32': torch.float32, 'f64': torch.float64}[dt]
    reps = max(4, min(60, int(2.0e9 / (n * n * (16 if dt in ('c128', 'f64') else 8)))))
    if kind in ('focus', 'unfocus'):
        q = float(parts[3]) if len(parts) > 3 else 1
        x = torch.randn(n, n, dtype=cd, device='cuda')
        f = P.focus if kind == 'focus' else P.unfocus
        return (lambda: f(x, q)), reps
    if kind == 'as':
        x = torch.randn(n, n, dtype=cd, device='cuda')
        q = int(parts[3]) if len(parts) > 3 else 1
        if dt == 'c64':
            from prysm_amd.conf import config
            config.precision = 32
        return (lambda: P.angular_spectrum(x, 0.6328, 0.01, 10.0, Q=q)), reps
    if kind == 'mdft':     # config 4: matrix-DFT focus n^2 -> 512^2 (two complex GEMMs on MFMA)
        from prysm_amd.conf import config
        config.precision = 32 if dt == 'c64' else 64
        x = torch.randn(n, n, dtype=cd, device='cuda')
        ex = P.prepare_executor(10 / n, (n, n), 0.6328 * 10 / 8, (512, 512), 0.6328, 100.0)
        return (lambda: P.focus_dft(x, ex)), 30
    if kind == 'synth':     # Wavefront.from_amp_and_phase(amp, opd).focus(Q): the pupil synthesised in the row loads, |.|^2 not taken
        from prysm_amd import _ops
        q = int(parts[3]) if len(parts) > 3 else 1
        opd = (300 * torch.randn(n, n, dtype=cd, device='cuda'))
        amp = (torch.rand(n, n, device='cuda') > 0.2).to(cd)
        N = n * q
        off = (N - n) // 2
        return (lambda: _ops.fft2(opd, direction=-1, scale=1.0, shape=(N, N), in_off=(off, off), in_shift=(N // 2, N // 2), out_shift=(N // 2, N // 2),
                                  synth=(amp, 2 * 3.141592653589793 / 0.55 / 1e3))), reps
    if kind == 'mtf':
        psf = torch.rand(n, n, dtype=cd, device='cuda') + 0.01
        return (lambda: otf.mtf_from_psf(psf, 1.0)), reps
    raise SystemExit('unknown workload ' + spec)


knob = sys.argv[1].encode()
values = [int(v) for v in sys.argv[2].split(',')]
specs = [a for a in sys.argv[3:] if '/' in a]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
warm = torch.randn(4096, 4096, dtype=torch.complex64, device='cuda')
for _ in range(400):
    P.focus(warm, 1)
torch.cuda.synchronize()
del warm
for spec in specs:
    fn, reps = make(spec)
    rows = {v: [] for v in values}
    for r in range(rounds):
        for v in values:
            _lib.check(lib.pm_set_tuning(knob, v))
            rows[v].append(timed(fn, reps))
    print('KNOB %s  %-22s ' % (knob.decode(), spec) + '   '.join('%s=%d: ' % (knob.decode(), v) + ' / '.join('%.1f' % t for t in rows[v]) for v in values), flush=True)
    del fn
    torch.cuda.empty_cache()
