"""where do the microseconds of focus_dft(config 4) go beyond its two GEMMs?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prysm_amd import _ops, propagation as P
from prysm_amd.conf import config
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record()
    host = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, host
config.precision = 32
rng = np.random.default_rng(2048)
x = torch.from_numpy((rng.standard_normal((2048, 2048)) + 1j * rng.standard_normal((2048, 2048))).astype(np.complex64)).cuda()
ex = P.prepare_executor(10 / 2048, (2048, 2048), 0.6328 * 10 / 8, (512, 512), 0.6328, 100.0)
print('focus_dft           gpu %.1f us, host issue %.1f us' % t(lambda: P.focus_dft(x, ex)))
print('raw pair (bases)    gpu %.1f us, host issue %.1f us' % t(lambda: _ops.cgemm(_ops.cgemm(ex.Ey, x), ex.Ex, 0, 2, alpha=ex.norm)))
r = lambda *s: torch.randn(*s, dtype=torch.complex64, device='cuda')
Ey, Ex = r(512, 2048), r(512, 2048)
print('raw pair (randn)    gpu %.1f us, host issue %.1f us' % t(lambda: _ops.cgemm(_ops.cgemm(Ey, x), Ex, 0, 2)))
print('G1 bases            gpu %.1f us, host issue %.1f us' % t(lambda: _ops.cgemm(ex.Ey, x)))
T = _ops.cgemm(ex.Ey, x)
print('G2 bases            gpu %.1f us, host issue %.1f us' % t(lambda: _ops.cgemm(T, ex.Ex, 0, 2)))
print('prepare_executor    gpu %.1f us, host issue %.1f us' % t(lambda: P.prepare_executor(10 / 2048, (2048, 2048), 0.6328 * 10 / 8, (512, 512), 0.6328, 100.0), 20))
