"""config-4 geometry (2048^2 -> 512^2 complex64) by the three executors: MDFT (MFMA GEMMs), CZT (fused chirp-Z axes), FFTDFT"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prysm_amd import propagation as P
from prysm_amd.conf import config
def t(fn, reps=30):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for prec, cdt in ((32, np.complex64), (64, np.complex128)):
    config.precision = prec
    rng = np.random.default_rng(2048)
    x = torch.from_numpy((rng.standard_normal((2048, 2048)) + 1j * rng.standard_normal((2048, 2048))).astype(cdt)).cuda()
    args = (10 / 2048, (2048, 2048), 0.6328 * 10 / 8, (512, 512), 0.6328, 100.0)
    ref = None
    for kind in ('mdft', 'czt'):
        ex = P.prepare_executor(*args, kind=kind)
        out = P.focus_dft(x, ex)
        if ref is None:
            ref = out
        err = float((out - ref).abs().max() / ref.abs().max())
        g = torch.randn_like(out)
        print(f'precision {prec} {kind:6s}: focus_dft {t(lambda: P.focus_dft(x, ex)):7.1f} us, adjoint {t(lambda: P.unfocus_dft(g, ex)):7.1f} us, vs mdft {err:.2e}')
