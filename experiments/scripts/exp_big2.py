"""the big / Bluestein-big routes after the radix kernels lost their scratch arrays: 8000^2 (Bluestein at 16384), 16384^2, 1-D 16384 / 10240"""
import torch

from prysm_amd import propagation as P, _ops


def t(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (8000, 16384):
    x = torch.randn(n, n, dtype=torch.complex64, device='cuda')
    print(f'focus {n}^2 complex64: {t(lambda: P.focus(x, 1)):.0f} us', flush=True)
    del x
for n, batch in ((16384, 2048), (10240, 2048), (12288, 512)):
    x = torch.randn(batch, n, dtype=torch.complex64, device='cuda')
    xt = x.t().contiguous()
    print(f'fft1 n={n} batch={batch}: rows {t(lambda: _ops.fft1(x, n, axis=1)):.0f} us, columns {t(lambda: _ops.fft1(xt, n, axis=0)):.0f} us', flush=True)
