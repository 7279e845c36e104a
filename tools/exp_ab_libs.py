"""A/B of two builds of the library on the main workloads (child processes alternate between them; us per call):
python tools/exp_ab_libs.py <libA> <libB> [rounds]"""
import math
import os
import subprocess
import sys

if len(sys.argv) >= 3 and sys.argv[1] != '--child':
    a, b = sys.argv[1], sys.argv[2]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    for r in range(rounds):
        for tag, lib in (('A', a), ('B', b)):
            env = dict(os.environ, PRYSM_AMD_LIB=os.path.abspath(lib))
            out = subprocess.run([sys.executable, __file__, '--child'], env=env, capture_output=True, text=True)
            for line in out.stdout.splitlines():
                if line.startswith('AB'):
                    print(tag, line, flush=True)
            if out.returncode:
                print(tag, 'FAILED', out.stderr[-400:])
    sys.exit(0)

import numpy as np
import torch
from prysm_amd import _ops, _lib, propagation as P, otf
lib = _lib.load()


def timed(fn, reps=30):
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
for _ in range(600):
    _ops.fft2(w, direction=-1, scale=1.0)
torch.cuda.synchronize()
res = []
for dt, n, reps in ((torch.complex64, 4096, 40), (torch.complex64, 2048, 60), (torch.complex128, 4096, 20), (torch.complex64, 8192, 8), (torch.complex64, 1024, 60)):
    x = torch.randn(n, n, dtype=dt, device='cuda')
    o = torch.empty_like(x)
    res.append('focus %s %d: %.1f' % ('c64' if dt == torch.complex64 else 'c128', n, timed(lambda: P.focus(x, 1), reps)))
    del x, o
for n in (3000, 1000, 2000):
    x = torch.randn(n, n, dtype=torch.complex64, device='cuda')
    res.append('focus c64 %d: %.1f' % (n, timed(lambda: P.focus(x, 1), 20)))
    del x
obj = torch.rand(4096, 4096, dtype=torch.float32, device='cuda')
H = torch.randn(4096, 4096, dtype=torch.complex64, device='cuda')
try:
    from prysm_amd import convolution as CV
    res.append('conv real 4096: %.1f' % timed(lambda: CV.conv(obj, H, transfer_function=True) if 'transfer_function' in CV.conv.__code__.co_varnames else CV.conv(obj, H), 10))
except Exception as e:
    res.append('conv n/a')
del obj, H
x = torch.randn(4096, 4096, dtype=torch.complex128, device='cuda')
res.append('AS c128 4096: %.1f' % timed(lambda: P.angular_spectrum(x, 0.6328, 0.01, 10.0, Q=1), 15))
del x
psf = torch.rand(4096, 4096, dtype=torch.float32, device='cuda') + 0.01
res.append('mtf 4096: %.1f' % timed(lambda: otf.mtf_from_psf(psf, 1.0), 30))
n, nl = 4096, 16
g = torch.Generator(device='cuda').manual_seed(1)
amp = (torch.rand((n, n), device='cuda', generator=g) > 0.2).float()
opd = torch.randn((n, n), device='cuda', generator=g) * 50
packed = _ops.pack_amp_opd(amp, opd)
ks = [2 * math.pi / wv / 1e3 for wv in np.linspace(0.5, 0.7, nl)]
acc = torch.zeros((n, n), device='cuda', dtype=torch.float32)


def poly():
    acc.zero_()
    for k in ks:
        P.focus_intensity(packed, 1, out=acc, synth=('packed', k), weight=1.0)


res.append('poly per wavelength: %.1f' % (timed(poly, 3) / nl))
# accuracy of the 4096^2 complex64 transform against torch in complex128
x = torch.randn(4096, 4096, dtype=torch.complex64, device='cuda')
ref = torch.fft.fft2(x.to(torch.complex128))
got = _ops.fft2(x, direction=-1, scale=1.0).to(torch.complex128)
res.append('err4096 %.2e' % float((got - ref).abs().max() / ref.abs().max()))
print('AB', ' | '.join(res), flush=True)
