"""The planner's class weights after the twiddle change (fewer loads per butterfly favour larger factors): 2-D transform time per length and
precision for a few (w16, w20) settings, one child process each (experiment build; the weights are read from the environment once).
usage: python tools/exp_mix_weights.py"""
import os
import subprocess
import sys

if len(sys.argv) < 2:
    for w16, w20s, w20d in ((115, 130, 140), (105, 110, 110), (100, 100, 100), (110, 120, 125), (115, 115, 120), (100, 130, 140)):
        env = dict(os.environ, PM_MIX_W16=str(w16), PM_MIX_W20S=str(w20s), PM_MIX_W20D=str(w20d))
        out = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
        print('W16 %d W20S %d W20D %d:' % (w16, w20s, w20d), out.stdout.strip() or out.stderr[-300:], flush=True)
    sys.exit(0)

import torch
from prysm_amd import _ops, _lib
lib = _lib.load()


def timed(fn, reps=10):
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


w = torch.randn(3000, 3000, dtype=torch.complex64, device='cuda')
for _ in range(300):
    _ops.fft2(w, direction=-1, scale=1.0)
res = []
for dt, ns in ((torch.complex64, (500, 1000, 1200, 1500, 1536, 2000, 2400, 3000, 3600, 4000)), (torch.complex128, (500, 1000, 1200, 1536, 2000, 3000, 4000))):
    for n in ns:
        x = torch.randn(n, n, dtype=dt, device='cuda')
        res.append('%s%d %.1f' % ('s' if dt == torch.complex64 else 'd', n, timed(lambda: _ops.fft2(x, direction=-1, scale=1.0))))
print(' '.join(res))
