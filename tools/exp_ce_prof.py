"""Workload for rocprofv3 --kernel-trace --stats: 2-D transforms of the composite register engine's sizes with the engine on, then off.
    python tools/exp_ce_prof.py [c64|c128] [sizes]"""
import sys

import torch

from prysm_amd import _lib, _ops

lib = _lib.load()
dt = torch.complex128 if (len(sys.argv) > 1 and sys.argv[1] == 'c128') else torch.complex64
sizes = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1000, 1500, 2000, 3000]
for n in sizes:
    x = torch.randn(n, n, dtype=dt, device='cuda')
    for eng in (1, 0):
        _lib.check(lib.pm_set_tuning(b'mix_engine', eng))
        for _ in range(40):
            _ops.fft2(x, direction=-1, scale=1.0, in_shift=(n // 2, n // 2), out_shift=(n // 2, n // 2))
        torch.cuda.synchronize()
