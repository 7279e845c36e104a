"""Where the time of the mixed-radix kernels goes: the 2-D transform at 3000^2 (complex64 and complex128) with parts of the kernels switched
off (experiment build, knob mix_ablate: 1 no global loads, 2 no twiddle loads, 4 no global stores, 8 no butterflies -- the results are
wrong, only the kernel durations mean anything).  Run once per value under `rocprofv3 --kernel-trace --stats` (tools/gpu_s27.sh):
ABL=<bits> python tools/exp_mix_ablate.py"""
import os
import torch
from prysm_amd import _ops, _lib

lib = _lib.load()
abl = int(os.environ.get('ABL', '0'))
if abl:
    assert lib.pm_set_tuning(b'mix_ablate', abl) == 0, 'needs the experiment build (PRYSM_AMD_LIB=prysm_amd/alt/libprysm_amd.so)'
for dt in (torch.complex64, torch.complex128):
    x = torch.randn(3000, 3000, dtype=dt, device='cuda')
    for _ in range(30):
        _ops.fft2(x, direction=-1, scale=1.0)
torch.cuda.synchronize()
