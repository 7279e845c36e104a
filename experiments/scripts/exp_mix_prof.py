"""workload for rocprofv3: a few 2-D transforms on the mixed-radix kernel (3000^2 complex64 / complex128, 1000^2 complex64)"""
import torch
from prysm_amd import _ops
for shp, dt in (((3000, 3000), torch.complex64), ((3000, 3000), torch.complex128), ((1000, 1000), torch.complex64)):
    x = torch.randn(*shp, dtype=dt, device='cuda')
    for _ in range(6):
        y = _ops.fft2(x, direction=-1, scale=1.0)
    torch.cuda.synchronize()
