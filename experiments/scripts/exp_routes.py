"""kernel breakdown of the awkward-length routes under rocprofv3: 1000^2 / 3000^2 (Bluestein both axes, fused chirps), 1000 x 1024 (one axis)"""
import torch

from prysm_amd import propagation as P

for shape in ((1000, 1000), (3000, 3000), (1000, 1024), (1024, 1000)):
    x = torch.randn(*shape, dtype=torch.complex64, device='cuda')
    for _ in range(5):
        y = P.focus(x, 1)
    torch.cuda.synchronize()
