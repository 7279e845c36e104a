import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prysm_amd import otf
for n, dt in ((4096, torch.float32), (4096, torch.float64)):
    psf = torch.rand(n, n, dtype=dt, device='cuda') + 0.01
    for _ in range(20):
        otf.transform_psf(psf, 1.0)
        otf.mtf_from_psf(psf, 1.0)
torch.cuda.synchronize()
