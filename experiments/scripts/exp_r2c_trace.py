import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prysm_amd import otf, _lib
lib = _lib.load()
for n, dt in ((4096, torch.float32), (2048, torch.float32)):
    psf = torch.rand(n, n, dtype=dt, device='cuda') + 0.01
    for _ in range(20):
        otf.mtf_from_psf(psf, 1.0)
    lib.pm_set_tuning(b'r2c', 2)
    for _ in range(20):
        otf.transform_psf(psf, 1.0)
    lib.pm_set_tuning(b'r2c', 1)
torch.cuda.synchronize()
