"""mtf_from_psf (and the plain spectrum) per size: the round-2 Hermitian route against the transposed one (fold off / on), to derive the
planner's auto rule"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from prysm_amd import _lib as L, _ops, otf
from exp_herm_t import ev_ms
for dt in (torch.float32, torch.float64):
    for (M, N) in ((128, 128), (256, 256), (512, 512), (1024, 1024), (1024, 4096), (4096, 1024), (2048, 2048), (2048, 4096), (4096, 2048), (4096, 4096), (2048, 8192), (8192, 2048), (4096, 8192), (8192, 4096)):
        if dt == torch.float64 and N > 4096:
            continue
        psf = torch.rand(M, N, dtype=dt, device='cuda') + 0.01
        row = []
        for ht, hf in ((0, 0), (1, 0), (1, 1)):
            if hf and M < 2048 or (M == 8192 and ht and not hf):
                row.append(float('nan'))
                continue
            with L.tuning_local(herm_t=ht, herm_t_fold=hf):
                row.append(ev_ms(lambda: otf.mtf_from_psf(psf, 1.0)) * 1e3)
        print(f'{M}x{N} {str(dt)[6:]}: mtf r2c {row[0]:.1f}  transposed {row[1]:.1f}  transposed+fold {row[2]:.1f}', flush=True)
        del psf
        torch.cuda.empty_cache()
