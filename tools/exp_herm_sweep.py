"""mtf_from_psf on the transposed Hermitian route: sweep of the column pass's XCD grouping (col_log_g), the fold and the row tiling"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prysm_amd import _lib as L, otf
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from exp_herm_t import ev_ms
for (M, N, dt) in ((4096, 4096, torch.float32), (2048, 2048, torch.float32), (4096, 4096, torch.float64), (8192, 8192, torch.float32)):
    psf = torch.rand(M, N, dtype=dt, device='cuda') + 0.01
    with L.tuning_local(herm_t=0):
        base = ev_ms(lambda: otf.mtf_from_psf(psf, 1.0)) * 1e3
    print(f'{M}x{N} {str(dt)[6:]}: r2c route {base:.1f} us', flush=True)
    for fold in (0, 1):
        if M == 8192 and not fold:
            continue
        for rv in ((4, 0) if (dt == torch.float32 and N == 4096) else (-1,)):
            row = []
            for lg in (0, 1, 2, 3, 4, 5):
                with L.tuning_local(herm_t=1, herm_t_fold=fold, col_log_g=lg, herm_t_rowvar=rv):
                    row.append(ev_ms(lambda: otf.mtf_from_psf(psf, 1.0)) * 1e3)
            print(f'   fold {fold} rowvar {rv}: col_log_g 0..5: ' + ' '.join(f'{t:.1f}' for t in row), flush=True)
    del psf
    torch.cuda.empty_cache()
