"""rocprofv3 target: mtf_from_psf / plain spectrum of a real field on one route (argv: M N dtype herm_t [epi])"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prysm_amd import _lib as L, _ops, otf
M, N, dt, ht = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
epi = sys.argv[5] if len(sys.argv) > 5 else 'mtf'
extra = dict(kv.split('=') for kv in sys.argv[6:])
psf = torch.rand(M, N, dtype=torch.float32 if dt == 'f32' else torch.float64, device='cuda') + 0.01
with L.tuning_local(herm_t=ht, r2c=2, **{k: int(v) for k, v in extra.items()}):
    for _ in range(60):
        if epi == 'mtf':
            otf.mtf_from_psf(psf, 1.0)
        else:
            _ops.fft2(psf, direction=-1, scale=1.0, in_shift=(M // 2, N // 2), out_shift=(M // 2, N // 2), flags=L.PM_FLAG_REAL_INPUT)
torch.cuda.synchronize()
