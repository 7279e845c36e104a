"""mtf_from_psf 4096^2 / 2048^2 fp32 on the transposed form with the start-up stagger forced on its single-round launches
(fft_stagger_r2c: pass B, fft_stagger_herm: pass A; 100 + units forces a single round)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from prysm_amd import _lib as L, otf
from exp_herm_t import ev_ms
for n in (4096, 2048):
    psf = torch.rand(n, n, dtype=torch.float32, device='cuda') + 0.01
    base = ev_ms(lambda: otf.mtf_from_psf(psf, 1.0)) * 1e3
    print(n, 'default', f'{base:.1f}', flush=True)
    for knob in ('fft_stagger_r2c', 'fft_stagger_herm'):
        row = []
        for v in (101, 102, 104, 108):
            with L.tuning_local(**{knob: v}):
                row.append(ev_ms(lambda: otf.mtf_from_psf(psf, 1.0)) * 1e3)
        print(n, knob, '101 102 104 108:', ' '.join(f'{t:.1f}' for t in row), flush=True)
