"""model_7plane_1024 (complex64, eager, us per wavelength) under the GEMM planner's knobs: workgroups the split of K aims for
(gemm_dma_wgs), K-groups inside a workgroup (gemm_wk), tile edge (gemm_tile)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from prysm_amd import propagation as P, _lib as L
from prysm_amd.conf import config
config.precision = 32
inp = bench.model7_inputs()
dev = {k: torch.from_numpy(v).to(torch.float32).cuda() for k, v in inp.items() if k != 'dx'}
w = 0.6
fdx, ddx = bench.model7_grids(w)
fpm = torch.from_numpy(bench.model7_fpm(w)).to(torch.float32).cuda()
exa = P.prepare_executor(inp['dx'], 1024, fdx, 256, w, 1000.0)
exb = P.prepare_executor(inp['dx'], 1024, ddx, 256, w, 1000.0)
one = lambda: bench.model7(P, dev['amp'], dev['opd'], dev['dm'], fpm, dev['lyot'], w, inp['dx'], exa, exb)
ref = one().clone()
print('default', f'{bench._event_ms(one, 50) * 1e3:.1f} us')
for kn in (dict(gemm_dma_wgs=128), dict(gemm_dma_wgs=256), dict(gemm_dma_wgs=1024), dict(gemm_wk=0), dict(gemm_tile=64), dict(gemm_dma_wgs=256, gemm_wk=0),
           dict(gemm_dma=0)):
    with L.tuning_local(**kn):
        t = bench._event_ms(one, 50) * 1e3
        err = float((one() - ref).abs().max() / ref.abs().max())
    print(kn, f'{t:.1f} us', f'rel diff {err:.1e}', flush=True)
