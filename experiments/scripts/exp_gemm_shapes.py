"""config-4 / config-5 GEMM shapes under the tile / split knobs of the LDS-DMA kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prysm_amd import _ops, _lib
lib = _lib.load()
def t(fn, reps=30):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
torch.manual_seed(0)
mk = lambda *s: torch.randn(*s, dtype=torch.complex64, device='cuda')
Ey, ary, Ex = mk(512, 2048), mk(2048, 2048), mk(512, 2048)
Ey5, ary5, Ex5 = mk(512, 4096), mk(4096, 4096), mk(512, 4096)
for tile in (64, 128):
    for wgs in (256, 512, 1024):
        lib.pm_set_tuning(b'gemm_tile', tile); lib.pm_set_tuning(b'gemm_dma_wgs', wgs)
        g1 = t(lambda: _ops.cgemm(Ey, ary)); T = _ops.cgemm(Ey, ary)
        g2 = t(lambda: _ops.cgemm(T, Ex, 0, 2))
        pair = t(lambda: _ops.cgemm(_ops.cgemm(Ey, ary), Ex, 0, 2))
        g15 = t(lambda: _ops.cgemm(Ey5, ary5), 10); T5 = _ops.cgemm(Ey5, ary5)
        g25 = t(lambda: _ops.cgemm(T5, Ex5, 0, 2), 10)
        print(f'tile {tile:5d} wgs {wgs:4d}: config4 G1 {g1:6.1f} G2 {g2:5.1f} pair {pair:6.1f} us ({21.47e3/pair:5.1f} TF) | config5 G1 {g15:6.1f} G2 {g25:5.1f} ({77.3e3/(g15+g25):5.1f} TF)')
