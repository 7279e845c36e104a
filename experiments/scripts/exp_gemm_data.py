"""does the f32 MFMA rate depend on the DATA?  the config-4 large product and 4096^3 on random / constant / zero operands"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from prysm_amd import _ops
def t(fn, reps=20):
    for _ in range(3): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
torch.manual_seed(0)
for name, mk in (('randn', lambda *s: torch.randn(*s, dtype=torch.complex64, device='cuda')),
                 ('ones', lambda *s: torch.ones(*s, dtype=torch.complex64, device='cuda')),
                 ('zeros', lambda *s: torch.zeros(*s, dtype=torch.complex64, device='cuda')),
                 ('small-int', lambda *s: torch.complex(torch.randint(-2, 3, s, device='cuda').float(), torch.randint(-2, 3, s, device='cuda').float()))):
    A = mk(512, 2048); B = mk(2048, 2048)
    us = t(lambda: _ops.cgemm(A, B))
    A4 = mk(4096, 4096); B4 = mk(4096, 4096)
    us4 = t(lambda: _ops.cgemm(A4, B4), 5)
    print(f'{name:10s} 512x2048x2048: {us:7.1f} us ({8*512*2048*2048/us/1e6:6.1f} TF alg)   4096^3: {us4:8.1f} us ({8*4096**3/us4/1e6:6.1f} TF alg)')
