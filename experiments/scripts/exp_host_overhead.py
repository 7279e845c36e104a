"""Host-side cost of one propagation call (Python + ctypes + 2 launches) and the 4096^2 step measured three ways."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prysm_amd import propagation as P

x = torch.randn(64, 64, dtype=torch.complex64, device='cuda')
for _ in range(200): P.focus(x, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5000): f = P.focus(x, 1)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'64^2: host issue {t_issue / 5000 * 1e6:.1f} us/call, wall {t_all / 5000 * 1e6:.1f} us/call')

x = torch.randn(4096, 4096, dtype=torch.complex64, device='cuda')
for _ in range(20): f = P.focus(x, 1)
torch.cuda.synchronize()
for trial in range(3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(200):
        f = None
        f = P.focus(x, 1)
    e1.record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'4096^2: host issue {t_issue / 200 * 1e6:.1f} us/step, wall {t_all / 200 * 1e6:.1f} us/step, events {e0.elapsed_time(e1) / 200 * 1e3:.1f} us/step')
