"""16384^2 complex64 focus (radix-2 step around 8192-point engine transforms) and 5120^2 (radix 5): per-kernel times under rocprofv3."""
import sys

import torch

from prysm_amd import propagation as P

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
x = torch.randn(n, n, dtype=torch.complex64, device='cuda')
for _ in range(4):
    y = P.focus(x, 1)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(3):
    y = P.focus(x, 1)
b.record()
torch.cuda.synchronize()
print(f'n={n}: {a.elapsed_time(b) / 3 * 1e3:.0f} us per focus')
