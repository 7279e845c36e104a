"""cProfile of the host side of 20000 focus(x, 1) calls on a 64 x 64 field (GPU work negligible): where the ~21 us per call go."""
import cProfile
import pstats

import torch

from prysm_amd import propagation as P

x = torch.randn(64, 64, dtype=torch.complex64, device='cuda')
for _ in range(500):
    P.focus(x, 1)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20000):
    P.focus(x, 1)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(18)
