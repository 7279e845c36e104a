"""BASELINE config 5's image (64 wavelengths x n^2, variant F) computed by WORLD_SIZE ranks (gloo: the ranks share GPU 0; nccl: one GPU
per rank) with a given reduce form, rank 0 writes it to argv[1] (.npy).  tools/check_2rank.py compares the 2-rank images of every
reduce form with the 1-rank image: the N > 1 path of bench.py's `polychromatic` section against the path a one-GPU box runs."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

out, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2048
world = int(os.environ.get('WORLD_SIZE', '1'))
rank = int(os.environ.get('RANK', '0'))
backend = os.environ.get('PM_TEST_BACKEND', 'gloo')
torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) if backend == 'nccl' else 0)
if world > 1:
    dist.init_process_group(backend)
from prysm_amd.polychromatic import polychromatic_psf, PsfPipeline
from prysm_amd.conf import config
config.precision = 32
ax = (torch.arange(n, device='cuda', dtype=torch.float64) - n // 2) * (10.0 / n)
r = torch.hypot(ax[None, :], ax[:, None])
amp = (r <= 5).to(torch.float32)
opd = (500.0 * (r / 5) ** 4).to(torch.float32)
wvls, wts = np.linspace(0.5, 0.7, 64), np.ones(64)
imgs = {}
for m in (('reduce', 'a2a', 'rs') if world > 1 else ('reduce',)):
    img = polychromatic_psf(amp, opd, wvls, wts, 10.0 / n, 100.0, Q=1, reduce_to_all=False, reduce_method=m)
    if rank == 0:
        imgs[m] = img.cpu().numpy()
pipe = PsfPipeline(wvls, wts, 10.0 / n, 100.0, Q=1, reduce_to_all=False, reduce_method='a2a' if world > 1 else 'reduce', depth=2, cache_pupil=True)
pend = [pipe.submit(amp, opd) for _ in range(3)]
res = [p.result() for p in pend]
pipe.drain()
if rank == 0:
    imgs['pipelined_last'] = res[-1].cpu().numpy()
    np.savez(out, **imgs)
    print(f'poly_image: world {world} n {n} forms {sorted(imgs)} -> {out}', flush=True)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
