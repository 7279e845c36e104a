"""Bluestein path vs the direct O(n^2) kernel on non-power-of-two sizes (one process, knob blue_min toggled):
python tools/exp_bluestein.py  -> time per 2-D transform, both ways, and the accuracy of each against numpy."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prysm_amd import _lib as L, _ops  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    lib = L.load()
    rng = np.random.default_rng(3)
    for n, dt in [(96, np.complex64), (100, np.complex64), (130, np.complex64), (200, np.complex64), (260, np.complex64), (384, np.complex64),
                  (520, np.complex64), (1000, np.complex64), (1000, np.complex128), (1536, np.complex64), (2000, np.complex64),
                  (3000, np.complex64), (3000, np.complex128), (4000, np.complex64)]:
        x = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))).astype(dt)
        xd = torch.from_numpy(x).cuda()
        want = np.fft.fft2(x.astype(np.complex128))
        res = {}
        for name, lo in (('bluestein', 1), ('unfused', 1), ('peraxis', 1), ('direct', 0)):
            lib.pm_set_tuning(b'blue_min', lo)
            lib.pm_set_tuning(b'blue_2d', 0 if name == 'peraxis' else 1)
            lib.pm_set_tuning(b'blue_fuse', 0 if name == 'unfused' else 1)
            f = lambda: _ops.fft2(xd, direction=-1, scale=1.0)
            if name == 'direct' and n > 2000:
                res[name] = (float('nan'), float('nan'))
                continue
            err = np.abs(f().cpu().numpy() - want).max() / np.abs(want).max()
            res[name] = (timeit(f, 20 if name != 'direct' or n <= 520 else 3), err)
        lib.pm_set_tuning(b'blue_min', 96)
        lib.pm_set_tuning(b'blue_2d', 1)
        lib.pm_set_tuning(b'blue_fuse', 1)
        print('EXP n=%5d %-10s bluestein 2-D %8.1f us (err %.1e)   chirps as separate kernels %8.1f us   per axis %8.1f us (err %.1e)   direct %10.1f us (err %.1e)' %
              (n, np.dtype(dt).name, res['bluestein'][0], res['bluestein'][1], res['unfused'][0], res['peraxis'][0], res['peraxis'][1],
               res['direct'][0], res['direct'][1]), flush=True)


if __name__ == '__main__':
    main()
