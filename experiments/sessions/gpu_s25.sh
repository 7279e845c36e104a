#!/bin/bash
# round 4, session 9: full GPU suite on the final sources (radices 17 / 19), a long fuzz run over the round-4 routes, 17/19 timings
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s25; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > $O/pytest_gpu.log 2>&1
( timeout 1500 python tools/fuzz_fft2.py 700 404 2>&1 | tail -40 ) > $O/fuzz_700_seed404.log 2>&1
( timeout 300 python - <<'PY'
import torch, numpy as np
from prysm_amd import _ops, _lib
lib = _lib.load()
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
for n in (1020, 1900, 2040, 3230, 4913):
    for dt in (torch.complex64, torch.complex128):
        x = torch.randn(n, n, dtype=dt, device='cuda')
        res = []
        for mix in (1, 0):
            lib.pm_set_tuning(b'mix', mix)
            res.append(timed(lambda: _ops.fft2(x, direction=-1, scale=1.0)))
        lib.pm_set_tuning(b'mix', 1)
        print(f'n={n} {"c64 " if dt == torch.complex64 else "c128"}: mixed radix (17 / 19 as radices) {res[0]:8.1f} us   Bluestein {res[1]:8.1f} us', flush=True)
PY
) > $O/exp_primes_17_19.log 2>&1
tail -4 $O/pytest_gpu.log; tail -12 $O/fuzz_700_seed404.log; cat $O/exp_primes_17_19.log
