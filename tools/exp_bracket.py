"""Where do the one-off microseconds of a K-step timed region go?  (bench.py Ranks.timed: synchronize, K calls, synchronize.)  Host
timestamps after every call, a HIP event before the first and after the last kernel, for K = 20."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from prysm_amd import propagation as P
x = torch.from_numpy(bench.make_field(4096, np.complex64, 4096)).cuda()
f = None
t_end = time.perf_counter() + 0.6
while time.perf_counter() < t_end:          # run in
    for _ in range(40):
        f = None
        f = P.focus(x, 1)
    torch.cuda.synchronize()
K = 20
for trial in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    ts = []
    for _ in range(K):
        f = None
        f = P.focus(x, 1)
        ts.append(time.perf_counter())
    e1.record()
    t_issue = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    gpu = e0.elapsed_time(e1) * 1e3
    print(f'trial {trial}: wall {(t1 - t0) * 1e6:7.1f} us = {(t1 - t0) * 1e6 / K:.2f}/step | events {gpu:7.1f} us = {gpu / K:.2f}/step | first call returned at {(ts[0] - t0) * 1e6:.1f} us, '
          f'last issued at {(t_issue - t0) * 1e6:.1f} us, per call {np.mean(np.diff(ts)) * 1e6:.1f} us | wall - events {(t1 - t0) * 1e6 - gpu:.1f} us', flush=True)
