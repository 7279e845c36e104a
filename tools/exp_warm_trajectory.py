"""How long does the device take to reach its steady clock under the headline loop?  ms per step of consecutive 40-step batches over
~1.5 s, with the sysfs clock / power beside every 5th batch (bench.py propagation_loop's run-in criterion comes from this)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from prysm_amd import propagation as P
x = torch.from_numpy(bench.make_field(4096, np.complex64, 4096)).cuda()
f = None
t_start = time.perf_counter()
k = 0
while time.perf_counter() - t_start < 1.5:
    t0 = time.perf_counter()
    for _ in range(40):
        f = None
        f = P.focus(x, 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40 * 1e3
    if k < 12 or k % 5 == 0:
        g = bench.gpu_state(0)
        print(f'batch {k:3d} at {(time.perf_counter() - t_start) * 1e3:7.1f} ms: {dt * 1e3:6.2f} us/step  sclk {g.get("sclk_mhz")} MHz  power {g.get("power_w")} W', flush=True)
    k += 1
