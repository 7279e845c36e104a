"""Config 5 variant F (64 wavelengths x 4096^2): the loop of transform pairs on ONE stream against the same kernels on TWO streams -- the
row pass of wavelength k + 1 (stream A, intermediate k + 1 mod 2) runs beside the accumulating column pass of wavelength k (stream B):
the column passes stay in wavelength order (same sums, bit for bit), each kernel's tail overlaps the other's head."""
import ctypes
import math
import sys

import numpy as np
import torch

from prysm_amd import _lib as L, _ops
from prysm_amd.propagation import focus_intensity

lib = L.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nl = 64
g = torch.Generator(device='cuda').manual_seed(1)
ax = (torch.arange(n, device='cuda', dtype=torch.float64) - n // 2) * (10.0 / n)
r = torch.hypot(ax[None, :], ax[:, None])
amp = (r <= 5).to(torch.float32)
opd = (500.0 * (r / 5) ** 4).to(torch.float32)
packed = _ops.pack_amp_opd(amp, opd)
wl = np.linspace(0.5, 0.7, nl)
ks = [2 * math.pi / w / 1e3 for w in wl]
wts = [1.0] * nl


def loop(acc):
    focus_intensity(packed, 1, out=acc, synth=('packed', ks[0]), spectral=(ks, wts))


acc0 = torch.zeros((n, n), device='cuda', dtype=torch.float32)
loop(acc0)
torch.cuda.synchronize()
# the descriptor focus_intensity built for (packed, Q = 1, accumulate): the last plan entry
focus_intensity(packed, 1, out=torch.zeros_like(acc0), weight=1.0, synth=('packed', ks[0]))
d0 = list(_ops._fft2_plans.values())[-1][0]
nbytes = lib.pm_fft2_workspace(ctypes.byref(d0))
ws = [torch.empty(int(nbytes), dtype=torch.uint8, device='cuda') for _ in range(2)]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def overlapped(acc):
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur)
    sb.wait_stream(cur)
    ev_row = [torch.cuda.Event() for _ in range(nl)]
    ev_col = [torch.cuda.Event() for _ in range(nl)]
    for k in range(nl):
        d = L.pm_fft2_desc.from_buffer_copy(d0)
        d.synth_k = ks[k]
        d.weight = wts[k]
        d.flags = d0.flags | L.PM_FLAG_PASS1_ONLY
        if k >= 2:
            sa.wait_event(ev_col[k - 2])            # the intermediate of wavelength k - 2 has been read
        L.check(lib.pm_fft2(ctypes.byref(d), packed.data_ptr(), acc.data_ptr(), ws[k & 1].data_ptr(), nbytes, ctypes.c_void_p(sa.cuda_stream)))
        ev_row[k].record(sa)
        d.flags = d0.flags | L.PM_FLAG_PASS2_ONLY
        sb.wait_event(ev_row[k])
        L.check(lib.pm_fft2(ctypes.byref(d), packed.data_ptr(), acc.data_ptr(), ws[k & 1].data_ptr(), nbytes, ctypes.c_void_p(sb.cuda_stream)))
        ev_col[k].record(sb)
    cur.wait_stream(sb)
    cur.wait_stream(sa)


acc1 = torch.zeros_like(acc0)
overlapped(acc1)
torch.cuda.synchronize()
print('max rel diff overlapped vs loop: %.2e' % float((acc1 - acc0).abs().max() / acc0.abs().max()), flush=True)


def timed(fn, reps=5):
    acc = torch.zeros_like(acc0)
    fn(acc)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn(acc)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for rnd in range(3):
    t0, t1 = timed(loop), timed(overlapped)
    print(f'n={n}: loop {t0 * 1e3 / nl:6.1f} us per wavelength ({t0:.3f} ms)   two streams {t1 * 1e3 / nl:6.1f} us per wavelength ({t1:.3f} ms)', flush=True)
