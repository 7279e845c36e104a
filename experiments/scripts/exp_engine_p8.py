"""The radix-8 engine (8 points per thread, twice the waves, three exchanges) against the shipped radix-16 engine on the headline:
focus of a 4096^2 complex64 field, per pass (experiment build: PRYSM_AMD_LIB=prysm_amd/alt/libprysm_amd.so)."""
import ctypes
import numpy as np
import torch
from prysm_amd import _lib as L, _ops
from prysm_amd import propagation as P

lib = L.load()
n = 4096
rng = np.random.default_rng(n)
x = torch.from_numpy((rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))).astype(np.complex64)).cuda()
ref = P.focus(x, 1).clone()


def timed(fn, reps=100):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best


def passes():
    d = L.pm_fft2_desc()
    d.dtype = L.code(x); d.direction = -1; d.scale = 1.0 / n; d.weight = 1.0
    d.in_y = d.in_x = d.out_y = d.out_x = _ops._axis(n, n, 0, n // 2)
    d.in_ld = d.out_ld = n
    out = torch.empty_like(x)
    nb = lib.pm_fft2_workspace(ctypes.byref(d))
    ws = L.workspace(nb)
    ms = (ctypes.c_double * 2)()
    L.check(lib.pm_fft2_time_passes(ctypes.byref(d), L.ptr(x), L.ptr(out), L.ptr(ws), ws.numel(), 100, ms, L.stream_ptr()))
    torch.cuda.synchronize()
    return ms[0] * 1e3, ms[1] * 1e3


for rnd in range(3):
    for knob in (0, 1, 2, 3, 5, 6, 7):
        if lib.pm_set_tuning(b'engine_p8', knob) != 0:
            raise SystemExit('needs the experiment build (PRYSM_AMD_LIB=prysm_amd/alt/libprysm_amd.so)')
        t = timed(lambda: P.focus(x, 1))
        got = P.focus(x, 1)
        err = float((got - ref).abs().max() / ref.abs().max())
        r, c = passes()
        print(f'engine_p8={knob} (rows {knob & 1}, columns {(knob >> 1) & 1}, 64-register cap {(knob >> 2) & 1}): {t:6.2f} us per propagation; '
              f'row pass {r:5.2f} us, column pass {c:5.2f} us; max rel diff to the radix-16 engine {err:.1e}', flush=True)
lib.pm_set_tuning(b'engine_p8', 0)
