"""config 2 (2048^2 complex64 focus) with TWO units per workgroup in the row pass, the column pass, or both
(experiment build: PRYSM_AMD_LIB=prysm_amd/alt/libprysm_amd.so), interleaved with the shipped form.  Also per pass (PASS1 / PASS2 only)."""
import ctypes
import numpy as np
import torch
from prysm_amd import _lib as L, _ops
from prysm_amd import propagation as P

lib = L.load()
rng = np.random.default_rng(2048)
x = torch.from_numpy((rng.standard_normal((2048, 2048)) + 1j * rng.standard_normal((2048, 2048))).astype(np.complex64)).cuda()
ref = P.focus(x, 1).clone()


def timed(fn, reps=200):
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
    d.dtype = L.code(x); d.direction = -1; d.scale = 1.0 / 2048; d.weight = 1.0
    d.in_y = d.in_x = d.out_y = d.out_x = _ops._axis(2048, 2048, 0, 1024)
    d.in_ld = d.out_ld = 2048
    out = torch.empty_like(x)
    nb = lib.pm_fft2_workspace(ctypes.byref(d))
    ws = L.workspace(nb)
    ms = (ctypes.c_double * 2)()
    L.check(lib.pm_fft2_time_passes(ctypes.byref(d), L.ptr(x), L.ptr(out), L.ptr(ws), ws.numel(), 200, ms, L.stream_ptr()))
    torch.cuda.synchronize()
    return ms[0] * 1e3, ms[1] * 1e3


for rnd in range(3):
    for tu in (0, 1, 2, 3):
        if lib.pm_set_tuning(b'two_units', tu) != 0:
            raise SystemExit('needs the experiment build (PRYSM_AMD_LIB=prysm_amd/alt/libprysm_amd.so)')
        t = timed(lambda: P.focus(x, 1))
        err = float((P.focus(x, 1) - ref).abs().max())
        r, c = passes()
        print(f'two_units={tu} (rows {tu & 1}, columns {(tu >> 1) & 1}): {t:6.2f} us per propagation; row pass {r:5.2f} us, column pass {c:5.2f} us; max diff to the shipped form {err:.1e}', flush=True)
lib.pm_set_tuning(b'two_units', 0)
