"""The transposed Hermitian form (csrc/fft_hermt.h, knob herm_t) against the round-2 form (csrc/fft_r2c.h) and numpy: parity at
several shapes / rotations / epilogues, then HIP-event times of mtf_from_psf and the plain real-input spectrum per size.

    python tools/exp_herm_t.py [quick]
"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from prysm_amd import _lib as L
from prysm_amd import _ops, otf



def ev_ms(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + 0.04
    while time.perf_counter() < t_end:
        fn()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps)
    return sorted(best)[1]


EPIS = {'none': L.PM_EPI_NONE, 'abs2': L.PM_EPI_ABS2, 'abs': L.PM_EPI_ABS, 'arg': L.PM_EPI_ARG}


def ref(x, in_shift, out_shift, norm_dc, epi):
    a = np.roll(x.astype(np.float64), (-in_shift[0], -in_shift[1]), axis=(0, 1))       # logical element i read from position i + shift
    F = np.fft.fft2(a)
    if norm_dc:
        F = F / F[0, 0]
    F = np.roll(F, out_shift, axis=(0, 1))
    if epi == 'none':
        return F
    if epi == 'abs2':
        return np.abs(F) ** 2
    if epi == 'abs':
        return np.abs(F)
    return np.angle(F)



if __name__ == '__main__':
    quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
    worst = 0.0
    bad = []
    shapes = [(32, 32), (64, 128), (128, 64), (256, 256), (512, 2048), (2048, 512), (1024, 1024)] + ([] if quick else [(4096, 4096), (2048, 8192)])
    rng = np.random.default_rng(3)
    for dt, tol in ((np.float32, 3e-5), (np.float64, 1e-11)):
        for (M, N) in shapes:
            if dt == np.float64 and N > 4096:
                continue
            x = (rng.random((M, N)) + 0.05).astype(dt)
            xd = torch.from_numpy(x).cuda()
            for in_shift in ((0, 0), (M // 2, N // 2), (M // 2, 0)):
                for out_shift in ((0, 0), (M // 2, N // 2), (0, N // 2)):
                    for epi in EPIS:
                        for norm_dc in (False, True):
                            if M * N > 1 << 22 and (in_shift, out_shift) != ((M // 2, N // 2), (M // 2, N // 2)):
                                continue
                            want = ref(x, in_shift, out_shift, norm_dc, epi)
                            kw = dict(direction=-1, scale=1.0, in_shift=in_shift, out_shift=out_shift, epilogue=EPIS[epi],
                                      flags=L.PM_FLAG_REAL_INPUT | (L.PM_FLAG_NORM_DC if norm_dc else 0))
                            got = {}
                            for ht in (1, 0, 2):
                                if ht == 2 and M < 2048:
                                    continue
                                with L.tuning_local(herm_t=min(ht, 1), r2c=2, herm_t_fold=1 if ht == 2 else 0):
                                    got[ht] = _ops.fft2(xd, **kw).cpu().numpy()
                            for ht in got:
                                g = got[ht]
                                if epi == 'arg':     # angles: compare on the unit circle, where the modulus is not tiny
                                    mag = np.abs(ref(x, in_shift, out_shift, norm_dc, 'none'))
                                    ok = mag > 1e-3 * mag.max()
                                    err = float(np.max(np.abs(np.exp(1j * g[ok]) - np.exp(1j * want[ok]))))
                                    err = err * 1e-2 if dt == np.float32 else err * 1e-3      # phase of small bins amplifies rounding
                                else:
                                    err = float(np.max(np.abs(g - want)) / np.max(np.abs(want)))
                                if ht >= 1:
                                    worst = max(worst, err)
                                if err > tol:
                                    bad.append((np.dtype(dt).name, M, N, in_shift, out_shift, epi, norm_dc, ('r2c', 'herm_t', 'herm_t_fold')[ht], err))
    print(f'parity: worst relative error of the transposed form {worst:.2e}; failures: {len(bad)}')
    for b in bad[:20]:
        print('  FAIL', b)

    if not quick:
        print('times (us): size, dtype | mtf_from_psf r2c / transposed / transposed + fold | plain spectrum r2c / transposed / transposed + fold')
        for (M, N, dt) in ((4096, 4096, torch.float32), (2048, 2048, torch.float32), (1024, 1024, torch.float32), (8192, 8192, torch.float32), (2048, 8192, torch.float32), (8192, 2048, torch.float32),
                           (4096, 8192, torch.float32), (4096, 4096, torch.float64), (2048, 2048, torch.float64)):
            psf = torch.rand(M, N, dtype=dt, device='cuda') + 0.01
            row = []
            for ht, hf in ((0, 0), (1, 0), (1, 1)):
                with L.tuning_local(herm_t=ht, herm_t_fold=hf):
                    row.append(ev_ms(lambda: otf.mtf_from_psf(psf, 1.0)) * 1e3)
            kw = dict(direction=-1, scale=1.0, in_shift=(M // 2, N // 2), out_shift=(M // 2, N // 2), flags=L.PM_FLAG_REAL_INPUT)
            for ht, hf in ((0, 0), (1, 0), (1, 1)):
                with L.tuning_local(herm_t=ht, r2c=2, herm_t_fold=hf):
                    row.append(ev_ms(lambda: _ops.fft2(psf, **kw)) * 1e3)
            print(f'  {M}x{N} {str(dt)[6:]}: mtf {row[0]:.1f} / {row[1]:.1f} / {row[2]:.1f}   spectrum {row[3]:.1f} / {row[4]:.1f} / {row[5]:.1f}', flush=True)
            del psf
            torch.cuda.empty_cache()
    sys.exit(1 if bad else 0)
