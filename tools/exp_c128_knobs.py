"""config 3 (angular spectrum 4096^2 complex128) and focus 4096^2 complex128 under the planner's knobs: re-check of the defaults"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from prysm_amd import propagation as P, _lib as L
x = torch.from_numpy(bench.make_field(4096, np.complex128, 4096)).cuda()
cases = [dict(), dict(fold=0), dict(log_k=1), dict(log_k=2), dict(log_k=3), dict(log_k=4), dict(col_var=0), dict(nt_in=0), dict(nt_in=1), dict(nt_out=1), dict(nt_in=1, nt_out=1),
         dict(fft_stagger=0), dict(fft_stagger=4), dict(fft_stagger_col=0), dict(fft_stagger_col=4), dict(fft_stagger_col=16), dict(fft_stagger_mid=4), dict(fft_stagger_mid=8), dict(stagger_group=1),
         dict(col_log_g=0), dict(col_log_g=2), dict(col_log_g=3), dict(row_log_g=0), dict(row_log_g=2), dict(colmul_mode=0)]
for kn in cases:
    try:
        with L.tuning_local(**kn):
            a = bench._event_ms(lambda: P.angular_spectrum(x, 0.6328, 0.01, 10.0, Q=1), 20) * 1e3
            f = bench._event_ms(lambda: P.focus(x, 1), 20) * 1e3
        print(kn, f'angular spectrum {a:.1f} us   focus {f:.1f} us', flush=True)
    except Exception as exc:
        print(kn, 'refused:', str(exc)[:80], flush=True)
