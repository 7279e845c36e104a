"""Checks of the N > 1 rehearsal (tools/gpu_multi_rank.sh): the 2-rank bench line carries the keys the driver's SCALE run will be read
by, and the 2-rank polychromatic images (every reduce form, and pipelined frames) equal the 1-rank image.  Exit code 1 on any miss."""
import json
import sys

import numpy as np

line_path, img1, img2 = sys.argv[1:4]
bad = []
line = None
for ln in open(line_path):
    ln = ln.strip()
    if ln.startswith('{') and '"metric"' in ln:
        line = json.loads(ln)
if line is None:
    print('check_2rank: no JSON line in', line_path)
    sys.exit(1)


def get(path):
    d = line
    for k in path.split('/'):
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


for key in ('value', 'ms_per_step', 'roofline/frac', 'n2048/value', 'polychromatic/variant_F_fft_focus/psf_ms',
            'polychromatic/variant_F_fft_focus/psf_ms_by_reduce_method/a2a', 'polychromatic/variant_F_fft_focus/psf_ms_by_reduce_method/rs',
            'polychromatic/variant_F_fft_focus/pipelined_ms_per_psf', 'polychromatic/variant_M_mdft_512/psf_ms',
            'polychromatic/reduce_alone_ms/reduce', 'polychromatic/reduce_alone_ms/a2a', 'polychromatic/reduce_alone_ms/rs',
            'polychromatic/scaling_model/measured_this_run/n_gpus', 'polychromatic_2048/spectral_groups/psf_ms', 'summary/c5F_psf_ms'):
    if get(key) is None:
        bad.append(f'missing {key}')
if get('n_gpus') != 2 or get('polychromatic/scaling_model/measured_this_run/n_gpus') != 2:
    bad.append(f"n_gpus {get('n_gpus')} / measured_this_run {get('polychromatic/scaling_model/measured_this_run/n_gpus')} (want 2)")
if get('extras_error') or get('extras'):
    bad.append(f"side measurements did not finish: {get('extras_error') or get('extras')}")
a, b = np.load(img1), np.load(img2)
ref = a['reduce'].astype(np.float64)
worst = {}
for k in b.files:
    worst[k] = float(np.max(np.abs(b[k].astype(np.float64) - ref)) / np.max(np.abs(ref)))
    if worst[k] > 2e-6:         # fp32 partial sums in a different order (two blocks of 32 wavelengths against one of 64)
        bad.append(f'2-rank image ({k}) differs from the 1-rank image: {worst[k]:.2e}')
w1 = float(np.max(np.abs(a['pipelined_last'].astype(np.float64) - ref)) / np.max(np.abs(ref)))
print(json.dumps({'check_2rank': 'FAILED' if bad else 'ok', 'problems': bad, 'value_2rank': get('value'), 'c5F_psf_ms': get('polychromatic/variant_F_fft_focus/psf_ms'),
                  'c5F_psf_ms_by_reduce_method': get('polychromatic/variant_F_fft_focus/psf_ms_by_reduce_method'),
                  'reduce_alone_ms': get('polychromatic/reduce_alone_ms'), 'image_rel_err_vs_1rank': worst, 'pipelined_1rank_rel_err': w1}))
sys.exit(1 if bad else 0)
