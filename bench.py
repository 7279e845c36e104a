#!/usr/bin/env python
"""Headline benchmark: pupil -> focus FFT propagations per second at 4096^2 complex64.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs N ranks, one process per GPU, over RCCL (torch.distributed backend "nccl").  Either the caller launches
them (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
--gpus N ...: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are then in the environment), or -- when WORLD_SIZE is not set --
bench.py re-executes itself under torch.distributed.run with exactly that command line.  N = 1 is the same code path
without a process group.

A "step" is ONE propagation ``focus(x, Q=1)`` = fftshift(fft2(ifftshift(x), norm='ortho')) of a synthetic
4096 x 4096 complex64 field already resident in HBM (BASELINE.json metric; config "4096^2 pupil->focus").  The path
shards over wavelengths / fields with no data-path collective (weak scaling): every rank propagates its own field, the
timed region is exactly K propagations per rank between barrier + synchronize pairs, MAX over ranks.

After the timed region, outside of it and on every rank:
* ``n2048``: the same measurement at 2048^2 (the second size the north star names);
* ``polychromatic``: BASELINE config 5 as ONE timed call of prysm_amd.polychromatic.polychromatic_psf -- 64
  wavelengths x 4096^2 fp32 sharded over the N ranks, |.|^2 accumulated per rank, one sum-reduce of the image to rank 0
  over RCCL -- variant F (FFT focus, Q = 1) and variant M (matrix-DFT focus 4096^2 -> 512^2 on MFMA), plus the
  reduce of the 67 MB fp32 image on its own in both root-only forms (``reduce_alone_ms``: one torch.distributed.reduce, and
  all-to-all of slices + ordered local sum + gather); variant F with either form (``--reduce-method auto`` reports the
  faster) and as a pipelined sequence of PSFs (PsfPipeline: frame k's reduce on a side stream under frame k + 1's
  transforms); ``scaling_model``: what N = 2, 4, 8 should do, from this run's per-wavelength time.

The headline line is complete before any of that starts; if the side measurements hang (their collectives meet RCCL on
an N > 1 node for the first time in the driver's runs) a watchdog prints it without them after --extras-budget seconds.

Rank 0 prints ONE JSON line.  ``roofline``: the dominant kernel (the slower of the two FFT passes), its duration
measured with HIP events recorded between the kernels on the launch stream, in sequence; achieved = 2 N^2 s
algorithmic bytes (read + write of one pass) / duration, peak 8 TB/s; ``traffic`` = PMC bytes per launch from the
committed rocprofv3 summary, printed only while the kernel sources still hash to the fingerprint the summary was
collected at.  ``cpu_baseline``: the CPU oracle (numpy / scipy restatement of prysm's focus, scipy.fft single-threaded
exactly as prysm ships it) timed on this box's host cores on the same workload (N = 1 only).
"""
import argparse
import ctypes
import glob
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_EDGE = 4096
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)
HBM_COPY_CEILING_GBS = 6290.0   # measured float4-copy ceiling quoted by the same guide (report both fractions)
F32_MFMA_PEAK_TF = 157.3
F64_MFMA_PEAK_TF = 78.6    # same guide: FP64 matrix (v_mfma_f64_16x16x4_f64)
N_WAVELENGTHS = 64      # BASELINE config 5


def _slim(o):
    """the line without its prose: every 'note' / 'workload' string below the top level (they are documented once, in DESIGN.md 5
    "Keys of the bench line") and floats at six significant digits -- the driver keeps the last 2000 characters of the output"""
    if isinstance(o, dict):
        return {k: (v if k == 'config' else _slim(v)) for k, v in o.items() if k not in ('note', 'workload') or not isinstance(v, str)}
    if isinstance(o, list):
        return [_slim(v) for v in o]
    if isinstance(o, float):
        return float(f'{o:.6g}')
    return o


def _summary(line):
    """The numbers a reader of the stored tail needs, as the LAST key of the line (flat, short names, ms unless named otherwise)."""
    def get(path, d=line):
        for k in path.split('/'):
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return float(f'{d:.4g}') if isinstance(d, float) else d
    oc = 'other_configs/'
    pairs = {
        'psfs_per_s': get('value'), 'ms': get('ms_per_step'), 'repeat_ms': get('repeat_ms'), 'repeat_spread': get('repeat_spread'), 'prewarm_ms': get('prewarm_ms'),
        'sclk_before': get('gpu_before/sclk_mhz'), 'sclk_after': get('gpu_after/sclk_mhz'), 'power_w_after': get('gpu_after/power_w'), 'row_ms': get('roofline/row_pass_ms'), 'col_ms': get('roofline/column_pass_ms'),
        'frac': get('roofline/frac'), 'two_copies_ms': get('roofline/two_plain_copies_ms'), 'n2048_per_s': get('n2048/value'),
        'psf_variant_ms': get('psf_variant/ms_per_psf'),
        'm7_c64_eager_ms': get(oc + 'model_7plane_1024/c64/eager_ms_per_wavelength'), 'm7_c64_graph_ms': get(oc + 'model_7plane_1024/c64/graph_ms_per_wavelength'),
        'm7_c64_9wvl_graph_ms': get(oc + 'model_7plane_1024/c64/graph_9wvl_ms'), 'm7_c64_9wvl_graph_branches_ms': get(oc + 'model_7plane_1024/c64/graph_branches_9wvl_ms'),
        'm7_c64_czt_graph_ms': get(oc + 'model_7plane_1024/c64/czt/graph_ms_per_wavelength'), 'm7_c64_czt_9wvl_graph_branches_ms': get(oc + 'model_7plane_1024/c64/czt/graph_branches_9wvl_ms'),
        'm7_c128_eager_ms': get(oc + 'model_7plane_1024/c128/eager_ms_per_wavelength'), 'm7_c128_graph_ms': get(oc + 'model_7plane_1024/c128/graph_ms_per_wavelength'),
        'm7_published_titan_xp_ms': get(oc + 'model_7plane_1024/published/titan_xp_cupy_ms_per_wavelength'),
        'c2_ms': get(oc + 'config2_focus_2048_c64/ms'), 'c2_two_streams_ms': get(oc + 'config2_focus_2048_c64/two_streams/ms'),
        'c2_sequence_ms': get(oc + 'config2_focus_2048_c64/sequence_block/ms'), 'f1000_sequence_ms': get(oc + 'focus_1000_c64_sequence_block/ms'),
        'c2_loop_ms': get(oc + 'config2_focus_2048_c64/sequence_block/loop_ms'), 'f1000_loop_ms': get(oc + 'focus_1000_c64_sequence_block/loop_ms'),
        'c128_2048_loop_ms': get(oc + 'focus_2048_c128_sequence_block/loop_ms'), 'c128_2048_sequence_ms': get(oc + 'focus_2048_c128_sequence_block/ms'),
        'c3_ms': get(oc + 'config3_angular_spectrum_4096_c128/ms'), 'c3_moved_frac': get(oc + 'config3_angular_spectrum_4096_c128/moved_frac_of_hbm_peak'),
        'c4_ms': get(oc + 'config4_mdft_2048_to_512_c64/ms'), 'c4_build_ms': get(oc + 'config4_mdft_2048_to_512_c64/prepare_executor_ms'),
        'c4_frac_mfma': get(oc + 'config4_mdft_2048_to_512_c64/frac_of_f32_mfma_peak'), 'c4_mfma_busy': get(oc + 'config4_mdft_2048_to_512_c64/mfma_busy'),
        'c128_4096_ms': get(oc + 'focus_4096_c128/ms'), 'c64_8192_ms': get(oc + 'focus_8192_c64/ms'),
        'mtf_4096_ms': get(oc + 'mtf_from_psf_4096_f32/ms'), 'mtf_3000_ms': get(oc + 'mtf_from_psf_3000_f32/ms'),
        'mtf_3000_composed_ms': get(oc + 'mtf_from_psf_3000_f32/composed_ms'), 'conv_4096_ms': get(oc + 'conv_real_4096_f32/ms'),
        'f3000_c64_ms': get(oc + 'focus_3000_c64_mixed_radix/ms'), 'f3000_c128_ms': get(oc + 'focus_3000_c128_mixed_radix/ms'),
        'f1000_c64_ms': get(oc + 'focus_1000_c64_mixed_radix/ms'), 'f3000_c64_general_ms': get(oc + 'focus_3000_c64_mixed_radix/general_kernel_ms'),
        'f1536_c64_ms': get(oc + 'focus_1536_c64_composite/ms'), 'as3000_c64_ms': get(oc + 'angular_spectrum_3000_c64_composite/ms'),
        'as3000_c128_ms': get(oc + 'angular_spectrum_3000_c128_composite/ms'),
        'stack16x500_ms_per_field': get(oc + 'focus_stack_16x500_c64_composite/ms_per_field'),
        'stack16x500_loop_ms_per_field': get(oc + 'focus_stack_16x500_c64_composite/loop_ms_per_field'),
        'c5F_psf_ms': get('polychromatic/variant_F_fft_focus/psf_ms'), 'c5F_ms_per_wvl': get('polychromatic/variant_F_fft_focus/per_wavelength_ms_per_gpu'),
        'c5F_psf_ms_by_reduce': get('polychromatic/variant_F_fft_focus/psf_ms_by_reduce_method'),
        'c5F_pipelined_ms': get('polychromatic/variant_F_fft_focus/pipelined_ms_per_psf'), 'c5M_psf_ms': get('polychromatic/variant_M_mdft_512/psf_ms'), 'c5M_czt_psf_ms': get('polychromatic/variant_M_czt_512/psf_ms'),
        'reduce_method': get('polychromatic/reduce_method'), 'reduce_alone_ms': get('polychromatic/reduce_alone_ms'),
        'model_eff8_a2a': get('polychromatic/scaling_model/per_N/8/efficiency_single_shot/a2a'),
        'model_eff8_reduce': get('polychromatic/scaling_model/per_N/8/efficiency_single_shot/reduce'),
        'c5_2048_us_per_wvl': get('polychromatic_2048/spectral_groups/per_wavelength_us_per_gpu'),
        'cpu_1core_per_s': get('cpu_baseline/value'), 'cpu_best_workers_per_s': get('cpu_baseline/tuned/value'), 'cpu_best_workers': get('cpu_baseline/tuned/cores'), 'cpu': get('cpu_baseline/host/cpu'),
    }
    return {k: v for k, v in pairs.items() if v is not None}


def emit(line):
    out = _slim(line)
    out['summary'] = _summary(out)
    print(json.dumps(out), flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--edge', dest='n', type=int, default=N_EDGE, help='transform edge (default 4096, the headline); not --n: torch.distributed.run would read that as an abbreviation of its own options')
    ap.add_argument('--dtype', default='c64', choices=['c64', 'c128'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-poly', action='store_true', help='only the headline loop (profiling runs)')
    ap.add_argument('--only', default='', help='profiling runs: time only this other_configs entry (model7|config2|config3|config4|c128|n8192|padded|composite|mtf|conv|adjoint|poly2048)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget for the CPU baseline sample')
    ap.add_argument('--reduce-method', default='auto', choices=['auto', 'reduce', 'a2a', 'rs'],
                    help='how the polychromatic image reaches rank 0: one torch.distributed.reduce, or all-to-all of slices + ordered local '
                         'sum + gather (one message per xGMI link); auto = whichever reduces the 67 MB image faster in this run')
    ap.add_argument('--extras-budget', type=float, default=900.0,
                    help='seconds the side measurements after the timed region may take before the headline line is printed without them')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='torch.distributed backend (nccl = RCCL; gloo only to exercise the N > 1 path when the ranks share a GPU)')
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    have = torch.cuda.device_count()
    if args.backend == 'nccl' and have < args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus}: only {have} GPU(s) visible; RCCL needs one device per rank '
                         '(--backend gloo lets ranks share a device, for testing the N > 1 path only)')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def make_field(n, cdtype, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))).astype(cdtype)
    return x


def plain_copy_pair_ms(x, reps=50):
    """Two dependent device-to-device copies of the field, in -> workspace -> out (torch's copy kernel): what the data movement of a
    two-pass transform costs on this box with no arithmetic at all -- the practical ceiling the step is compared with (round 4)."""
    y, z = torch.empty_like(x), torch.empty_like(x)
    for _ in range(5):
        y.copy_(x)
        z.copy_(y)
    best = None
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            y.copy_(x)
            z.copy_(y)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps
        best = t if best is None or t < best else best
    return best


def kernel_pass_times(x, n, reps=100):
    """Average duration (ms) of the row pass and of the column pass: each pass as its own back-to-back loop of `reps` launches between
    one pair of HIP events on the launch stream (pm_fft2_time_passes)."""
    from prysm_amd import _lib as L
    from prysm_amd import _ops
    lib = L.load()
    d = L.pm_fft2_desc()
    d.dtype = L.code(x)
    d.direction = -1
    d.scale = 1.0 / n
    d.weight = 1.0
    ax = _ops._axis(n, n, 0, n // 2)
    d.in_y = d.in_x = d.out_y = d.out_x = ax
    d.in_ld = d.out_ld = n
    out = torch.empty_like(x)
    nbytes = lib.pm_fft2_workspace(ctypes.byref(d))
    ws = L.workspace(nbytes)
    ms = (ctypes.c_double * 2)()
    L.check(lib.pm_fft2_time_passes(ctypes.byref(d), L.ptr(x), L.ptr(out), L.ptr(ws), ws.numel(), reps, ms,
                                    L.stream_ptr()))
    torch.cuda.synchronize()
    return float(ms[0]), float(ms[1])


def source_fingerprint():
    """sha256 over the kernel sources and the public header: what a PMC summary is valid for."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'prysm_amd', 'csrc', '*.h')) + glob.glob(os.path.join(ROOT, 'prysm_amd', 'csrc', '*.hip')) +
                   [os.path.join(ROOT, 'prysm_amd', 'csrc', 'Makefile'), os.path.join(ROOT, 'include', 'prysm_amd.h')])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, n, dtype_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary of THIS command
    (profiles/pmc_bench_summary.json: separate --pmc FETCH_SIZE and WRITE_SIZE passes, FETCH_SIZE doubled per
    the gfx950 correction in MI355X_MICROARCH.md).  Counters cannot be read from inside the process, so the summary
    carries the fingerprint of the kernel sources it was collected at; when the sources have changed since, the number
    no longer describes this build and traffic is null."""
    path = os.path.join(ROOT, 'profiles', 'pmc_bench_summary.json')
    if not os.path.exists(path):
        return None, 'no PMC summary committed'
    try:
        tab = json.load(open(path))
    except Exception:
        return None, 'PMC summary unreadable'
    stamp = tab.get('_meta', {}).get('source_fingerprint')
    now = source_fingerprint()
    if stamp != now:
        return None, f'stale: PMC summary collected at source fingerprint {stamp}, this build is {now}'
    want = f"fft_{kernel}"
    real = 'float' if dtype_name == 'c64' else 'double'
    # the folded column pass runs kernels of n/2 (n/4) points, several planes per launch; the row pass kernels of n points
    for nn in (n, n // 2, n // 4):
        for k, v in tab.items():
            if k.startswith(want) and k.endswith(f'_{real}_N{nn}'):
                return v['hbm_traffic_bytes'], f'profiles/pmc_bench_summary.json:{k} (source fingerprint {stamp})'
    return None, 'kernel not in the PMC summary'


SIDE_PREWARM_MS = 100.0    # untimed run-in of every side measurement (the headline loop has its own, PREWARM_MIN_MS): they follow each other with
                           # host-side set-up in between, during which the device starts to drop the clock it had reached


def _event_ms(fn, reps, warm=3, batches=3, prewarm_ms=SIDE_PREWARM_MS):
    """HIP-event time per call: the median of `batches` back-to-back batches of `reps` calls (side measurements only; the
    headline loop is timed once, as the contract says).  Before the batches the call runs untimed for `prewarm_ms` of wall time:
    between the sections of this script the device idles through host-side set-up and drops its clocks, and a 3-call warm-up left
    the MFMA products of config 4 at 152 - 156 us where the same call in a steady loop takes 140.6 (profiles/r05/exp_knob_gemm.log,
    tools/exp_knob_ab.py) -- the figure a model that calls it repeatedly sees."""
    for _ in range(warm):
        fn()
    if prewarm_ms > 0:
        torch.cuda.synchronize()
        t_end = time.perf_counter() + prewarm_ms * 1e-3
        while time.perf_counter() < t_end:
            fn()
        torch.cuda.synchronize()
    out = []
    for _ in range(batches):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return sorted(out)[len(out) // 2]


def _hbm_entry(ms, nbytes, note=None):
    gbs = nbytes / (ms * 1e-3) / 1e9
    e = {'ms': ms, 'algorithmic_GBps': gbs, 'frac_of_hbm_peak': gbs / HBM_PEAK_GBS,
         'frac_of_measured_copy_ceiling': gbs / HBM_COPY_CEILING_GBS}
    if note:
        e['note'] = note
    return e


# ---- the workload the reference publishes numbers for: a seven-plane Lyot-coronagraph model at 1024^2 per wavelength
MODEL7 = {'n': 1024, 'fpm_samples': 256, 'det_samples': 256, 'efl': 1000.0, 'pupil_diameter': 10.0, 'fpm_radius_lod': 3.0, 'lyot_fraction': 0.9}
MODEL7_PUBLISHED = {'titan_xp_cupy_ms_per_wavelength': 2.0, 'titan_xp_9_wavelengths_ms': 60.0, 'xeon_6248R_x2_ms_per_plane': 50.0,
                    'lowfs_model_ms_per_wavelength': 0.43,
                    'source': 'docs/source/how-tos/GPU and Exascale Computing.ipynb (file line 74); docs/source/releases/v0.20.rst:31-37'}


def model7_inputs(n=MODEL7['n'], seed=7):
    """Static data of the seven-plane model as float64 numpy arrays (cast by the caller): entrance-pupil amplitude (circle) and OPD
    (nm, smooth random), a deformable-mirror surface (nm), a Lyot stop, and -- per focal grid -- the occulter: a hard-edged spot of
    fpm_radius_lod lambda/D (real-valued, 0 inside), on the FPM grid of to_fpm_and_back."""
    rng = np.random.default_rng(seed)
    D = MODEL7['pupil_diameter']
    dx = D / n
    ax = (np.arange(n) - n // 2) * dx
    r = np.hypot(ax[None, :], ax[:, None])
    amp = (r <= D / 2).astype(np.float64)
    yy, xx = ax[:, None] / (D / 2), ax[None, :] / (D / 2)
    c = rng.standard_normal(6)
    opd = 20.0 * (c[0] * xx + c[1] * yy + c[2] * (2 * (xx * xx + yy * yy) - 1) + c[3] * (xx * xx - yy * yy) + c[4] * 2 * xx * yy +
                  c[5] * (3 * (xx * xx + yy * yy) - 2) * xx)
    dm = 5.0 * np.cos(2 * np.pi * 6 * xx) * np.cos(2 * np.pi * 4 * yy)
    lyot = (r <= MODEL7['lyot_fraction'] * D / 2).astype(np.float64)
    return {'amp': amp, 'opd': opd, 'dm': dm, 'lyot': lyot, 'dx': dx}


def model7_grids(wvl):
    """(fpm_dx, det_dx) in um: the FPM plane sampled at 8 samples per lambda/D over +-16 lambda/D, the detector at 4 per lambda/D"""
    lod = wvl * MODEL7['efl'] / MODEL7['pupil_diameter']        # um
    return lod / 8.0, lod / 4.0


def model7_fpm(wvl):
    fpm_dx, _ = model7_grids(wvl)
    m = MODEL7['fpm_samples']
    ax = (np.arange(m) - m // 2) * fpm_dx
    rr = np.hypot(ax[None, :], ax[:, None])
    lod = wvl * MODEL7['efl'] / MODEL7['pupil_diameter']
    return (rr > MODEL7['fpm_radius_lod'] * lod).astype(np.float64)


def model7(P, amp, opd, dm, fpm, lyot, wvl, dx, ex_fpm, ex_det):
    """The seven planes, written against the reference's Wavefront API (prysm/propagation/wavefront.py, coronagraph.py:12-43):
    1 entrance pupil (from_amp_and_phase) -> 2 deformable mirror (phase_screen, multiply) -> 3 focal-plane-mask plane (focus_dft)
    -> 4 after the mask (multiply) -> 5 Lyot plane (unfocus_dft) -> 6 after the Lyot stop (multiply) -> 7 detector (focus_dft,
    intensity).  `P` is prysm_amd.propagation."""
    wf = P.Wavefront.from_amp_and_phase(amp, opd, wvl, dx)
    wf = wf * P.Wavefront.phase_screen(dm, wvl, dx)
    at_lyot = wf.to_fpm_and_back(fpm, ex_fpm)
    after_lyot = at_lyot * P.Wavefront(lyot, wvl, dx)
    return after_lyot.focus_dft(ex_det).intensity.data


def model7_flops():
    n, a, b = MODEL7['n'], MODEL7['fpm_samples'], MODEL7['det_samples']
    one = lambda m: 8.0 * m * n * (n + m)      # noqa: E731   two complex GEMMs of a (n x n) <-> (m x m) matrix DFT
    return {'focus_to_fpm': one(a), 'unfocus_to_lyot': one(a), 'focus_to_detector': one(b), 'total': 2 * one(a) + one(b)}


def model7_bytes(es):
    """algorithmic HBM bytes of the pointwise planes (complex element size es): synthesis (2 maps in, field out), DM screen folded
    into a multiply (map in, field in / out), mask multiply on the FPM grid, Lyot multiply, |.|^2"""
    n, a, b = MODEL7['n'], MODEL7['fpm_samples'], MODEL7['det_samples']
    r = es // 2
    return {'pupil_synthesis': n * n * (2 * r + es), 'dm_screen_and_multiply': n * n * (r + es) + n * n * 3 * es, 'fpm_multiply': a * a * (2 * es + r),
            'lyot_multiply': n * n * (2 * es + r), 'intensity': b * b * (es + r)}


def sec_model7(out, wavelengths=9):
    """model_7plane_1024: per wavelength -- eager, inside graph.sequence(), as one hipGraph replay -- at complex64
    and complex128, plus the 9-wavelength aggregate the reference quotes; `published` = BASELINE.md's figures (other hardware)."""
    from prysm_amd import propagation as P
    from prysm_amd import graph as G
    from prysm_amd.conf import config
    res = {'planes': 7, 'pupil': f"{MODEL7['n']}^2", 'fpm_grid': f"{MODEL7['fpm_samples']}^2", 'detector': f"{MODEL7['det_samples']}^2",
           'flops_per_wavelength': model7_flops(), 'published': MODEL7_PUBLISHED}
    wvls = list(np.linspace(0.55, 0.65, wavelengths))
    prec0 = config.precision
    try:
        for prec, tag in ((32, 'c64'), (64, 'c128')):
            config.precision = prec
            rdt = torch.float32 if prec == 32 else torch.float64
            inp = model7_inputs()
            dev = {k: torch.from_numpy(v).to(rdt).cuda() for k, v in inp.items() if k != 'dx'}
            dx = inp['dx']
            def build(kind):
                per = []
                for w in wvls:
                    fdx, ddx = model7_grids(w)
                    per.append((w, torch.from_numpy(model7_fpm(w)).to(rdt).cuda(),
                                P.prepare_executor(dx, MODEL7['n'], fdx, MODEL7['fpm_samples'], w, MODEL7['efl'], kind=kind),
                                P.prepare_executor(dx, MODEL7['n'], ddx, MODEL7['det_samples'], w, MODEL7['efl'], kind=kind)))
                return per

            def measure(per, full):
                def one(k=0):
                    w, fpm, exa, exb = per[k]
                    return model7(P, dev['amp'], dev['opd'], dev['dm'], fpm, dev['lyot'], w, dx, exa, exb)

                def all_wvls():
                    acc = None
                    for k in range(len(per)):
                        i = one(k)
                        acc = i if acc is None else acc + i
                    return acc

                def all_wvls_block():
                    with G.sequence(streams=3):
                        imgs = [one(k) for k in range(len(per))]
                    acc = imgs[0]
                    for i in imgs[1:]:
                        acc = acc + i
                    return acc

                e = {}
                e['eager_ms_per_wavelength'] = _event_ms(one, 50)
                g1 = G.capture(lambda a, o: model7(P, a, o, dev['dm'], per[0][1], dev['lyot'], per[0][0], dx, per[0][2], per[0][3]), dev['amp'], dev['opd'])
                e['graph_ms_per_wavelength'] = _event_ms(g1.graph.replay, 100)
                e['graph_identical_to_eager'] = bool(torch.equal(g1(dev['amp'], dev['opd']), one()))
                e['eager_9wvl_ms'] = _event_ms(all_wvls, 10)
                if full:
                    e['sequence_9wvl_ms'] = _event_ms(all_wvls_block, 10)
                g9 = G.capture(lambda a, o: all_wvls(), dev['amp'], dev['opd'])
                e['graph_9wvl_ms'] = _event_ms(g9.graph.replay, 20)
                # the nine independent chains captured INSIDE a sequence block: the hipGraph gets one branch per ring stream, the chains
                # overlap on the device (small GEMMs that fill a quarter of the CUs each) and no host dispatch is left to pay for it
                g9s = G.capture(lambda a, o: all_wvls_block(), dev['amp'], dev['opd'])
                e['graph_branches_9wvl_ms'] = _event_ms(g9s.graph.replay, 20)
                e['graph_branches_identical'] = bool(torch.equal(g9s(dev['amp'], dev['opd']), g9(dev['amp'], dev['opd'])))
                e['graph_branches_ms_per_wavelength'] = e['graph_branches_9wvl_ms'] / len(per)
                del g1, g9, g9s
                return e

            per = build('mdft')
            e = measure(per, True)
            e['bytes_pointwise_planes'] = model7_bytes(8 if prec == 32 else 16)
            fl = model7_flops()['total']
            peak = F32_MFMA_PEAK_TF if prec == 32 else F64_MFMA_PEAK_TF
            e['graph_frac_of_mfma_peak'] = fl / (e['graph_ms_per_wavelength'] * 1e-3) / 1e12 / peak
            e['graph_branches_frac_of_mfma_peak'] = fl / (e['graph_branches_ms_per_wavelength'] * 1e-3) / 1e12 / peak
            del per
            # the same planes through the chirp-Z executors the reference offers beside the matrix DFT (prepare_executor(kind='czt'),
            # prysm/fttools.py:235-389): one fused convolution kernel per axis instead of two GEMMs
            per = build('czt')
            e['czt'] = measure(per, False)
            res[tag] = e
            del per, dev
            torch.cuda.empty_cache()
    finally:
        config.precision = prec0
    out['model_7plane_1024'] = res


def other_configs(only=''):
    """BASELINE configs 2 - 4 and the two honest-HBM focus cases (nothing fits the 256 MiB Infinity Cache) measured in
    the same run: algorithmic bytes / flops per SURVEY 8(d) over HIP-event time on the launch stream.  Parity-tested
    cases reported for the roofline the north star asks for; not the headline value."""
    from prysm_amd import propagation as P
    from prysm_amd.conf import config
    out = {}

    def want(key):
        return not only or only == key

    # every section is its own try: a failure in one side measurement costs that entry, not the others
    def sec_config2():   # config 2: 2048^2 complex64 focus, 4 N^2 s bytes
        x2 = torch.from_numpy(make_field(2048, np.complex64, 2048)).cuda()
        out['config2_focus_2048_c64'] = _hbm_entry(_event_ms(lambda: P.focus(x2, 1), 100), 4 * 2048 ** 2 * 8)
        # the same propagation as a SEQUENCE of independent fields alternating between two HIP streams (prysm_amd.graph.StreamRing):
        # throughput of a wavelength / field-point loop at this size, where one launch pair alone is latency-bound
        from prysm_amd.graph import StreamRing
        ring, x2b, keep = StreamRing(2), x2.clone(), [None, None]

        def sequence(k=100):       # 2 k propagations, free-running on the two streams, forked / joined once
            ring.fork()
            for _ in range(k):
                keep[0] = None
                keep[0] = ring.run(P.focus, x2, 1)
                keep[1] = None
                keep[1] = ring.run(P.focus, x2b, 1)
            ring.join()
        e2 = _hbm_entry(_event_ms(sequence, 3, warm=1) / 200, 4 * 2048 ** 2 * 8)
        out['config2_focus_2048_c64']['two_streams'] = dict(e2, note='a sequence of 200 independent 2048^2 propagations alternating between two HIP '
                                                                     'streams (StreamRing), joined once at the end; ms per propagation')
        # ... and the same loop written as plain calls inside a prysm_amd.graph.sequence() block (round 5): the block picks the streams.
        # `loop_ms` is the SAME loop (two inputs alternating, two results alive) on one stream.
        from prysm_amd import graph as G

        def loop(xa, xb, k):
            for _ in range(k):
                keep[0] = None
                keep[0] = P.focus(xa, 1)
                keep[1] = None
                keep[1] = P.focus(xb, 1)

        def block(xa, xb, k):
            with G.sequence():
                loop(xa, xb, k)

        def pair(xa, xb, k, nbytes):
            plain = _event_ms(lambda: loop(xa, xb, k), 3, warm=1) / (2 * k)
            return dict(_hbm_entry(_event_ms(lambda: block(xa, xb, k), 3, warm=1) / (2 * k), nbytes), loop_ms=plain)
        out['config2_focus_2048_c64']['sequence_block'] = pair(x2, x2b, 100, 4 * 2048 ** 2 * 8)
        del x2, x2b
        # complex128 at the same size: two such propagations do not share the Infinity Cache, the block keeps them on one stream
        xc = torch.from_numpy(make_field(2048, np.complex128, 2048)).cuda()
        xd = xc.clone()
        out['focus_2048_c128_sequence_block'] = pair(xc, xd, 50, 4 * 2048 ** 2 * 16)
        del xc, xd
        xa = torch.from_numpy(make_field(1000, np.complex64, 1000)).cuda()
        xb = xa.clone()
        out['focus_1000_c64_sequence_block'] = pair(xa, xb, 100, 4 * 1000 ** 2 * 8)
        del xa, xb, keep
    def sec_config3():   # config 3: 4096^2 complex128 angular-spectrum step (fused 3 passes), graded on 8 N^2 s bytes
        x3 = torch.from_numpy(make_field(4096, np.complex128, 4096)).cuda()
        out['config3_angular_spectrum_4096_c128'] = _hbm_entry(
            _event_ms(lambda: P.angular_spectrum(x3, 0.6328, 0.01, 10.0, Q=1), 30), 8 * 4096 ** 2 * 16,
            'graded on 8 N^2 s (two 2-D transforms); the fused chain moves 6 N^2 s: moved_* is the fraction on those')
        e3 = out['config3_angular_spectrum_4096_c128']
        e3['moved_GBps'] = 0.75 * e3['algorithmic_GBps']
        e3['moved_frac_of_hbm_peak'] = 0.75 * e3['frac_of_hbm_peak']
        del x3
    def sec_c128():      # nothing of this one fits the Infinity Cache: 256 MiB in, 256 MiB intermediate, 256 MiB out
        x5 = torch.from_numpy(make_field(4096, np.complex128, 4096)).cuda()
        out['focus_4096_c128'] = _hbm_entry(_event_ms(lambda: P.focus(x5, 1), 30), 4 * 4096 ** 2 * 16)
        del x5
    def sec_n8192():
        x6 = torch.from_numpy(make_field(8192, np.complex64, 8192)).cuda()
        out['focus_8192_c64'] = _hbm_entry(_event_ms(lambda: P.focus(x6, 1), 20), 4 * 8192 ** 2 * 8)
        del x6
    def sec_composite():
        # lengths scipy factors natively (prysm/propagation/fft.py:24): the composite register engine (csrc/fft_ce.h, round 5) where the
        # length has a compile-time plan, else the general mixed-radix kernel (csrc/fft_mixed.h); 4 N^2 s bytes
        from prysm_amd import _lib as L_
        for n, cdt, key in ((3000, np.complex64, 'focus_3000_c64_mixed_radix'), (1000, np.complex64, 'focus_1000_c64_mixed_radix'),
                            (3000, np.complex128, 'focus_3000_c128_mixed_radix')):
            xc = torch.from_numpy(make_field(n, cdt, n)).cuda()
            out[key] = _hbm_entry(_event_ms(lambda: P.focus(xc, 1), 50), 4 * n ** 2 * xc.element_size(),
                                  'composite length on its own factors, register-resident with a compile-time plan (round 5; until round 4 '
                                  'LDS-resident with a run-time plan: general_kernel_ms)')
            with L_.tuning_local(mix_engine=0):
                out[key]['general_kernel_ms'] = _event_ms(lambda: P.focus(xc, 1), 50)
            del xc
        # round 5: the Q = 1.5 pad of a 1024^2 pupil (prysm/propagation/fft.py:7-25 with Q = 1.5 -> 1536^2, plan 24 x 8 x 8) and a 2000^2 grid
        for n, cdt, key in ((1536, np.complex64, 'focus_1536_c64_composite'), (2000, np.complex128, 'focus_2000_c128_composite')):
            xc = torch.from_numpy(make_field(n, cdt, n)).cuda()
            out[key] = _hbm_entry(_event_ms(lambda: P.focus(xc, 1), 50), 4 * n ** 2 * xc.element_size())
            with L_.tuning_local(mix_engine=0):
                out[key]['general_kernel_ms'] = _event_ms(lambda: P.focus(xc, 1), 50)
            del xc
        # round 5: (B, m, n) stacks of small composite fields (the reference's multi-field batches, prysm/x/polarization.py:478-553) run as ONE
        # launch pair on the register engine (grid.y = fields); `loop_ms_per_field`: the same fields as B calls
        for n, B, key in ((500, 16, 'focus_stack_16x500_c64_composite'), (1000, 8, 'focus_stack_8x1000_c64_composite')):
            xs = torch.from_numpy(np.stack([make_field(n, np.complex64, n + b) for b in range(B)])).cuda()
            t = _event_ms(lambda: P.focus(xs, 1), 50)
            out[key] = {'ms': t, 'ms_per_field': t / B, 'fields': B, 'algorithmic_GBps': 4 * B * n ** 2 * 8 / t / 1e6,
                        'loop_ms_per_field': _event_ms(lambda: [P.focus(xs[b], 1) for b in range(B)], 20) / B}
            del xs
        # 6006 = 6 x 7 x 11 x 13 (the mixed-radix kernel as it is) and 10000 = 2 x 5000 (round 4: one radix-2 step around
        # mixed-radix sub-transforms; rounds 1 - 3 convolved it at 32768 points per axis)
        for n, key in ((6006, 'focus_6006_c64_mixed_radix'), (10000, 'focus_10000_c64_radix2_x_mixed_radix')):
            xc = torch.from_numpy(make_field(n, np.complex64, n)).cuda()
            out[key] = _hbm_entry(_event_ms(lambda: P.focus(xc, 1), 10, warm=2), 4 * n ** 2 * xc.element_size())
            del xc
        # the free-space step on a composite grid (prysm/propagation/angular_spectrum.py:9-42 takes any size): three passes with the
        # middle pass resident on chip (round 4: in LDS; round 5: in registers), graded like config 3 on 8 N^2 s; `composed_ms`: two pm_fft2
        # calls (rounds 1 - 3)
        for n, cdt, prec_, key in ((3000, np.complex128, 64, 'angular_spectrum_3000_c128_composite'), (3000, np.complex64, 32, 'angular_spectrum_3000_c64_composite')):
            xa = torch.from_numpy(make_field(n, cdt, n + 1)).cuda()
            prec0 = config.precision
            config.precision = prec_
            try:
                e = _hbm_entry(_event_ms(lambda: P.angular_spectrum(xa, 0.6328, 0.01, 10.0, Q=1), 30), 8 * n ** 2 * xa.element_size(),
                               'fft2 x H ifft2 on a 3000^2 grid: row pass, middle pass with the column tile resident in registers (forward transform, '
                               'x H, inverse transform), inverse row pass; graded on 8 N^2 s, moves 6 N^2 s')
                with L_.tuning_local(mix_engine=0):
                    e['general_kernel_ms'] = _event_ms(lambda: P.angular_spectrum(xa, 0.6328, 0.01, 10.0, Q=1), 30)
                with L_.tuning_local(mix_fused=0):
                    e['composed_ms'] = _event_ms(lambda: P.angular_spectrum(xa, 0.6328, 0.01, 10.0, Q=1), 30)
            finally:
                config.precision = prec0
            e['moved_frac_of_hbm_peak'] = 0.75 * e['frac_of_hbm_peak']
            out[key] = e
            del xa
    def sec_padded():    # the padded (Q = 2) cases reported separately, graded on 4 N^2 s of the TRANSFORM size although
        for npup, key in ((2048, 'focus_Q2_2048_to_4096_c64'), (1024, 'focus_Q2_1024_to_2048_c64')):     # the row pass skips the zero rows
            xp = torch.from_numpy(make_field(npup, np.complex64, npup + 7)).cuda()
            out[key] = _hbm_entry(_event_ms(lambda: P.focus(xp, 2), 50), 4 * (2 * npup) ** 2 * 8,
                                  'pad2d fused into the load window: the row pass transforms only the stored rows, so fewer bytes move than the '
                                  '4 N^2 s (N = output edge) the fraction is graded on')
            moved = 2.25 * (2 * npup) ** 2 * 8      # read N^2 s / 4, write + read the N/2 stored rows of the intermediate, write N^2 s
            out[key]['moved_GBps'] = moved / (out[key]['ms'] * 1e-3) / 1e9
            out[key]['moved_frac_of_hbm_peak'] = out[key]['moved_GBps'] / HBM_PEAK_GBS
            del xp
    def sec_mtf():       # MTF of a real 4096^2 fp32 PSF -- Hermitian transform with the centre normalisation and |.| in
        from prysm_amd import otf     # the column pass's epilogue (one launch pair) against transform + elementwise sweeps
        psf = torch.rand(4096, 4096, dtype=torch.float32, device='cuda') + 0.01
        ms = _event_ms(lambda: otf.mtf_from_psf(psf, 1.0), 30)
        msc = _event_ms(lambda: otf.mtf_from_psf(psf, 1.0, return_more=True), 10)
        out['mtf_from_psf_4096_f32'] = {'ms': ms, 'composed_ms': msc,
                                        'note': 'fused: real-input (Hermitian) transform, N/2 columns, DC normalisation + abs in the store; composed '
                                                '(return_more=True): complex spectrum + division + abs as separate device sweeps'}
        del psf
        # the same on a composite grid (round 5): half-size transform on the mixed-radix kernels + one untangling sweep with the
        # normalisation and |.| fused (_ops.fft2_real); graded on 12 B per sample (4 read, 8 intermediate ... the transform moves 16 + 8)
        psf3 = torch.rand(3000, 3000, dtype=torch.float32, device='cuda') + 0.01
        ms3 = _event_ms(lambda: otf.mtf_from_psf(psf3, 1.0), 30)
        ms3c = _event_ms(lambda: otf.mtf_from_psf(psf3, 1.0, return_more=True), 10)
        out['mtf_from_psf_3000_f32'] = {'ms': ms3, 'composed_ms': ms3c,
                                        'note': 'real 3000^2 PSF: (3000 x 1500) complex transform + pm_r2c_untangle; composed = return_more=True'}
        del psf3
    def sec_conv():      # a real 4096^2 fp32 object through a transfer function (apply_transfer_functions: the image-chain
        from prysm_amd import _ops     # step after the PSF) -- half spectra end to end against the complex chain on the same arrays
        obj = torch.rand(4096, 4096, dtype=torch.float32, device='cuda')
        Hc = torch.randn(4096, 4096, dtype=torch.complex64, device='cuda')
        kw = dict(scale=1.0 / 4096 ** 2, mul=Hc, in_shift=(2048, 2048), out_shift=(2048, 2048))
        ms = _event_ms(lambda: _ops.fft2_mul_ifft2(obj, real_out=True, **kw), 30)
        msc = _event_ms(lambda: _ops.fft2_mul_ifft2(obj, **kw), 30)
        out['conv_real_4096_f32'] = {'ms': ms, 'complex_chain_ms': msc, 'algorithmic_GBps': 32 * 4096 ** 2 / (ms * 1e-3) / 1e9,
                                     'frac_of_hbm_peak': 32 * 4096 ** 2 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     'note': 'real(ifft2(fft2(obj) H)) of a real object: R2C rows, Hermitian part of H between the column transforms, '
                                             'C2R rows = 32 B per sample (4 + 4, 8 + 8 of H, 4 + 4); the complex chain moves 56'}
        del obj, Hc
    def sec_adjoint():   # the gradient path (what optimisers run), same grading as the forward operators
        adj = {}
        g = torch.from_numpy(make_field(4096, np.complex64, 77)).cuda()
        adj['focus_adjoint_4096_to_2048_c64_Q2'] = _hbm_entry(
            _event_ms(lambda: P.focus_adjoint(g, 2), 50), 4 * 4096 ** 2 * 8,
            'ifft2 (ortho) of a 4096^2 focal-plane gradient + crop to the 2048^2 pupil in the store window; graded on 4 N^2 s of the '
            'transform size, moves 2.75 N^2 s (the last pass only transforms and stores the kept rows)')
        e = adj['focus_adjoint_4096_to_2048_c64_Q2']
        e['moved_frac_of_hbm_peak'] = 2.5 / 4 * e['frac_of_hbm_peak']
        adj['focus_adjoint_4096_c64_Q1'] = _hbm_entry(_event_ms(lambda: P.focus_adjoint(g, 1), 50), 4 * 4096 ** 2 * 8)
        W = P.Wavefront(g, 0.6328, 1.0, space='psf')
        ib = torch.rand(4096, 4096, dtype=torch.float32, device='cuda')
        adj['intensity_adjoint_4096_c64'] = _hbm_entry(_event_ms(lambda: W.intensity_adjoint(ib), 50), (4 + 8 + 8) * 4096 ** 2,
                                                      '2 Ibar E as one sweep (pm_rmul): 4 + 8 B read, 8 B written per sample')
        del g, W, ib
        g3 = torch.from_numpy(make_field(4096, np.complex128, 78)).cuda()
        adj['angular_spectrum_adjoint_4096_c128'] = _hbm_entry(
            _event_ms(lambda: P.angular_spectrum_adjoint(g3, 0.6328, 0.01, 10.0, Q=1), 30), 8 * 4096 ** 2 * 16,
            'the same fused 3-pass chain with conj(H); graded on 8 N^2 s, moves 6 N^2 s')
        adj['angular_spectrum_adjoint_4096_c128']['moved_frac_of_hbm_peak'] = 0.75 * adj['angular_spectrum_adjoint_4096_c128']['frac_of_hbm_peak']
        del g3
        prec = config.precision
        try:
            config.precision = 32
            ex = P.prepare_executor(10 / 2048, (2048, 2048), 0.6328 * 10 / 8, (512, 512), 0.6328, 100.0)
            gm = torch.from_numpy(make_field(512, np.complex64, 79)).cuda()
            ms = _event_ms(lambda: P.focus_dft_adjoint(gm, ex), 50)
            fl = 8 * 2048 * 512 * (512 + 2048)     # Ey^H (2048 x 512) @ g (512 x 512) @ conj(Ex) (512 x 2048), cheaper product first
            adj['mdft_adjoint_512_to_2048_c64'] = {
                'ms': ms, 'algorithmic_TFLOPs': fl / (ms * 1e-3) / 1e12, 'frac_of_f32_mfma_peak': fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TF,
                'bound': 'mfma', 'note': 'MDFT.adjoint of config 4: (Ey^H @ g @ conj(Ex)) norm, 512^2 -> 2048^2, two complex GEMMs'}
            del gm, ex
        finally:
            config.precision = prec
        out['adjoints'] = adj
    def sec_config4():   # config 4: matrix-DFT focus 2048^2 -> 512^2 complex64 on MFMA, 8 My Nx (Ny + Mx) real flops
        prec = config.precision
        try:
            config.precision = 32
            x4 = torch.from_numpy(make_field(2048, np.complex64, 2048)).cuda()
            ex = P.prepare_executor(10 / 2048, (2048, 2048), 0.6328 * 10 / 8, (512, 512), 0.6328, 100.0)
            ms = _event_ms(lambda: P.focus_dft(x4, ex), 50)
            # the executor BUILD (two basis matrices on the device, the reference's one-off cost per grid and
            # wavelength) timed apart from its application
            build_ms = _event_ms(lambda: P.prepare_executor(10 / 2048, (2048, 2048), 0.6328 * 10 / 8, (512, 512), 0.6328, 100.0), 20)
            fl = 8 * 512 * 2048 * (2048 + 512)
            out['config4_mdft_2048_to_512_c64'] = {
                'ms': ms, 'prepare_executor_ms': build_ms, 'algorithmic_TFLOPs': fl / (ms * 1e-3) / 1e12,
                'frac_of_f32_mfma_peak': fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TF,
                'bound': 'mfma', 'note': 'two complex GEMMs on v_mfma_f32_32x32x2_f32; peak 157.3 TFLOP/s (MI355X_MICROARCH.md)'}
            # the graded flops are the four-product form's; the kernels run the 3M form (three real products per complex one), so they
            # EXECUTE 0.75 of them: `frac_of_f32_mfma_peak` is an accounting fraction, `executed_frac_of_f32_mfma_peak` and the counters'
            # `mfma_busy` (SQ_VALU_MFMA_BUSY_CYCLES over SIMD-cycles, from the committed PMC pass of this command while the kernel
            # sources are unchanged) are utilisation
            e4 = out['config4_mdft_2048_to_512_c64']
            e4['executed_TFLOPs'] = 0.75 * e4['algorithmic_TFLOPs']
            e4['executed_frac_of_f32_mfma_peak'] = 0.75 * e4['frac_of_f32_mfma_peak']
            try:
                tab = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_config4_mfma_busy.json')))
                if tab.get('_meta', {}).get('source_fingerprint') == source_fingerprint():
                    e4['mfma_busy'] = tab.get('mfma_busy_time_weighted_over_the_two_products')
                else:
                    e4['mfma_busy_note'] = 'stale: counters collected at other kernel sources'
            except (OSError, ValueError):
                pass
            if not only:
                # the same focal grid by the chirp-Z executor (prysm/fttools.py:235-389), for comparison
                exz = P.prepare_executor(10 / 2048, (2048, 2048), 0.6328 * 10 / 8, (512, 512), 0.6328, 100.0, kind='czt')
                msz = _event_ms(lambda: P.focus_dft(x4, exz), 20)
                out['czt_2048_to_512_c64'] = {'ms': msz, 'note': 'kind="czt" executor on the config-4 grid (one fused convolution kernel per axis)'}
                # ... and by the FFT-accelerated DFT (prysm/fttools.py:392-535) on the same shapes with K = 8192 per axis: one
                # pm_fft1_ramp kernel per axis.  complex128 on binary-spaced grids (dx = 1/256, dfx = 1/32): the reference's spacing test
                # (32 eps) rejects prepare_executor's decimal grids at this size, at either precision
                try:
                    from prysm_amd import fttools
                    config.precision = 64
                    rr = lambda n_: (torch.arange(n_, dtype=torch.float64) - n_ // 2).numpy()     # noqa: E731
                    xs, fs = rr(2048) / 256.0, rr(512) / 32.0
                    exf = fttools.FFTDFT(xs, xs, fs, fs, norm=1.0 / 8192)
                    exm = fttools.MDFT(xs, xs, fs, fs, norm=1.0 / 8192)
                    x4d = x4.to(torch.complex128)
                    out['fftdft_2048_to_512_K8192_c128'] = {
                        'ms': _event_ms(lambda: exf(x4d), 20), 'mdft_c128_ms': _event_ms(lambda: exm(x4d), 10),
                        'note': 'fttools.FFTDFT, complex128: ramps in the load / store of one 8192-point transform kernel per axis; '
                                'mdft_c128_ms: the matrix DFT on the same grids at the same precision'}
                    del x4d, exf, exm
                except Exception as exc:
                    out['fftdft_2048_to_512_K8192_c128'] = {'error': repr(exc)}
                config.precision = 32
            del x4, ex
        finally:
            config.precision = prec
    for key, fn in (('model7', lambda: sec_model7(out)), ('config2', sec_config2), ('config3', sec_config3), ('c128', sec_c128), ('n8192', sec_n8192), ('padded', sec_padded), ('composite', sec_composite), ('mtf', sec_mtf), ('conv', sec_conv), ('adjoint', sec_adjoint), ('config4', sec_config4)):
        if not want(key):
            continue
        try:
            fn()
        except Exception as exc:     # that entry only
            out[key + '_error'] = repr(exc)
        torch.cuda.empty_cache()
    return out


def cpu_baseline(n, cdtype, budget_s):
    """The oracle (port of prysm.propagation.focus) on host cores; scipy.fft workers=1 as prysm ships."""
    from oracle import prysm_oracle as O
    x = make_field(n, cdtype, 4096)
    O.focus(x, 1)   # warm-up
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 25):
        t0 = time.perf_counter()
        O.focus(x, 1)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    import scipy
    model = 'unknown'
    try:
        with open('/proc/cpuinfo') as fh:
            model = next((ln.split(':', 1)[1].strip() for ln in fh if ln.startswith('model name')), 'unknown')
    except OSError:
        pass
    out = {
        'value': 1.0 / med, 'unit': 'propagations/s', 'cores': 1, 'kind': 'port',
        'host': {'cpu': model, 'logical_cores': os.cpu_count(), 'numpy': np.__version__, 'scipy': scipy.__version__},
        'sample': f'{len(times)} x oracle.focus({n}x{n} {np.dtype(cdtype).name}, Q=1), median {med * 1e3:.1f} ms, '
                  f'min {min(times) * 1e3:.1f} ms; scipy.fft workers=1 (as prysm ships), host has {os.cpu_count()} cores',
    }
    # second arm, informational: the knob prysm's docs recommend (scipy.fft.set_workers), swept over worker counts up to the host's
    # logical cores -- the best one is reported (256 workers on a 64-core part was a straw man: 60 ms)
    try:
        from scipy import fft as sfft
        ncpu = os.cpu_count() or 1
        sweep = {}
        for workers in sorted({w for w in (8, 16, 32, 64, 128) if w <= ncpu} | ({ncpu} if ncpu < 8 else set())):
            with sfft.set_workers(workers):
                O.focus(x, 1)
                t2 = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    O.focus(x, 1)
                    t2.append(time.perf_counter() - t0)
            sweep[workers] = float(np.median(t2))
        best = min(sweep, key=sweep.get)
        out['tuned'] = {'value': 1.0 / sweep[best], 'cores': best, 'ms_by_workers': {str(k): v * 1e3 for k, v in sweep.items()},
                        'sample': f'5 x the same call under scipy.fft.set_workers(w), w in {sorted(sweep)}; best w = {best}, median {sweep[best] * 1e3:.1f} ms'}
    except Exception as exc:   # pragma: no cover
        out['tuned'] = {'error': repr(exc)}
    return out


class Ranks:
    """The process group of this run (or none at N = 1): barrier, MAX over ranks."""

    def __init__(self, world):
        self.world = world

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def max(self, v):
        if self.world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(self, fn):
        """Wall time of fn() bracketed by barrier + synchronize on both sides, MAX over ranks (seconds)."""
        torch.cuda.synchronize()
        self.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        self.barrier()
        torch.cuda.synchronize()
        return self.max(time.perf_counter() - t0)


PREWARM_MIN_MS = 400.0     # untimed run-in: at least this long ...
PREWARM_CAP_MS = 2000.0    # ... at most this long, ended as soon as the batch times have stopped drifting (propagation_loop)
PREWARM_TOL = 0.01


def gpu_state(dev_index=0):
    """Core / memory clock (MHz), power (W) and temperature of the device, read from sysfs (microseconds: cheap enough to take right
    before and right after the timed region) -- so that a line measured on a throttled or not-yet-ramped box says so itself
    (the driver's five headline readings spread 13 % and nothing in the lines could tell why)."""
    out = {}
    try:
        cards = sorted(c for c in glob.glob('/sys/class/drm/card[0-9]*/device') if os.path.exists(os.path.join(c, 'pp_dpm_sclk')))
        if not cards:
            return {'error': 'no /sys/class/drm/card*/device/pp_dpm_sclk'}
        # the box's sysfs shows every GPU of the host; the one this process computes on is found by its PCI address
        card = None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            bdf = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
            for c in cards:
                if os.path.basename(os.path.realpath(c)).lower() == bdf:
                    card = c
                    out['pci'] = bdf
        except Exception:
            pass
        if card is None:
            if len(cards) > 1:
                return {'error': f'{len(cards)} cards in sysfs and no PCI address to pick this device by'}
            card = cards[0]

        def current_level(name):
            try:
                for ln in open(os.path.join(card, name)):
                    if ln.rstrip().endswith('*'):
                        return float(ln.split(':')[1].lower().replace('mhz', '').replace('*', '').strip())
            except (OSError, ValueError, IndexError):
                pass
            return None

        out['sclk_mhz'] = current_level('pp_dpm_sclk')
        out['mclk_mhz'] = current_level('pp_dpm_mclk')
        out['fclk_mhz'] = current_level('pp_dpm_fclk')
        for hw in glob.glob(os.path.join(card, 'hwmon', 'hwmon*')):
            for key, fname, scale in (('power_w', 'power1_average', 1e-6), ('power_w', 'power1_input', 1e-6),
                                      ('temp_c', 'temp1_input', 1e-3), ('temp_c', 'temp2_input', 1e-3), ('sclk_hwmon_mhz', 'freq1_input', 1e-6),
                                      ('power_cap_w', 'power1_cap', 1e-6)):
                if key in out and out[key] is not None:
                    continue
                try:
                    out[key] = float(open(os.path.join(hw, fname)).read().strip()) * scale
                except (OSError, ValueError):
                    pass
        try:
            out['busy_pct'] = float(open(os.path.join(card, 'gpu_busy_percent')).read().strip())
        except (OSError, ValueError):
            pass
    except Exception as exc:     # never lose a bench line to a sysfs surprise
        out['error'] = repr(exc)
    return {k: v for k, v in out.items() if v is not None}


def run_in_is_steady(hist, at, tol=PREWARM_TOL, lookback_ms=100.0):
    """The run-in's stopping rule (pure: tests/test_host_logic.py runs it on synthetic trajectories).  `hist`: seconds per batch, `at`: ms
    since the start at which each batch ended.  Steady = the mean of the last five batches is within `tol` of the mean of the five
    batches that ended about `lookback_ms` earlier (no drift over 100 ms: a plateau of the clock ramp is shorter than that) and the
    last five agree with each other within 3 tol."""
    if len(hist) < 10 or at[-5] - lookback_ms < at[0]:
        return False
    j = max(i for i in range(len(at)) if at[i] <= at[-5] - lookback_ms)
    if j < 4:
        return False
    recent = sum(hist[-5:]) / 5
    earlier = sum(hist[j - 4:j + 1]) / 5
    return abs(recent - earlier) <= tol * earlier and max(hist[-5:]) <= min(hist[-5:]) * (1.0 + 3 * tol)


def propagation_loop(ranks, x, steps, warmup, info=None, repeats=0):
    """warmup untimed + exactly `steps` timed focus(x, 1) per rank; returns seconds (MAX over ranks).

    Before the W warm-up steps the same propagation runs untimed until the device has reached its steady state: batches of 40
    propagations (synchronised one by one) for at least PREWARM_MIN_MS, until the mean of the last five batches is within 1 % of the
    mean of the five batches about 100 ms earlier (no drift), at most PREWARM_CAP_MS.  Why: a freshly leased MI355X raises its core
    clock over several HUNDRED milliseconds of sustained load (profiles/r06/exp_warm_trajectory.log: 1990 MHz / 316 W at 100 ms,
    2145 MHz at 190 ms, 2350 MHz / 1170 W at 400 ms), and the row + fold kernel follows it up to ~2150 MHz: 97.9 us per step at 120 ms,
    94.0 from 190 ms on.  Rounds 1 - 5 ran in for a fixed 60 ms and the driver's 20-step run (2 ms of work) was timed on that ramp
    (readings 9.4 - 10.7 k/s); three equal 4 ms batches are NOT a steady state either -- the ramp has plateaux (round 6's first form
    stopped at 88 ms on one).  It is ordinary warm-up -- the same call on the same buffers, nothing cached that a timed step reuses
    beyond what step 1 leaves for step 2 -- and what the metric (propagations per second) means: a loop that runs for a second.
    `info` (a dict) receives what was spent (`prewarm_ms`, `prewarm_batches`, the trajectory of batch times), the device's clocks /
    power right before and right after the timed region (gpu_state), and -- AFTER the contract's one timed region -- `repeats` more
    batches of `steps` timed the same way (`repeat_ms`: ms per step of each), so a line shows its own spread."""
    from prysm_amd import propagation as P
    f = None
    info = {} if info is None else info

    def batch(k):
        nonlocal f
        t0 = time.perf_counter()
        for _ in range(k):
            f = None
            f = P.focus(x, 1)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    hist, at = [], []
    steady = False
    while True:
        hist.append(batch(40))
        spent = (time.perf_counter() - t_start) * 1e3
        at.append(spent)
        steady = run_in_is_steady(hist, at)
        if (steady and spent >= PREWARM_MIN_MS) or spent >= PREWARM_CAP_MS:
            break
    info['prewarm_ms'] = (time.perf_counter() - t_start) * 1e3
    info['prewarm_batches'] = len(hist)
    info['prewarm_steady'] = bool(steady)
    # ms per step of the batch that ended nearest to each mark (the first batch carries the first call's one-off set-up)
    info['prewarm_trajectory_ms_per_step'] = {f'{mark}ms': hist[min(range(len(at)), key=lambda i: abs(at[i] - mark))] / 40 * 1e3
                                              for mark in (50, 100, 200, 400, 800, 1600) if at[-1] >= mark}
    info['prewarm_last_batches_ms_per_step'] = [t / 40 * 1e3 for t in hist[-3:]]
    for _ in range(warmup):
        f = None
        f = P.focus(x, 1)

    def run():
        nonlocal f
        for _ in range(steps):
            f = None       # release the previous focal field first: the caching allocator then hands the same block
            f = P.focus(x, 1)   # back, so the steady state touches in + workspace + out (not two alternating outputs)

    dev = torch.cuda.current_device()
    info['gpu_before'] = gpu_state(dev)
    elapsed = ranks.timed(run)
    info['gpu_after'] = gpu_state(dev)
    if repeats:
        info['repeat_ms'] = [ranks.timed(run) / steps * 1e3 for _ in range(repeats)]
    return elapsed, f


def polychromatic_config5(ranks, n, reduce_ms, method='auto', reps=3, frames=6):
    """BASELINE config 5: 64 wavelengths np.linspace(0.5, 0.7, 64) um, uniform weights, n^2 circular pupil with a
    500 nm W040 OPD, fp32.  ONE polychromatic_psf call = this rank's ceil(64 / N) wavelengths (pupil synthesis inside the
    row pass, FFT focus, |.|^2 accumulated by the column pass's epilogue) + one sum-reduce of the image to rank 0.
    `reduce_ms`: reduce_alone_ms() of this run; `method` 'auto' takes the faster form.  Variant F is timed with BOTH reduce forms
    and as a pipelined sequence of `frames` PSFs (PsfPipeline: frame k's reduce on a side stream under frame k + 1's transforms)."""
    from prysm_amd.polychromatic import polychromatic_psf, PsfPipeline
    ax = (torch.arange(n, device='cuda', dtype=torch.float64) - n // 2) * (10.0 / n)
    r = torch.hypot(ax[None, :], ax[:, None])
    amp = (r <= 5).to(torch.float32)
    opd = (500.0 * (r / 5) ** 4).to(torch.float32)
    del r
    wvls = np.linspace(0.5, 0.7, N_WAVELENGTHS)
    wts = np.ones(N_WAVELENGTHS)
    dx = 10.0 / n
    if method == 'auto':
        method = min(('reduce', 'a2a', 'rs'), key=lambda m: reduce_ms.get(m, 1e9)) if ranks.world > 1 else 'reduce'
    forms = {'reduce': 'torch.distributed.reduce(SUM) of the real image to rank 0 (RCCL over xGMI)',
             'a2a': 'all_to_all_single of image slices + ordered local sum (pm_sum_modes) + gather into the root image (RCCL over xGMI)',
             'rs': 'reduce_scatter_tensor + gather into the root image (RCCL over xGMI)'}
    res = {'wavelengths': N_WAVELENGTHS, 'wavelengths_per_gpu': math.ceil(N_WAVELENGTHS / ranks.world), 'pupil': f'{n}x{n} fp32',
           'reduce_method': method if ranks.world > 1 else 'none (one rank)',
           'reduce': forms[method] if ranks.world > 1 else 'none (one rank)'}

    def var_f(m=method):
        polychromatic_psf(amp, opd, wvls, wts, dx, 100.0, Q=1, reduce_to_all=False, reduce_method=m)

    def var_m():
        polychromatic_psf(amp, opd, wvls, wts, dx, 100.0, focal_dx=0.55 * 10 / 4, samples=512, kind='mdft', reduce_to_all=False,
                          reduce_method=method)

    def timed_median(fn, k):
        fn()    # warm: plans, communicator, allocator
        ts = sorted(ranks.timed(fn) for _ in range(k))
        return ts[len(ts) // 2]

    def entry(t):
        return {'psf_ms': t * 1e3, 'psfs_per_s': 1.0 / t, 'wavelengths_per_s': N_WAVELENGTHS / t,
                'per_wavelength_ms_per_gpu': t * 1e3 / math.ceil(N_WAVELENGTHS / ranks.world)}

    from prysm_amd.conf import config
    prec = config.precision
    config.precision = 32     # fp32 maps -> complex64 pupils; at the default precision 64 the executors' bases would be complex128
    try:                      # and promote the whole matrix DFT to fp64 MFMA (numpy's result-type rule)
        def var_c():
            polychromatic_psf(amp, opd, wvls, wts, dx, 100.0, focal_dx=0.55 * 10 / 4, samples=512, kind='czt', reduce_to_all=False,
                              reduce_method=method)

        res['variant_F_fft_focus'] = entry(timed_median(var_f, reps))
        if ranks.world > 1:     # the other reduce form on the same call, for the record
            by = {method: res['variant_F_fft_focus']['psf_ms']}
            for other in ('reduce', 'a2a', 'rs'):
                if other != method:
                    by[other] = timed_median(lambda o=other: var_f(o), reps) * 1e3
            res['variant_F_fft_focus']['psf_ms_by_reduce_method'] = by
        # pipelined: a sequence of PSFs (frames of a time series / forward passes of an optimiser), one in flight behind the next
        pipe = PsfPipeline(wvls, wts, dx, 100.0, Q=1, reduce_to_all=False, reduce_method=method, depth=2, cache_pupil=True)

        def run_frames():
            pend = [pipe.submit(amp, opd) for _ in range(frames)]
            for p_ in pend:
                p_.result()
            pipe.drain()

        tp = timed_median(run_frames, reps)
        res['variant_F_fft_focus'].update({'pipelined_psfs_per_s': frames / tp, 'pipelined_ms_per_psf': tp * 1e3 / frames,
                                           'pipelined_frames': frames})
        res['variant_M_mdft_512'] = entry(timed_median(var_m, reps))
        res['variant_M_czt_512'] = entry(timed_median(var_c, reps))
    finally:
        config.precision = prec
    fl = 8 * 512 * n * (n + 512) * N_WAVELENGTHS
    res['variant_M_mdft_512']['algorithmic_TFLOPs_whole_job'] = fl / (res['variant_M_mdft_512']['psf_ms'] * 1e-3) / 1e12
    res['note'] = ('timed polychromatic_psf calls (barrier + synchronize on both sides, MAX over ranks, median): F = per wavelength '
                   'pupil synthesis + FFT focus (Q = 1) with fused |.|^2 accumulate; M = prepare_executor + matrix-DFT focus to a '
                   '512^2 grid (focal_dx 1.375 um) + |.|^2 accumulate; variant_M_czt_512 = the same grid and image through the chirp-Z '
                   'executor prysm offers beside the matrix DFT (kind="czt": two fused convolution kernels per wavelength instead of two '
                   'GEMMs); all end with the sum-reduce of the image to rank 0.  pipelined_*: the same PSF `pipelined_frames` times through '
                   'PsfPipeline -- the reduce of frame k runs on a side stream while frame k + 1 computes; the (amplitude, OPD) maps are '
                   'packed once per tensor pair, outside the call')
    return res


def polychromatic_2048(ranks, n=2048, reps=5):
    """The same 64-wavelength sum on a 2048^2 pupil (north_star quotes the propagation metric at 2048^2 and 4096^2): the driver's
    default there -- each rank's share as ONE pm_fft2_spectral call, groups of 8 wavelengths per launch pair -- beside the
    per-wavelength loop and the stacked form it replaces at this size."""
    from prysm_amd.polychromatic import polychromatic_psf
    ax = (torch.arange(n, device='cuda', dtype=torch.float64) - n // 2) * (10.0 / n)
    r = torch.hypot(ax[None, :], ax[:, None])
    amp = (r <= 5).to(torch.float32)
    opd = (500.0 * (r / 5) ** 4).to(torch.float32)
    wvls = np.linspace(0.5, 0.7, N_WAVELENGTHS)
    wts = np.ones(N_WAVELENGTHS)
    res = {'wavelengths': N_WAVELENGTHS, 'pupil': f'{n}x{n} fp32', 'Q': 1}
    forms = (('spectral_groups', dict()), ('per_wavelength_loop', dict(spectral=False, batched=False)), ('stacks', dict(batched=True)))
    for name, kw in forms:
        fn = lambda: polychromatic_psf(amp, opd, wvls, wts, 10.0 / n, 100.0, Q=1, reduce_to_all=False, **kw)   # noqa: E731
        fn()
        ts = sorted(ranks.timed(fn) for _ in range(reps))
        t = ts[len(ts) // 2]
        res[name] = {'psf_ms': t * 1e3, 'wavelengths_per_s': N_WAVELENGTHS / t,
                     'per_wavelength_us_per_gpu': t * 1e6 / math.ceil(N_WAVELENGTHS / ranks.world)}
    # bytes per sample and wavelength of the grouped form: 16 + 16 / 8 (csrc/fft_spectral.h)
    per = res['spectral_groups']['per_wavelength_us_per_gpu'] * 1e-6
    res['spectral_groups']['frac_of_hbm_peak'] = 18.0 * n * n / per / 1e9 / HBM_PEAK_GBS
    # the same maps in float64 -- prysm's default precision: complex128 transforms, pupil synthesised in the row load, grouped kernels
    amp64, opd64 = amp.double(), opd.double()
    res['float64_maps'] = {}
    for name, kw in forms[:2]:
        fn = lambda: polychromatic_psf(amp64, opd64, wvls, wts, 10.0 / n, 100.0, Q=1, reduce_to_all=False, **kw)   # noqa: E731
        fn()
        ts = sorted(ranks.timed(fn) for _ in range(3))
        res['float64_maps'][name] = {'psf_ms': ts[1] * 1e3, 'per_wavelength_us_per_gpu': ts[1] * 1e6 / math.ceil(N_WAVELENGTHS / ranks.world)}
    return res


def reduce_alone_ms(ranks, n):
    """The one data-path collective on its own, both root-only forms: sum-reduce of an n^2 fp32 image to rank 0 (median of 5) as
    ONE torch.distributed.reduce and as all-to-all of slices + ordered local sum + gather (polychromatic._reduce_image)."""
    if ranks.world == 1:
        return {'reduce': 0.0, 'a2a': 0.0, 'rs': 0.0}
    from prysm_amd.polychromatic import _reduce_image
    img = torch.ones((n, n), dtype=torch.float32, device='cuda')
    out = {}
    for method in ('reduce', 'a2a', 'rs'):
        fn = lambda: _reduce_image(img, ranks.world, None, False, method, True)   # noqa: E731
        fn()   # warm
        ts = sorted(ranks.timed(fn) for _ in range(5))
        out[method] = ts[2] * 1e3
    return out


XGMI_LINK_GBS = 153.0     # per direction and link, 7 links per GPU (task brief / SURVEY 8e)


def scaling_model(t1_ms, image_bytes, measured=None):
    """What config 5 variant F should do at N = 2, 4, 8 -- printed so the driver's SCALE run can be checked against it.
    compute(N) = t1 / N (contiguous wavelength blocks, no collective inside); the image reduce by its two forms over fully
    connected xGMI: 'a2a' = every rank sends N - 1 slices of S / N over N - 1 distinct links, then the root receives N - 1 reduced
    slices over N - 1 links: 2 (S / N) / BW_link; 'reduce' (RCCL ring) = the image crosses N - 1 links in pipelined chunks, bounded
    below by S / BW_link.  Link efficiency 0.8 assumed.  single-shot efficiency = t1 / (N (t1 / N + reduce)); pipelined = the
    reduce of frame k under the transforms of frame k + 1 (PsfPipeline): t1 / (N max(t1 / N, reduce))."""
    bw = XGMI_LINK_GBS * 0.8 * 1e9
    out = {'t1_ms': t1_ms, 'image_MB': image_bytes / 1e6, 'link_GBps_assumed': XGMI_LINK_GBS * 0.8, 'per_N': {}}
    for N in (2, 4, 8):
        comp = t1_ms / N
        red = {'a2a': 2 * (image_bytes / N) / bw * 1e3 + 0.03, 'reduce': image_bytes / bw * 1e3 + 0.03}
        out['per_N'][str(N)] = {
            'compute_ms': comp, 'reduce_ms_model': red,
            'efficiency_single_shot': {k: t1_ms / (N * (comp + v)) for k, v in red.items()},
            'efficiency_pipelined': {k: t1_ms / (N * max(comp, v)) for k, v in red.items()}}
    if measured:
        out['measured_this_run'] = measured
    return out


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU path)')
    if args.backend == 'nccl' and world > torch.cuda.device_count():
        raise SystemExit(f'bench.py: {world} ranks but {torch.cuda.device_count()} GPU(s): RCCL needs one device per rank')
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))   # RCCL on ROCm
        else:
            dist.init_process_group('gloo')
    ranks = Ranks(world)
    from prysm_amd import propagation as P

    n = args.n
    cdtype = np.complex64 if args.dtype == 'c64' else np.complex128
    es = np.dtype(cdtype).itemsize
    if args.only:   # profiling runs: one other_configs entry, nothing else
        if args.only == 'poly2048':
            res = polychromatic_2048(ranks)
            if rank == 0:
                print(json.dumps({'polychromatic_2048': res}), flush=True)
            return
        if rank == 0:
            print(json.dumps(other_configs(args.only)), flush=True)
        return
    x = torch.from_numpy(make_field(n, cdtype, 4096 + rank)).cuda()
    if world > 1:
        dist.all_reduce(torch.zeros(1, device='cuda'))     # create the communicator outside every timed region
    loop_info = {}
    elapsed, f = propagation_loop(ranks, x, args.steps, args.warmup, info=loop_info, repeats=4)
    del f

    # ---- the headline line, complete as the contract wants it, BEFORE any side measurement
    line = None
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        p1, p2 = kernel_pass_times(x, n)
        two_copies_ms = plain_copy_pair_ms(x)
        dom, dom_ms = ('row_pass', p1) if p1 >= p2 else ('column_pass', p2)
        alg_bytes_kernel = 2.0 * n * n * es           # one pass reads N^2 s and writes N^2 s
        achieved = alg_bytes_kernel / (dom_ms * 1e-3) / 1e9
        alg_bytes_step = 4.0 * n * n * es             # 4 N^2 s per propagation
        traffic, traffic_src = pmc_traffic(dom, n, args.dtype)
        line = {
            'metric': 'pupil->focus FFT propagations per second (PSFs/s), 4096^2',
            'value': value, 'unit': 'propagations/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'c64' if cdtype == np.complex64 else 'c128', 'data': 'synthetic',
            'config': {'workload': f'focus(x, Q=1) on a {n}x{n} {np.dtype(cdtype).name} field resident in HBM '
                                   '(fftshift(fft2(ifftshift(x), norm=ortho)), complex field out)',
                       'fields_per_gpu_per_step': 1, 'parallelism': f'one field/wavelength per GPU x{world}',
                       'backend': ('RCCL (torch.distributed nccl)' if args.backend == 'nccl' else 'gloo') if world > 1 else 'none',
                       'reduce': 'none in the timed region (independent fields); the polychromatic sum-reduce is timed in `polychromatic`'},
            'whole_step_algorithmic_GBps_per_gpu': alg_bytes_step / (ms_step * 1e-3) / 1e9,
            'whole_step_frac_of_hbm_peak': alg_bytes_step / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'whole_step_frac_of_measured_copy_ceiling': alg_bytes_step / (ms_step * 1e-3) / 1e9 / HBM_COPY_CEILING_GBS,
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'frac_of_measured_copy_ceiling': achieved / HBM_COPY_CEILING_GBS,
                         'traffic': traffic, 'traffic_unit': 'bytes per launch',
                         'traffic_source': traffic_src, 'algorithmic_bytes_per_launch': alg_bytes_kernel,
                         'row_pass_ms': p1, 'column_pass_ms': p2, 'passes_over_step': (p1 + p2) / ms_step,
                         'two_plain_copies_ms': two_copies_ms, 'step_over_two_plain_copies': ms_step / two_copies_ms,
                         'note': '2*N^2*s algorithmic bytes per pass / average HIP-event duration of that pass, each pass timed as its own '
                                 'back-to-back loop of 100 launches on the launch stream (pm_fft2_time_passes); passes_over_step = '
                                 '(row + column) / ms_per_step; two_plain_copies_ms = torch copies in -> ws -> out of the same field, '
                                 'measured in this run'},
        }
        line.update(loop_info)
        rep = loop_info.get('repeat_ms') or []
        if rep:
            allr = rep + [ms_step]
            line['repeat_spread'] = (max(allr) - min(allr)) / min(allr)

    # The side measurements below include this code's collectives (config 5).  A hang or a crash there must not cost the run its
    # headline: after --extras-budget seconds rank 0 prints the line it already has and every rank leaves (one JSON line either way).
    printed = threading.Event()

    def give_up():
        if rank == 0 and not printed.is_set():
            printed.set()
            line['extras'] = f'not finished within --extras-budget {args.extras_budget:.0f} s; headline only'
            emit(line)
        os._exit(0)

    dog = threading.Timer(args.extras_budget + (0.0 if rank == 0 else 5.0), give_up)
    dog.daemon = True
    dog.start()

    extra = {}
    if not args.no_poly:
        try:
            x2 = torch.from_numpy(make_field(2048, np.complex64, 2048 + rank)).cuda()
            k2 = max(args.steps, 50)
            t2, _ = propagation_loop(ranks, x2, k2, max(args.warmup, 5))
            del x2, _
            extra['n2048'] = {'value': world * k2 / t2, 'unit': 'propagations/s', 'ms_per_step': t2 / k2 * 1e3, 'steps': k2,
                              'whole_step_frac_of_hbm_peak': 4 * 2048 ** 2 * 8 / (t2 / k2) / 1e9 / HBM_PEAK_GBS,
                              'workload': 'focus(x, Q=1) on a 2048x2048 complex64 field per GPU, timed like the headline'}
            red = reduce_alone_ms(ranks, n)
            extra['polychromatic'] = polychromatic_config5(ranks, n, red, args.reduce_method)
            extra['polychromatic']['reduce_alone_ms'] = red
            extra['reduce_ms'] = red.get(extra['polychromatic']['reduce_method'], red['reduce'])
            f = extra['polychromatic']['variant_F_fft_focus']
            t1 = f['per_wavelength_ms_per_gpu'] * N_WAVELENGTHS      # this run's compute rate scaled to one GPU's 64 wavelengths
            extra['polychromatic']['scaling_model'] = scaling_model(
                t1 if world == 1 else max(f['psf_ms'] - extra['reduce_ms'], 0.0) * world, n * n * 4,
                {'n_gpus': world, 'psf_ms': f['psf_ms'], 'reduce_alone_ms': red, 'pipelined_ms_per_psf': f['pipelined_ms_per_psf']})
            extra['polychromatic_2048'] = polychromatic_2048(ranks)
        except Exception as exc:     # a rank that fails here leaves the others in a collective: the watchdog ends them
            extra['extras_error'] = repr(exc)
            if world > 1:
                if rank == 0:
                    printed.set()
                    line.update(extra)
                    emit(line)
                os._exit(0 if rank == 0 else 1)

    if rank == 0:
        if not args.no_poly:
            # the intensity form of the same step: |focus(x)|^2 with the modulus fused into the column pass (no complex PSF in memory)
            try:
                acc_i = P.focus_intensity(x, 1)
                psf_ms = _event_ms(lambda: P.focus_intensity(x, 1, out=acc_i), 50)
                del acc_i
                line['psf_variant'] = {'ms_per_psf': psf_ms, 'psfs_per_s_per_gpu': 1e3 / psf_ms,
                                       'note': 'focus_intensity(x, 1): the same propagation storing |.|^2 (fp32 image) instead of the complex field'}
            except Exception as exc:
                line['psf_variant'] = {'error': repr(exc)}
        line.update(extra)
        if not args.no_poly and world == 1:
            try:
                line['other_configs'] = other_configs()
            except Exception as exc:   # never lose the headline line to a side measurement
                line['other_configs'] = {'error': repr(exc)}
        if not args.no_cpu_baseline and world == 1:     # reported at N = 1 only (the other ranks would just wait)
            try:
                line['cpu_baseline'] = cpu_baseline(n, cdtype, args.cpu_seconds)
            except Exception as exc:
                line['cpu_baseline'] = {'error': repr(exc)}
        if not printed.is_set():
            printed.set()
            emit(line)
    dog.cancel()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
