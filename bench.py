#!/usr/bin/env python
"""Headline benchmark: pupil -> focus FFT propagations per second at 4096^2 complex64.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE propagation ``focus(x, Q=1)`` = fftshift(fft2(ifftshift(x), norm='ortho')) of a
synthetic 4096 x 4096 complex64 field already resident in HBM (BASELINE.json metric; config
"4096^2 pupil->focus").  Each rank (one process per GPU) propagates its own field: the path
shards over wavelengths / fields with no data-path collective (weak scaling), so the timed region
is exactly K propagations per rank.  The one real exchange of the polychromatic recipe -- |E|^2 and a
sum all-reduce of the 67 MB fp32 image over RCCL -- happens once per polychromatic PSF, not per
propagation; it is run and timed AFTER the timed region (``reduce_ms``, median of 3) and folded into
``polychromatic.psf_64wvl_ms``, the time of BASELINE config 5 (64 wavelengths over the N ranks, each = pupil
synthesis + focus with the fused |.|^2 accumulate, measured on rank 0, + one reduce).

Rank 0 prints ONE JSON line.  ``roofline``: the dominant kernel (the slower of the two FFT passes),
its duration measured with HIP events recorded between the kernels on the launch stream, in
sequence; achieved = 2 N^2 s algorithmic bytes (read + write of one pass) / duration, peak 8 TB/s.
``cpu_baseline``: the CPU oracle (a numpy/scipy restatement of prysm's focus, scipy.fft
single-threaded exactly as prysm ships it) timed on this box's host cores on the same workload.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_EDGE = 4096
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)
HBM_COPY_CEILING_GBS = 6290.0   # measured float4-copy ceiling quoted by the same guide (report both fractions)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--n', type=int, default=N_EDGE, help='transform edge (default 4096, the headline)')
    ap.add_argument('--dtype', default='c64', choices=['c64', 'c128'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-poly', action='store_true', help='skip the polychromatic per-wavelength measurement (profiling runs)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget for the CPU baseline sample')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='torch.distributed backend (nccl = RCCL; gloo only to exercise the N > 1 path on one GPU)')
    return ap.parse_args()


def make_field(n, cdtype, seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))).astype(cdtype)
    return x


def kernel_pass_times(x, n, reps=20):
    """Average duration (ms) of the row pass and of the column pass, HIP events on the launch stream."""
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


def pmc_traffic(kernel, n, dtype_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary of THIS command
    (profiles/pmc_bench_summary.json: separate --pmc FETCH_SIZE and WRITE_SIZE passes, FETCH_SIZE doubled per
    the gfx950 correction in MI355X_MICROARCH.md).  Counters cannot be read from inside the process."""
    path = os.path.join(ROOT, 'profiles', 'pmc_bench_summary.json')
    if not os.path.exists(path):
        return None, None
    try:
        tab = json.load(open(path))
    except Exception:
        return None, None
    want = f"fft_{kernel}"
    real = 'float' if dtype_name == 'c64' else 'double'
    # the folded column pass runs kernels of n/2 points (two planes per launch), the row pass kernels of n points
    for nn in (n, n // 2):
        for k, v in tab.items():
            if k.startswith(want) and k.endswith(f'_{real}_N{nn}'):
                return v['hbm_traffic_bytes'], f'profiles/pmc_bench_summary.json:{k}'
    return None, None


def polychromatic_per_wavelength_ms(n, cdtype, reps=8):
    """One wavelength of BASELINE config 5, variant F, as the driver runs it per GPU: pupil synthesis
    (from_amp_and_phase of a circular amplitude and a W040 OPD map) + FFT focus with the fused |.|^2 accumulate."""
    from prysm_amd import propagation as P
    rdt = torch.float32 if cdtype == np.complex64 else torch.float64
    ax = (torch.arange(n, device='cuda', dtype=torch.float64) - n // 2) * (10.0 / n)
    r = torch.hypot(ax[None, :], ax[:, None])
    amp = (r <= 5).to(rdt)
    opd = (500.0 * (r / 5) ** 4).to(rdt)
    acc = torch.zeros((n, n), dtype=rdt, device='cuda')

    def one(wvl):
        wf = P.Wavefront.from_amp_and_phase(amp, opd, wvl, 10.0 / n)
        fus = wf._fusable(1)     # complex64: the pupil is synthesised inside the row pass, never written
        if fus is not None:
            P.focus_intensity(fus[1], 1, out=acc, weight=1.0, synth=(fus[0], fus[2]))
        else:
            P.focus_intensity(wf.data, 1, out=acc, weight=1.0)

    for k in range(2):
        one(0.5 + 0.01 * k)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        one(0.5 + 0.2 * k / 63)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _event_ms(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def other_configs():
    """BASELINE configs 2 - 4 measured in the same run (parity-tested cases, reported here for the roofline the north star
    asks for; not the headline value): algorithmic bytes / flops per SURVEY 8(d) over HIP-event time on the launch stream."""
    from prysm_amd import propagation as P
    from prysm_amd.conf import config
    out = {}
    # config 2: 2048^2 complex64 focus, 4 N^2 s bytes
    x2 = torch.from_numpy(make_field(2048, np.complex64, 2048)).cuda()
    ms = _event_ms(lambda: P.focus(x2, 1), 100)
    out['config2_focus_2048_c64'] = {'ms': ms, 'algorithmic_GBps': 4 * 2048 ** 2 * 8 / (ms * 1e-3) / 1e9,
                                     'frac_of_hbm_peak': 4 * 2048 ** 2 * 8 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del x2
    # config 3: 4096^2 complex128 angular-spectrum step (fused 3 passes), graded on 8 N^2 s bytes
    x3 = torch.from_numpy(make_field(4096, np.complex128, 4096)).cuda()
    ms = _event_ms(lambda: P.angular_spectrum(x3, 0.6328, 0.01, 10.0, Q=1), 30)
    b = 8 * 4096 ** 2 * 16
    out['config3_angular_spectrum_4096_c128'] = {'ms': ms, 'algorithmic_GBps': b / (ms * 1e-3) / 1e9,
                                                 'frac_of_hbm_peak': b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 'note': 'graded on 8 N^2 s (two 2-D transforms); the fused chain moves 6 N^2 s'}
    del x3
    # config 4: matrix-DFT focus 2048^2 -> 512^2 complex64 on MFMA, 8 My Nx (Ny + Mx) real flops
    prec = config.precision
    try:
        config.precision = 32
        x4 = torch.from_numpy(make_field(2048, np.complex64, 2048)).cuda()
        ex = P.prepare_executor(10 / 2048, (2048, 2048), 0.6328 * 10 / 8, (512, 512), 0.6328, 100.0)
        ms = _event_ms(lambda: P.focus_dft(x4, ex), 50)
    finally:
        config.precision = prec
    fl = 8 * 512 * 2048 * (2048 + 512)
    out['config4_mdft_2048_to_512_c64'] = {'ms': ms, 'algorithmic_TFLOPs': fl / (ms * 1e-3) / 1e12,
                                           'frac_of_f32_mfma_peak': fl / (ms * 1e-3) / 1e12 / 157.3, 'bound': 'mfma',
                                           'note': 'two complex GEMMs on v_mfma_f32_32x32x2_f32; peak 157.3 TFLOP/s (MI355X_MICROARCH.md)'}
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
    out = {
        'value': 1.0 / med, 'unit': 'propagations/s', 'cores': 1, 'kind': 'port',
        'sample': f'{len(times)} x oracle.focus({n}x{n} {np.dtype(cdtype).name}, Q=1), median {med * 1e3:.1f} ms, '
                  f'min {min(times) * 1e3:.1f} ms; scipy.fft workers=1 (as prysm ships), host has {os.cpu_count()} cores',
    }
    # second arm, informational: the knob prysm's docs recommend (scipy.fft.set_workers), all host cores
    try:
        from scipy import fft as sfft
        workers = os.cpu_count() or 1
        with sfft.set_workers(workers):
            O.focus(x, 1)
            t2 = []
            for _ in range(5):
                t0 = time.perf_counter()
                O.focus(x, 1)
                t2.append(time.perf_counter() - t0)
        out['tuned'] = {'value': 1.0 / float(np.median(t2)), 'cores': workers,
                        'sample': f'5 x the same call under scipy.fft.set_workers({workers}), median {np.median(t2) * 1e3:.1f} ms'}
    except Exception as exc:   # pragma: no cover
        out['tuned'] = {'error': repr(exc)}
    return out


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU path)')
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))   # RCCL on ROCm
        else:
            dist.init_process_group('gloo')
    from prysm_amd import propagation as P
    from prysm_amd import _ops

    n = args.n
    cdtype = np.complex64 if args.dtype == 'c64' else np.complex128
    es = np.dtype(cdtype).itemsize
    x = torch.from_numpy(make_field(n, cdtype, 4096 + rank)).cuda()
    acc = None

    def step():
        return P.focus(x, 1)

    for _ in range(args.warmup):
        f = step()
    if world > 1:
        acc = _ops.abs2(f)
        dist.all_reduce(acc)     # warm the communicator
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        f = None       # release the previous focal field first: the caching allocator then hands the same block
        f = step()     # back, so the steady state touches in + workspace + out (not two alternating outputs)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    reduce_ms = 0.0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the incoherent sum over wavelengths / fields: |E|^2 + one RCCL all-reduce over xGMI, outside the timed region
        samples = []
        for _ in range(3):
            dist.barrier()
            ev0 = torch.cuda.Event(enable_timing=True)
            ev1 = torch.cuda.Event(enable_timing=True)
            ev0.record()
            acc = _ops.abs2(f, out=acc)
            dist.all_reduce(acc)
            ev1.record()
            torch.cuda.synchronize()
            samples.append(ev0.elapsed_time(ev1))
        r = torch.tensor([sorted(samples)[1]], dtype=torch.float64, device='cuda')
        dist.all_reduce(r, op=dist.ReduceOp.MAX)
        reduce_ms = float(r.item())

    poly_ms = polychromatic_per_wavelength_ms(n, cdtype) if (rank == 0 and not args.no_poly) else 0.0
    psf_ms = 0.0
    if rank == 0 and not args.no_poly:
        # the intensity form of the same step: |focus(x)|^2 with the modulus fused into the column pass (no complex PSF in memory)
        acc_i = P.focus_intensity(x, 1)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            P.focus_intensity(x, 1, out=acc_i)
        e1.record()
        torch.cuda.synchronize()
        psf_ms = e0.elapsed_time(e1) / 50
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        p1, p2 = kernel_pass_times(x, n)
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
                       'reduce': 'none in the timed region (independent fields); the polychromatic sum-reduce is timed separately'},
            'whole_step_algorithmic_GBps_per_gpu': alg_bytes_step / (ms_step * 1e-3) / 1e9,
            'whole_step_frac_of_hbm_peak': alg_bytes_step / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'whole_step_frac_of_measured_copy_ceiling': alg_bytes_step / (ms_step * 1e-3) / 1e9 / HBM_COPY_CEILING_GBS,
            'psf_variant': {'ms_per_psf': psf_ms, 'psfs_per_s_per_gpu': (1e3 / psf_ms) if psf_ms else None,
                            'note': 'focus_intensity(x, 1): the same propagation storing |.|^2 (fp32 image) instead of the complex field'},
            'reduce_ms': reduce_ms,
            'polychromatic': {'per_wavelength_ms': poly_ms, 'wavelengths_per_gpu': math.ceil(64 / world),
                              'psf_64wvl_ms': math.ceil(64 / world) * poly_ms + reduce_ms,
                              'note': 'BASELINE config 5 variant F: per wavelength = pupil synthesis + FFT focus with fused '
                                      '|.|^2 accumulate (measured on rank 0 after the timed region), plus one sum-reduce'},
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'frac_of_measured_copy_ceiling': achieved / HBM_COPY_CEILING_GBS,
                         'traffic': traffic, 'traffic_unit': 'bytes per launch',
                         'traffic_source': traffic_src, 'algorithmic_bytes_per_launch': alg_bytes_kernel,
                         'row_pass_ms': p1, 'column_pass_ms': p2,
                         'note': '2*N^2*s algorithmic bytes per pass / HIP-event duration of that pass, in sequence'},
        }
        if not args.no_poly and world == 1:
            try:
                line['other_configs'] = other_configs()
            except Exception as exc:   # never lose the headline line to a side measurement
                line['other_configs'] = {'error': repr(exc)}
        if not args.no_cpu_baseline and world == 1:     # reported at N = 1 only (the other ranks would just wait)
            line['cpu_baseline'] = cpu_baseline(n, cdtype, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
