"""Helpers shared by the GPU parity tests (tests/test_gpu_*.py): tolerances, host conversion, seeded inputs, numpy restatements."""
import ctypes
import json
import math
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

from conftest import ROOT, rel_max
from oracle import prysm_oracle as O


TOL64 = 1e-10


TOL32 = 5e-6


TOL32_MDFT = 3e-5


def tonp(x):
    from prysm_amd.mathops import array_to_true_numpy
    if hasattr(x, 'data') and not isinstance(x, (np.ndarray, torch.Tensor)):
        x = x.data
    return array_to_true_numpy(x)


def _real_vdot(a, b):
    return float(np.real(np.vdot(np.asarray(a), np.asarray(b))))


def crandn_(rng, shape, dtype):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)


def _np_transform_psf(psf):
    return np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(psf.astype(np.float64))))


def _two_rank_backend():
    return 'nccl' if torch.cuda.device_count() >= 2 else 'gloo'


def _env():
    env = dict(os.environ)
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    return env


def _spectral_case(rng, m, n):
    amp = torch.from_numpy((rng.random((m, n)) > 0.25).astype(np.float32)).cuda()
    opd = torch.from_numpy((200 * rng.standard_normal((m, n))).astype(np.float32)).cuda()
    return amp, opd


def crandn(rng, shape, dtype=np.complex128):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dtype)


def _op_np(a, op):
    if op & 1:
        a = np.conj(a)
    if op & 2:
        a = a.T
    return a


def _poly_numpy(amp, opd, ks, wts, Q):
    """sum_b w_b |focus(amp exp(i k_b opd), Q)|^2 in fp64 (the how-to's loop: Polychromatic Propagation.ipynb cell 3)"""
    acc = 0.0
    for k, w in zip(ks, wts):
        acc = acc + w * O.intensity(O.focus(amp.astype(np.float64) * np.exp(1j * k * opd.astype(np.float64)), Q))
    return acc


def _seven_planes(P, amp, opd, wvl):
    """a small relay written as plain Wavefront code: pupil -> focus -> stop -> back -> free space -> focus -> intensity"""
    wf = P.Wavefront.from_amp_and_phase(amp, opd, wvl, 0.04)
    psf = wf.focus(100.0, Q=1)
    back = psf.unfocus(100.0, Q=1)
    back = back * P.Wavefront(amp.to(back.data.dtype), wvl, back.dx)
    moved = back.free_space(dz=5.0, Q=1)
    return moved.focus(100.0, Q=1).intensity.data


# ---------------------------------------------------------------------------
# composite register engine (csrc/fft_ce.h): every built plan against numpy fp64 and against the general mixed-radix kernel
# ---------------------------------------------------------------------------
CE_LENGTHS = [384, 500, 768, 900, 1000, 1152, 1280, 1500, 1536, 1600, 1800, 2000, 2304, 2500, 2560, 3000, 3072, 3600, 4000, 4500, 5000, 5120, 6000, 6144, 8000]


def _ce_ref(x, shape, in_off, in_shift, out_shift, direction):
    M, N = shape
    full = np.zeros((M, N), dtype=np.complex128)
    full[in_off[0]:in_off[0] + x.shape[0], in_off[1]:in_off[1] + x.shape[1]] = x
    full = np.roll(full, (-in_shift[0], -in_shift[1]), (0, 1))
    f = np.fft.fft2(full) if direction < 0 else np.fft.ifft2(full) * (M * N)
    return np.roll(f, out_shift, (0, 1))
