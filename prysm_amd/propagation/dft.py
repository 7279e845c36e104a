"""Matrix-DFT / chirp-Z / FFT-DFT pupil <-> focal propagation with arbitrary sampling
(prysm/propagation/dft.py).  The executors live in prysm_amd.fttools; MDFT runs its two complex
GEMMs on the MFMA matrix cores.
"""
import math
from collections.abc import Iterable

import torch

from .. import _lib as L
from ..conf import config
from ..fttools import fftrange, MDFT, CZT, FFTDFT


def coordinates_for_focus(pupil_dx, pupil_samples, focal_dx, focal_samples,
                          wavelength, efl, focal_shift=(0, 0)):
    """Coordinate / frequency vectors for an MDFT-based pupil <-> focal propagation (dft.py:12-66).

    Returns x, y (pupil coordinates, mm) and fx, fy (spatial frequencies, 1/mm) as device vectors
    in config.precision.
    """
    if not isinstance(pupil_samples, Iterable):
        pupil_samples = (pupil_samples, pupil_samples)
    if not isinstance(focal_samples, Iterable):
        focal_samples = (focal_samples, focal_samples)
    pny, pnx = pupil_samples
    fny, fnx = focal_samples
    fsx, fsy = focal_shift
    dtype = config.compute_precision
    x = fftrange(pnx, dtype=dtype) * pupil_dx
    y = fftrange(pny, dtype=dtype) * pupil_dx
    inv_lz = 1.0 / (wavelength * efl)
    fx = (fftrange(fnx, dtype=dtype) * focal_dx + fsx) * inv_lz
    fy = (fftrange(fny, dtype=dtype) * focal_dx + fsy) * inv_lz
    return x, y, fx, fy


def prepare_executor(pupil_dx, pupil_samples, focal_dx, focal_samples,
                     wavelength, efl, focal_shift=(0, 0), kind='mdft'):
    """Build a reusable MDFT, CZT, or FFTDFT pupil <-> focal operator (dft.py:69-117).

    norm = pupil_dx * focal_dx / (wavelength * efl) is baked into the executor; pupil_dx and
    focal_dx are stashed on it.  executor(pupil) focuses, executor.adjoint(focal) unfocuses.
    """
    norm = (pupil_dx * focal_dx) / (wavelength * efl)
    if kind == 'mdft':
        # MDFT(*coordinates_for_focus(...)) with the coordinate grids generated inside the basis kernel (two launches per
        # executor instead of a dozen small array operations -- the polychromatic recipe builds one executor per wavelength)
        ps = pupil_samples if isinstance(pupil_samples, Iterable) else (pupil_samples, pupil_samples)
        fs = focal_samples if isinstance(focal_samples, Iterable) else (focal_samples, focal_samples)
        op = MDFT._for_focus_grids(tuple(int(v) for v in ps), tuple(int(v) for v in fs), pupil_dx, focal_dx, focal_shift,
                                   1.0 / (wavelength * efl), L.torch_dtype(config.compute_precision), -1, norm)
        op.pupil_dx = pupil_dx
        op.focal_dx = focal_dx
        return op
    if kind == 'czt':
        # CZT(*coordinates_for_focus(...)) from the grid parameters, like the MDFT above (one kernel + one transform per axis)
        ps = pupil_samples if isinstance(pupil_samples, Iterable) else (pupil_samples, pupil_samples)
        fs = focal_samples if isinstance(focal_samples, Iterable) else (focal_samples, focal_samples)
        if min(int(v) for v in ps) >= 2 and min(int(v) for v in fs) >= 2:
            op = CZT._for_focus_grids(tuple(int(v) for v in ps), tuple(int(v) for v in fs), pupil_dx, focal_dx, focal_shift,
                                      1.0 / (wavelength * efl), L.torch_dtype(config.compute_precision), -1, norm)
            op.pupil_dx = pupil_dx
            op.focal_dx = focal_dx
            return op
    x, y, fx, fy = coordinates_for_focus(pupil_dx, pupil_samples, focal_dx, focal_samples,
                                         wavelength, efl, focal_shift)
    if kind == 'czt':
        op = CZT(x, y, fx, fy, sign=-1, norm=norm)
    elif kind == 'fftdft':
        op = FFTDFT(x, y, fx, fy, sign=-1, norm=norm)
    else:
        raise ValueError(f"kind must be 'mdft', 'czt', or 'fftdft', got {kind!r}")
    op.pupil_dx = pupil_dx
    op.focal_dx = focal_dx
    return op


def focus_fixed_sampling(wavefunction, input_dx, prop_dist, wavelength, output_dx, output_samples,
                         shift=(0, 0), method='mdft'):
    """Pre-0.22 name of the fixed-sampling focus (BASELINE.json uses it; v0.22.rst:163-205).

    Thin alias: builds the executor and applies it.  Prefer prepare_executor + focus_dft, which
    reuses the bases across calls.
    """
    ex = prepare_executor(input_dx, tuple(wavefunction.shape), output_dx, output_samples, wavelength, prop_dist,
                          focal_shift=shift, kind=method)
    return ex(wavefunction)


def unit_cell_focal_grid(pupil_dx, pupil_diameter, wavelength, efl, Q=2):
    """Focal grid (focal_dx, focal_samples) spanning the full DFT unit cell (dft.py:120-152)."""
    focal_samples = math.ceil(Q * pupil_diameter / pupil_dx)
    focal_dx = wavelength * efl / pupil_dx / focal_samples
    return focal_dx, focal_samples


def _smootherstep(t):
    """C2 smoothstep 6t^5 - 15t^4 + 10t^3, clipped to [0, 1] (dft.py:155-158)."""
    t = torch.clamp(t, 0, 1)
    return t * t * t * (t * (t * 6 - 15) + 10)


def _cumulative_window(r, a, b):
    """Radial taper that is 1 for r < a and 0 for r > b, with a C2 transition (dft.py:161-168)."""
    return 1 - _smootherstep((r - a) / (b - a))


class MultiResolutionExecutor:
    """A stack of arbitrary-sampling executors plus partition-of-unity windows (dft.py:171-212).

    Attributes: executors (coarsest first), windows (real, summing to one over the focal plane), xf, yf (per-level focal
    coordinate meshgrids, microns) -- device tensors.
    """

    __slots__ = ('executors', 'windows', 'xf', 'yf')

    def __init__(self, executors, windows, xf, yf):
        self.executors = executors
        self.windows = windows
        self.xf = xf
        self.yf = yf

    def __len__(self):
        return len(self.executors)


def prepare_multiresolution(pupil_dx, pupil_samples, focal_dx, focal_samples,
                            wavelength, efl, num_levels, scaling=4.0,
                            fine_samples=None, window=(0.2, 0.7), kind='mdft'):
    """Build a MultiResolutionExecutor for focal-plane-mask propagation (dft.py:215-294).

    Level k samples the focal plane at focal_dx / scaling**k over a field of view that shrinks by the same factor; every
    level's grid is shifted by half a sample so a mask singularity at the origin is never sampled; the windows telescope
    to a partition of unity.
    """
    if fine_samples is None:
        fine_samples = focal_samples
    inner, outer = window
    executors, xfs, yfs, radii, halves = [], [], [], [], []
    for k in range(num_levels):
        nf = focal_samples if k == 0 else fine_samples
        if not isinstance(nf, Iterable):
            nf = (nf, nf)
        nfy, nfx = nf
        fdx = focal_dx / scaling**k
        shift = fdx / 2.0
        ex = prepare_executor(pupil_dx, pupil_samples, fdx, nf, wavelength, efl, focal_shift=(shift, shift), kind=kind)
        xline = fftrange(nfx, dtype=config.compute_precision) * fdx + shift
        yline = fftrange(nfy, dtype=config.compute_precision) * fdx + shift
        yf, xf = torch.meshgrid(yline, xline, indexing='ij')
        executors.append(ex)
        xfs.append(xf)
        yfs.append(yf)
        radii.append(torch.hypot(xf, yf))
        halves.append(min(nfy, nfx) / 2.0 * fdx)
    windows = []
    for k in range(num_levels):
        r = radii[k]
        here = 1.0 if k == 0 else _cumulative_window(r, inner * halves[k], outer * halves[k])
        nxt = 0.0 if k == num_levels - 1 else _cumulative_window(r, inner * halves[k + 1], outer * halves[k + 1])
        win = here - nxt
        if not isinstance(win, torch.Tensor):
            win = torch.full_like(r, float(win))
        windows.append(win)
    return MultiResolutionExecutor(executors, windows, xfs, yfs)


def focus_dft(wavefunction, executor):
    """Propagate a pupil field to the PSF plane via a precomputed executor (dft.py:297-313)."""
    return executor(wavefunction)


def focus_dft_adjoint(wavefunction, executor):
    """Apply the adjoint of focus_dft (dft.py:316-332)."""
    return executor.adjoint(wavefunction)


def unfocus_dft(wavefunction, executor):
    """Propagate an image-plane field to the pupil (dft.py:335-351) -- the ADJOINT, not the inverse."""
    return executor.adjoint(wavefunction)


def unfocus_dft_adjoint(wavefunction, executor):
    """Apply the adjoint of unfocus_dft (dft.py:354-370)."""
    return executor(wavefunction)
