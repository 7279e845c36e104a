"""MTF / PTF / OTF from a PSF (prysm/otf.py) -- SURVEY 8(f) rank 1, a "next" row of the hot path.

The forward transform fftshift(fft2(ifftshift(psf))) is the same fused pm_fft2 call as `focus`
(unnormalised, shifts folded into the index maps).  A REAL PSF with power-of-two lengths takes the
Hermitian path of the library (half-length row transforms, half the column tiles, mirrored stores)
and, when only the MTF / PTF / OTF is wanted, the centre normalisation and abs / angle ride in the
epilogue of its column pass: one launch pair, no elementwise sweeps.  Other inputs, and
return_more=True (which also hands back the unnormalised transform), compose the transform with
elementwise device ops.
"""
import math
import numbers

import torch

from . import _lib as L
from . import _ops
from ._richdata import RichData


def _center(shape):
    """Pixel index of the (floor) center of a 2D array (prysm/otf.py:11-13)."""
    return tuple(int(math.floor(s / 2)) for s in shape)


def _unwrap_psf(psf, dx):
    """Resolve a PSF container-or-array to a bare array and its sample spacing (otf.py:16-25)."""
    if isinstance(psf, RichData) or (hasattr(psf, 'data') and hasattr(psf, 'dx') and not isinstance(psf, torch.Tensor)):
        dx = psf.dx        # a container's own sampling always wins, as in the reference (otf.py:17-19)
        psf = psf.data
    if dx is None:
        raise ValueError('dx is None: dx must be provided if psf is an array')
    return psf, dx


def transform_psf(psf, dx=None):
    """Transform a PSF to k-space without further modification (otf.py:28-33)."""
    psf, dx = _unwrap_psf(psf, dx)
    x = L.as_field(psf)          # a real PSF is read as it is (PM_FLAG_REAL_INPUT): no complex copy, half the bytes in pass 1
    M, N = x.shape
    shift = (M // 2, N // 2)
    if not x.is_complex() and not _hermitian_ok(x) and _ops.real_pairs_ok(x) and M * N >= 1 << 18:
        data = _ops.fft2_real(x, in_shift=shift, out_shift=shift)      # sizes off the engine: half-size transform + untangle (round 5)
    else:
        data = _ops.fft2(x, direction=-1, scale=1.0, in_shift=shift, out_shift=shift)
    df = 1000 / (data.shape[0] * dx)  # cy/um to cy/mm
    return data, df


def transform_psf_adjoint(data_bar):
    """Adjoint of transform_psf (otf.py:36-59): fftshift(ifft2(ifftshift(.), norm='forward'))."""
    x = L.as_complex(data_bar)
    M, N = x.shape
    shift = (M // 2, N // 2)
    return _ops.fft2(x, direction=+1, scale=1.0, in_shift=shift, out_shift=shift)


def _fused_from_psf(psf, dx, epilogue):
    """|F / F_c|, angle(F / F_c) or F / F_c straight out of the transform (PM_FLAG_NORM_DC + epilogue), or None when the
    library has no Hermitian path for this input (complex PSF, lengths that are not powers of two, ...)."""
    psf, dx = _unwrap_psf(psf, dx)
    x = L.as_field(psf)
    if x.is_complex() or x.dim() != 2:
        return None
    M, N = x.shape
    shift = (M // 2, N // 2)
    if not _hermitian_ok(x):
        # any other size with an even width (1000^2, 3000^2 ...): the array as complex pairs, a half-size transform on its own route and
        # one untangling sweep with the same normalisation and epilogue (round 5, _ops.fft2_real)
        if _ops.real_pairs_ok(x) and x.is_cuda:
            return _ops.fft2_real(x, scale=1.0, in_shift=shift, out_shift=shift, epilogue=epilogue, norm_dc=True), 1000 / (M * dx)
        return None
    try:
        data = _ops.fft2(x, direction=-1, scale=1.0, in_shift=shift, out_shift=shift, epilogue=epilogue, flags=L.PM_FLAG_NORM_DC)
    except NotImplementedError:      # the predicate above mirrors the library's (capi.hip r2c_legal); kept as the backstop
        return None
    return data, 1000 / (M * dx)


def _hermitian_ok(x):
    """Cheap mirror of the library's test for its Hermitian path (capi.hip r2c_legal): a real float32 / float64 2-D array, both
    lengths powers of two from 32 that one workgroup transforms (rows of at most 8192 samples, 4096 for float64), even leading
    dimension, base address aligned like a complex element.  Asked BEFORE building a descriptor: an ineligible PSF (1000^2, a
    sliced view ...) used to pay a failed library call and an exception on every MTF / PTF / OTF call."""
    if x.is_complex() or x.dim() != 2 or x.dtype not in (torch.float32, torch.float64):
        return False
    M, N = x.shape

    def pow2(n):
        return n >= 32 and (n & (n - 1)) == 0

    nmax = 8192 if x.dtype == torch.float32 else 4096
    return (pow2(M) and pow2(N) and M <= 8192 and N <= nmax and x.stride(1) == 1 and x.stride(0) % 2 == 0 and
            x.data_ptr() % (2 * x.element_size()) == 0)


def _normalized_transform(psf, dx):
    """Forward-transform a PSF and divide by its central value (otf.py:62-74)."""
    data, df = transform_psf(psf, dx)
    cy, cx = _center(data.shape)
    normalized = data / data[cy, cx]
    return normalized, data, df


def mtf_from_psf(psf, dx=None, return_more=False):
    """Compute the MTF from a given PSF (otf.py:77-103)."""
    if not return_more:
        fused = _fused_from_psf(psf, dx, L.PM_EPI_ABS)
        if fused is not None:
            return RichData(data=fused[0], dx=fused[1], wavelength=None)
    normalized, data, df = _normalized_transform(psf, dx)
    rd = RichData(data=torch.abs(normalized), dx=df, wavelength=None)
    if return_more:
        return rd, data
    return rd


def ptf_from_psf(psf, dx=None, return_more=False):
    """Compute the PTF from a given PSF (otf.py:106-135)."""
    if not return_more:
        fused = _fused_from_psf(psf, dx, L.PM_EPI_ARG)
        if fused is not None:
            return RichData(data=fused[0], dx=fused[1], wavelength=None)
    normalized, data, df = _normalized_transform(psf, dx)
    rd = RichData(data=torch.angle(normalized), dx=df, wavelength=None)
    if return_more:
        return rd, data
    return rd


def otf_from_psf(psf, dx=None, return_more=False):
    """Compute the OTF from a given PSF (otf.py:138-164)."""
    if not return_more:
        fused = _fused_from_psf(psf, dx, L.PM_EPI_NONE)
        if fused is not None:
            return RichData(data=fused[0], dx=fused[1], wavelength=None)
    normalized, data, df = _normalized_transform(psf, dx)
    rd = RichData(data=normalized, dx=df, wavelength=None)
    if return_more:
        return rd, data
    return rd


def mtf_ptf_otf_from_psf(psf, dx=None, return_more=False):
    """MTF, PTF and OTF with a single forward transform (otf.py:167-203)."""
    if not return_more:
        # the centre-normalised OTF straight out of the Hermitian transform pair, then |.| and the angle in ONE sweep (pm_abs_arg):
        # three launches, no torch arithmetic (round 2: transform + division + abs + angle as four device sweeps)
        fused = _fused_from_psf(psf, dx, L.PM_EPI_NONE)
        if fused is not None:
            a, g = _ops.abs_arg(fused[0])
            return (RichData(data=a, dx=fused[1], wavelength=None), RichData(data=g, dx=fused[1], wavelength=None),
                    RichData(data=fused[0], dx=fused[1], wavelength=None))
    normalized, data, df = _normalized_transform(psf, dx)
    if normalized.dim() == 2 and normalized.is_complex() and normalized.stride(1) == 1:
        a, g = _ops.abs_arg(normalized)
    else:
        a, g = torch.abs(normalized), torch.angle(normalized)
    mtf = RichData(data=a, dx=df, wavelength=None)
    ptf = RichData(data=g, dx=df, wavelength=None)
    otf = RichData(data=normalized, dx=df, wavelength=None)
    if return_more:
        return mtf, ptf, otf, data
    return mtf, ptf, otf


def _forward_data(psf, dx, data):
    if data is None:
        data, _ = transform_psf(psf, dx)
    return L.as_complex(data)


def mtf_from_psf_adjoint(mtf_bar, psf=None, dx=None, data=None):
    """Apply the adjoint of mtf_from_psf (otf.py:205-242): gradient on the centre-normalised MTF -> real PSF."""
    data = _forward_data(psf, dx, data)
    mtf_bar = L.as_device(mtf_bar).to(L._REAL_OF[data.dtype])
    cy, cx = _center(data.shape)
    mag = torch.abs(data)
    a = mag[cy, cx]
    data_bar = mtf_bar * data / mag / a
    S = torch.sum(mtf_bar * mag)
    data_bar[cy, cx] -= S * data[cy, cx] / a ** 3
    return transform_psf_adjoint(data_bar).real


def ptf_from_psf_adjoint(ptf_bar, psf=None, dx=None, data=None):
    """Apply the adjoint of ptf_from_psf (otf.py:245-279)."""
    data = _forward_data(psf, dx, data)
    ptf_bar = L.as_device(ptf_bar).to(L._REAL_OF[data.dtype])
    cy, cx = _center(data.shape)
    msq = data.real * data.real + data.imag * data.imag
    data_bar = ptf_bar * 1j * data / msq
    data_bar[cy, cx] -= torch.sum(ptf_bar) * 1j * data[cy, cx] / msq[cy, cx]
    return transform_psf_adjoint(data_bar).real


def otf_from_psf_adjoint(otf_bar, psf=None, dx=None, data=None):
    """Apply the adjoint of otf_from_psf (otf.py:282-316)."""
    data = _forward_data(psf, dx, data)
    otf_bar = L.as_complex(otf_bar).to(data.dtype)
    cy, cx = _center(data.shape)
    cc = torch.conj(data[cy, cx])
    data_bar = otf_bar / cc
    data_bar[cy, cx] -= torch.sum(torch.conj(data) * otf_bar) / cc ** 2
    return transform_psf_adjoint(data_bar).real


def encircled_energy(psf, dx, radius, return_more=False):
    """Encircled energy of the PSF at one radius or an iterable of radii, microns (otf.py:346-387; Baliga & Cohn 1988).

    One pm_encircled_energy pass over the MTF serves up to 8 radii (the reference re-evaluates the Bessel kernel on the full
    frequency grid per radius).  Returns a 0-d float64 device tensor for a scalar radius, else a vector.
    """
    mtf, data = mtf_from_psf(psf, dx, return_more=True)
    scalar = isinstance(radius, numbers.Number)
    radii = (radius,) if scalar else tuple(radius)
    out = _ops.encircled_energy(mtf.data, mtf.dx, [r / 1e3 for r in radii])
    if scalar:
        out = out[0]
    if return_more:
        return out, data
    return out


def encircled_energy_adjoint(ee_bar, psf=None, dx=None, radius=None, data=None):
    """Apply the adjoint of encircled_energy (otf.py:417-472): gradient on the encircled energies -> real PSF plane."""
    if data is not None:
        shape = tuple(data.shape)
        if dx is None:
            raise ValueError('dx is None: dx must be provided to set the frequency grid')
        dxv = dx
        cd = L.as_complex(data).dtype
    else:
        arr, dxv = _unwrap_psf(psf, dx)
        arr = L.as_field(arr)
        shape = tuple(arr.shape)
        cd = L.cdtype_of(arr)
    df = 1000 / (shape[0] * dxv)  # cy/um to cy/mm; matches transform_psf
    if isinstance(radius, numbers.Number):
        radii, bars = (radius,), (ee_bar,)
    else:
        radii, bars = tuple(radius), ee_bar
    bars = [float(b) for b in (bars.tolist() if hasattr(bars, 'tolist') else bars)]
    mtf_bar = _ops.encircled_energy_adjoint(shape, df, [r / 1e3 for r in radii], bars, L._REAL_OF[cd])
    return mtf_from_psf_adjoint(mtf_bar, psf=psf, dx=dx, data=data)
