"""FFT-based pupil <-> focal propagation (prysm/propagation/fft.py) as ONE fused device call each.

Extension over the reference: every routine here also takes a (batch, rows, cols) stack of fields
(wavelengths / field points of one model) and propagates the whole stack in one launch pair --
small fields (<= 1024^2) are launch- and latency-bound one at a time.

The reference composes pad2d -> ifftshift -> fft2 -> fftshift (five full-array sweeps around the
transform).  Here the zero padding is a load-side window, both shifts are index rotations, the
'ortho' scale rides on the last store and the crop of the adjoints is a store-side window: the
padded array, the shifted copies and the uncropped gradient never exist in memory.
"""
import math

from .. import _lib as L
from .. import _ops
from ._kernels import _padded_shape, _shape_before_pad


def _centered_fft2(x, Q, direction, crop_to=None, synth=None):
    """fftshift(fft2 | ifft2(ifftshift(pad2d(x, Q)), norm='ortho')) [+ crop_center]."""
    x = L.as_field(x)
    if x.dim() not in (2, 3):
        raise ValueError('propagation routines operate on 2-D arrays (or a (batch, rows, cols) stack of them)')
    m, n = x.shape[-2:]
    M, N = _padded_shape((m, n), Q)
    in_off = (math.ceil((M - m) / 2), math.ceil((N - n) / 2))       # pad2d: prysm/fttools.py:88-89
    shift = (M // 2, N // 2)   # ifftshift on the way in, fftshift on the way out (fft.py:24)
    out_shape, out_off = None, (0, 0)
    if crop_to is not None and tuple(crop_to) != (M, N):
        out_shape = tuple(crop_to)
        out_off = (math.ceil((M - crop_to[0]) / 2), math.ceil((N - crop_to[1]) / 2))  # crop_center: fttools.py:122-124
    return _ops.fft2(x, direction=direction, scale=1.0 / math.sqrt(M * N), shape=(M, N), in_off=in_off,
                     in_shift=shift, out_shape=out_shape, out_off=out_off, out_shift=shift, synth=synth)


def focus_from_amp_and_phase(amplitude, opd, k, Q):
    """focus(amplitude * exp(i k opd), Q) with the pupil synthesised inside the transform (no complex pupil in memory);
    the caller checks _ops.synth_supported."""
    return _centered_fft2(opd, Q, -1, synth=(amplitude, k))


def focus(wavefunction, Q):
    """Propagate a pupil plane to a PSF plane (prysm/propagation/fft.py:7-25)."""
    return _centered_fft2(wavefunction, Q, -1)


def focus_adjoint(wavefunction, Q):
    """Adjoint of focus (fft.py:28-45): inverse transform of the gradient, then crop to int(s//Q)."""
    shape = tuple(wavefunction.shape[-2:])
    return _centered_fft2(wavefunction, 1, +1, crop_to=_shape_before_pad(shape, Q))


def unfocus(wavefunction, Q):
    """Propagate a PSF plane to a pupil plane (fft.py:48-65)."""
    return _centered_fft2(wavefunction, Q, +1)


def unfocus_adjoint(wavefunction, Q):
    """Adjoint of unfocus (fft.py:68-85)."""
    shape = tuple(wavefunction.shape[-2:])
    return _centered_fft2(wavefunction, 1, -1, crop_to=_shape_before_pad(shape, Q))


def focus_intensity(wavefunction, Q, out=None, weight=None, synth=None, spectral=None):
    """|focus(wavefunction, Q)|^2 with the modulus fused into the last FFT pass.

    Equivalent to ``Wavefront.focus(...).intensity.data`` (wavefront.py:146-151, 478-504) but the
    complex focal field is never written: the column pass stores re^2 + im^2 directly.  With ``out``
    and ``weight`` the result is accumulated, ``out += weight * |.|^2`` (the incoherent sum of the
    polychromatic recipe).  ``spectral=(k values, weights)`` with ``synth`` and ``out`` runs that whole loop in one call:
    ``out += sum_b weights[b] |focus(amp exp(i k_b opd))|^2`` (pm_fft2_spectral).
    """
    x = L.as_field(wavefunction)
    m, n = x.shape[-2:]
    M, N = _padded_shape((m, n), Q)
    in_off = (math.ceil((M - m) / 2), math.ceil((N - n) / 2))
    shift = (M // 2, N // 2)
    epi = L.PM_EPI_ABS2 if (out is None or (weight is None and spectral is None)) else L.PM_EPI_ABS2_ACCUM
    return _ops.fft2(x, direction=-1, scale=1.0 / math.sqrt(M * N), shape=(M, N), in_off=in_off, in_shift=shift,
                     out_shift=shift, epilogue=epi, out=out, weight=1.0 if weight is None else weight, synth=synth, spectral=spectral)


def Q_for_sampling(input_diameter, prop_dist, wavelength, output_dx):
    """Value of Q for a given output sampling (fft.py:88-109)."""
    resolution_element = (wavelength * prop_dist) / (input_diameter)
    return resolution_element / output_dx


def pupil_sample_to_psf_sample(pupil_sample, samples, wavelength, efl):
    """Convert pupil sample spacing to PSF sample spacing (fft.py:112-132)."""
    return (efl * wavelength) / (pupil_sample * samples)


def psf_sample_to_pupil_sample(psf_sample, samples, wavelength, efl):
    """Convert PSF sample spacing to pupil sample spacing (fft.py:135-155)."""
    return (efl * wavelength) / (psf_sample * samples)
