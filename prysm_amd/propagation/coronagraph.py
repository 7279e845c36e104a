"""Propagation through Lyot-family / vortex coronagraphs (prysm/propagation/coronagraph.py) -- SURVEY 8(f) rank 2.

Pure compositions of the fixed-sampling executors (MFMA GEMM pairs), pointwise complex multiplies and sums:
the workloads behind prysm's published speed figures.  Same signatures and return conventions as the reference.
"""
import numbers
import operator

import torch

from .. import _lib as L
from .. import _ops
from ..conf import config
from ._kernels import _adjoint_multiply, field_multiply, field_combine
from .dft import focus_dft, focus_dft_adjoint, unfocus_dft, unfocus_dft_adjoint


def _mul(a, b):
    """a * b with the complex x complex and complex x real 2-D cases on the HIP kernels, numpy-style promotion otherwise."""
    return field_multiply(a, b)


def _one_minus(fpm):
    """1 - fpm of Babinet's principle.  Occulter masks are usually boolean (geometry.circle); numpy promotes `1 - bool_array`
    to integers, torch refuses it, so masks that are neither floating nor complex take config.precision first."""
    if isinstance(fpm, numbers.Number):
        return 1 - fpm
    t = L.as_device(fpm)
    if not (t.is_floating_point() or t.is_complex()):
        t = t.to(L.torch_dtype(config.compute_precision))
    return 1 - t


def to_fpm_and_back(wavefunction, fpm, executor, return_more=False):
    """Propagate to a focal plane mask, apply it, and return (coronagraph.py:12-43)."""
    field_at_fpm = focus_dft(wavefunction, executor)
    field_after_fpm = _mul(field_at_fpm, fpm)
    field_at_next_pupil = unfocus_dft(field_after_fpm, executor)
    if return_more:
        return field_at_next_pupil, field_at_fpm, field_after_fpm
    return field_at_next_pupil


def to_fpm_and_back_adjoint(wavefunction, fpm, executor, return_more=False,
                            return_fpm_grad=False, field_at_fpm=None):
    """Apply the adjoint of to_fpm_and_back (coronagraph.py:46-94)."""
    if return_fpm_grad and field_at_fpm is None:
        raise ValueError('return_fpm_grad=True requires field_at_fpm from the forward propagation')
    fpm_t = fpm if isinstance(fpm, numbers.Number) else L.as_device(fpm)
    fpm_is_complex = isinstance(fpm_t, complex) or (isinstance(fpm_t, torch.Tensor) and fpm_t.is_complex())
    Ebbar = unfocus_dft_adjoint(wavefunction, executor)
    if isinstance(fpm_t, numbers.Number):
        intermediate = Ebbar * (fpm_t.conjugate() if isinstance(fpm_t, complex) else fpm_t)
    else:
        intermediate = _adjoint_multiply(Ebbar, fpm_t)
    Eabar = focus_dft_adjoint(intermediate, executor)
    if return_fpm_grad:
        fpm_bar = _adjoint_multiply(Ebbar, field_at_fpm, real=not fpm_is_complex)
    if return_more:
        if return_fpm_grad:
            return Eabar, Ebbar, intermediate, fpm_bar
        return Eabar, Ebbar, intermediate
    elif return_fpm_grad:
        return Eabar, fpm_bar
    return Eabar


def vortex_phase_mask(charge):
    """Focal-plane-mask callable exp(i charge theta) of a vortex coronagraph (coronagraph.py:97-125)."""
    if not isinstance(charge, numbers.Integral):
        raise TypeError(f'charge must be an integer, got {charge!r}; non-integer charge has a branch cut at theta=pi')

    def fpm(xf, yf):
        xf, yf = L.as_device(xf), L.as_device(yf)
        return torch.exp((1j * charge) * torch.atan2(yf, xf))
    return fpm


def prepare_measured_fpm(measurement, dx, center=(0, 0), charge=None, fill=None, order=1):
    """Wrap a measured complex focal-plane-mask map as an fpm(xf, yf) callable (coronagraph.py:128-200).

    The map is resampled on the device at each level's focal grid (pm_sample_map: map_coordinates order 0 | 1,
    mode='nearest'; orders 2 .. 5: pm_spline_prefilter once, then pm_sample_spline); outside the measured extent the mask
    continues as `fill` (scalar or callable), an ideal vortex of `charge`, or 1.
    """
    meas = L.as_field(measurement)
    if not meas.is_complex():
        meas = meas.to(L.cdtype_of(meas))
    if order not in (0, 1, 2, 3, 4, 5):
        raise RuntimeError('spline order not supported')   # scipy.ndimage's error for orders outside 0 .. 5
    if fill is None:
        fill = vortex_phase_mask(charge) if charge is not None else 1.0
    coeff = _ops.spline_prefilter(meas, order) if order >= 2 else None   # once per measured map, as scipy does per call

    def fpm(xf, yf):
        fillv = fill(xf, yf) if callable(fill) else fill
        return _ops.sample_map(meas, dx, center, xf, yf, fill=fillv, order=order, coeff=coeff)
    return fpm


def to_fpm_and_back_multiresolution(wavefunction, fpm, executor, return_more=False):
    """Propagate to a focal plane mask and back at multiple resolutions (coronagraph.py:203-225)."""
    out = None
    fields_at_fpm, fields_after_fpm = [], []
    for ex, win, xf, yf in zip(executor.executors, executor.windows, executor.xf, executor.yf):
        field_at_fpm = focus_dft(wavefunction, ex)
        field_after_fpm = _mul(_mul(field_at_fpm, fpm(xf, yf)), win)
        contribution = unfocus_dft(field_after_fpm, ex)
        out = contribution if out is None else field_combine(operator.add, out, contribution)
        if return_more:
            fields_at_fpm.append(field_at_fpm)
            fields_after_fpm.append(field_after_fpm)
    if return_more:
        return out, fields_at_fpm, fields_after_fpm
    return out


def to_fpm_and_back_multiresolution_adjoint(wavefunction, fpm, executor, return_more=False,
                                            return_fpm_grad=False, field_at_fpm=None):
    """Apply the adjoint of to_fpm_and_back_multiresolution (coronagraph.py:228-305)."""
    if return_fpm_grad and field_at_fpm is None:
        raise ValueError('return_fpm_grad=True requires field_at_fpm from the forward propagation')
    out = None
    Ebbars, intermediates, fpm_bars = [], [], []
    levels = zip(executor.executors, executor.windows, executor.xf, executor.yf)
    for k, (ex, win, xf, yf) in enumerate(levels):
        m = L.as_device(fpm(xf, yf))
        Ebbar = unfocus_dft_adjoint(wavefunction, ex)
        intermediate = _adjoint_multiply(Ebbar, _mul(m, win) if m.is_complex() else m * win)
        contribution = focus_dft_adjoint(intermediate, ex)
        out = contribution if out is None else field_combine(operator.add, out, contribution)
        if return_more:
            Ebbars.append(Ebbar)
            intermediates.append(intermediate)
        if return_fpm_grad:
            fpm_bars.append(_adjoint_multiply(Ebbar, L.as_device(field_at_fpm[k]) * win, real=not m.is_complex()))
    if return_more:
        if return_fpm_grad:
            return out, Ebbars, intermediates, fpm_bars
        return out, Ebbars, intermediates
    elif return_fpm_grad:
        return out, fpm_bars
    return out


def babinet(wavefunction, lyot, fpm, executor, return_more=False):
    """Propagate through a Lyot-style coronagraph using Babinet's principle (coronagraph.py:308-360)."""
    wavefunction = L.as_complex(wavefunction)
    fpm = _one_minus(fpm)
    result = to_fpm_and_back(wavefunction, fpm=fpm, executor=executor, return_more=return_more)
    if return_more:
        field, field_at_fpm, field_after_fpm = result
    else:
        field = result
    if field.dtype != wavefunction.dtype:
        wavefunction = wavefunction.to(field.dtype)
    field_at_lyot = field_combine(operator.sub, wavefunction, field)
    if lyot is not None:
        field_after_lyot = _mul(field_at_lyot, lyot)
    else:
        field_after_lyot = field_at_lyot
    if return_more:
        return field_after_lyot, field_at_fpm, field_after_fpm, field_at_lyot
    return field_after_lyot


def babinet_adjoint(wavefunction, lyot, fpm, executor, field_at_fpm=None,
                    field_at_lyot=None, return_fpm_grad=False, return_lyot_grad=False):
    """Apply the adjoint of babinet (coronagraph.py:363-431)."""
    if return_lyot_grad and field_at_lyot is None:
        raise ValueError('return_lyot_grad=True requires field_at_lyot from the forward propagation')
    lyot_t = None if lyot is None else L.as_device(lyot)
    lyot_is_complex = True if lyot_t is None else lyot_t.is_complex()
    fpm = _one_minus(fpm)
    dbar = L.as_complex(wavefunction)
    cbar = _adjoint_multiply(dbar, lyot_t) if lyot_t is not None else dbar
    if return_fpm_grad:
        abar, fpm_bar = to_fpm_and_back_adjoint(cbar, fpm=fpm, executor=executor, return_fpm_grad=True,
                                                field_at_fpm=field_at_fpm)
    else:
        abar = to_fpm_and_back_adjoint(cbar, fpm=fpm, executor=executor)
    if cbar.dtype != abar.dtype:
        cbar = cbar.to(abar.dtype)
    abar = field_combine(operator.sub, cbar, abar)
    if not (return_fpm_grad or return_lyot_grad):
        return abar
    out = [abar]
    if return_fpm_grad:
        out.append(fpm_bar)
    if return_lyot_grad:
        out.append(_adjoint_multiply(dbar, field_at_lyot, real=not lyot_is_complex))
    return tuple(out)
