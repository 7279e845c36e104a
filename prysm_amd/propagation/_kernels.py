"""Padding / adjoint glue shared by the propagation routines (prysm/propagation/_kernels.py)."""
import math

import numpy as np

from .. import _lib as L
from .. import _ops
from ..fttools import pad2d, crop_center


def _maybe_pad(wavefunction, Q):
    """Symmetric-pad by factor Q, or pass through if Q == 1 (_kernels.py:7-11)."""
    if Q != 1:
        return pad2d(wavefunction, Q)
    return wavefunction


def _padded_shape(shape, Q):
    """Shape pad2d(x, Q) would produce (prysm/fttools.py:72-75) without materialising it."""
    if Q == 1:
        return tuple(shape)
    return tuple(math.ceil(s * Q) for s in shape)


def _shape_before_pad(padded_shape, Q):
    """Infer the input shape from the padded shape and padding factor (_kernels.py:14-18)."""
    if Q == 1:
        return tuple(padded_shape)
    return tuple(int(s // Q) for s in padded_shape)


def _adjoint_pad2d(array, Q):
    """Apply the adjoint of _maybe_pad(array, Q) (_kernels.py:21-26)."""
    out_shape = _shape_before_pad(array.shape, Q)
    if out_shape != tuple(array.shape):
        return crop_center(array, out_shape)
    return array


def _adjoint_multiply(grad, factor, real=False):
    """Adjoint with respect to x for y = x * factor (_kernels.py:29-37)."""
    grad = L.as_device(grad)
    factor = L.as_device(factor)
    if factor.is_complex():
        if grad.is_complex() and grad.dtype == factor.dtype and grad.dim() == 2 and grad.shape == factor.shape:
            out = _ops.cmul(grad, factor, conj_b=True)
        else:
            out = grad * factor.conj()
    else:
        out = grad * factor
    if real:
        return out.real if out.is_complex() else out
    return out


def phase_prefix(wavelength):
    """Scale factor such that multiplication with OPD in nm produces radians (_kernels.py:40-43)."""
    return 1j * 2 * np.pi / wavelength / 1e3
