"""Padding / adjoint glue shared by the propagation routines (prysm/propagation/_kernels.py)."""
import math
import numbers

import numpy as np
import torch

from .. import _lib as L
from .. import _ops
from ..fttools import pad2d, crop_center
from ..graph import sequenced


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


def _plain2d(t):
    return t.dim() == 2 and t.stride(1) == 1


@sequenced
def field_multiply(a, b):
    """a * b for a complex field `a` and a mask / screen `b` (numpy's promotion rules): the two hot cases run as one sweep of a HIP
    kernel -- complex x complex of one precision (pm_cmul) and complex x real of the matching precision (pm_rmul: focal-plane masks,
    Lyot stops and windows are usually real) -- everything else (scalars, mixed precisions, stacks) is the torch product.  Inside a
    graph.sequence() block the call is dispatched like every array-level entry point, so the torch fallback runs on the stream of
    the field's producer too (round 6: the fallback used to run on the caller's stream, unordered against the ring)."""
    if isinstance(b, numbers.Number):
        return a * b
    b = L.as_device(b)
    if isinstance(a, torch.Tensor) and not a.is_complex() and b.is_complex():
        a, b = b, a              # a real mask times a complex field: the same sweep
    if isinstance(a, torch.Tensor) and a.is_complex() and a.shape == b.shape and _plain2d(a) and _plain2d(b):
        if b.dtype == a.dtype:
            return _ops.cmul(a, b)
        if b.dtype == L._REAL_OF[a.dtype]:
            return _ops.rmul(b, a)
    return a * b


@sequenced
def field_combine(func, a, b):
    """func(a, b) (operator.add / sub / ...) of two fields as a dispatched call: inside a graph.sequence() block the torch operation
    runs on the stream of its producers"""
    return func(a, b)


@sequenced
def _adjoint_multiply(grad, factor, real=False):
    """Adjoint with respect to x for y = x * factor (_kernels.py:29-37)."""
    grad = L.as_device(grad)
    factor = L.as_device(factor)
    same = grad.shape == factor.shape and _plain2d(grad) and _plain2d(factor)
    if factor.is_complex():
        if grad.is_complex() and grad.dtype == factor.dtype and same:
            out = _ops.cmul(grad, factor, conj_b=True)
        else:
            out = grad * factor.conj()
    elif grad.is_complex() and same and factor.dtype == L._REAL_OF[grad.dtype]:
        out = _ops.rmul(factor, grad)
    else:
        out = grad * factor
    if real:
        return out.real if out.is_complex() else out
    return out


def phase_prefix(wavelength):
    """Scale factor such that multiplication with OPD in nm produces radians (_kernels.py:40-43)."""
    return 1j * 2 * np.pi / wavelength / 1e3
