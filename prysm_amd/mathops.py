"""Backend plumbing, the counterpart of prysm/mathops.py.

In prysm every array operation goes through the ``np`` / ``fft`` BackendShim
objects (prysm/mathops.py:11-45).  Here the array container is a torch tensor
in MI355X HBM and the arithmetic of the hot path is libprysm_amd.so; this module
provides the conversions a user of prysm's alternate backends expects
(``array_to_true_numpy``, prysm/mathops.py:119-165) and the helpers to move
inputs onto the device.
"""
from numbers import Number

import numpy as _np
import torch

from . import _lib as L

_scalar_types = (Number, _np.generic)


def to_device(*args):
    """Upload numpy arrays (or pass through tensors) to the current MI355X."""
    out = [a if isinstance(a, _scalar_types) or a is None else L.as_device(a) for a in args]
    return out[0] if len(out) == 1 else out


def array_to_true_numpy(*args):
    """Convert device arrays to numpy (needed for plotting, serialization, checking).

    Same contract as prysm.mathops.array_to_true_numpy (prysm/mathops.py:119-165): scalars and
    numpy arrays pass through; one argument returns one array, several return a list.
    """
    if len(args) == 0:
        return
    out = []
    for arg in args:
        if isinstance(arg, _scalar_types) or isinstance(arg, _np.ndarray):
            out.append(arg)
        elif isinstance(arg, torch.Tensor):
            out.append(arg.detach().cpu().numpy())
        elif hasattr(arg, 'get'):
            out.append(arg.get())
        else:
            out.append(_np.array(arg))
    return out[0] if len(out) == 1 else out


def is_power_of_2(value):
    """prysm/mathops.py is_power_of_2."""
    if value == 1:
        return False
    return bool(value and not value & (value - 1))
