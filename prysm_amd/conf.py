"""Configuration: the single precision switch, mirroring prysm/conf.py:28-96.

``config.precision`` is a numpy real dtype type (np.float32 / np.float64), settable
from 32 / 64 or any dtype-like; ``config.precision_complex`` is derived.  As in the
reference it decides the dtype of everything SYNTHESISED (coordinate vectors,
transfer functions, matrix-DFT bases); arrays passed in keep their own precision.
"""
from numbers import Integral

import numpy as np


def _coerce_real_dtype(precision):
    """prysm/conf.py:7-20."""
    if isinstance(precision, Integral) and not isinstance(precision, bool):
        precision = f'float{precision}'
    try:
        dtype = np.dtype(precision)
    except (TypeError, ValueError) as exc:
        raise ValueError('precision should be a real floating dtype.') from exc
    if dtype.kind != 'f':
        raise ValueError('precision should be a real floating dtype.')
    if dtype.itemsize not in (4, 8):
        raise ValueError('prysm_amd computes in float32 or float64 (complex64 / complex128) only.')
    return dtype.type


class Config:
    """Global configuration (prysm/conf.py:28-93)."""

    def __init__(self, precision=64):
        self.precision = precision

    @property
    def precision(self):
        return self._precision

    @property
    def precision_complex(self):
        return self._precision_complex

    @precision.setter
    def precision(self, precision):
        self._precision = _coerce_real_dtype(precision)
        self._precision_complex = np.result_type(self._precision, 1j).type


config = Config()
