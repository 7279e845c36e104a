"""Configuration: the single precision switch, mirroring prysm/conf.py:28-96.

``config.precision`` is a numpy real dtype type (np.float16 / np.float32 / np.float64), settable
from 16 / 32 / 64 or any dtype-like; ``config.precision_complex`` is derived (complex64 for float16
and float32, as numpy's result_type gives).  As in the reference it decides the dtype of everything
SYNTHESISED (coordinate vectors, transfer functions, matrix-DFT bases); arrays passed in keep their
own precision.

The device kernels compute in float32 / float64 only.  ``precision = 16`` is accepted as the reference
accepts it (prysm/conf.py:7-20, tests/config/test_config.py:29-50) and reports float16 / complex64, but
synthesised REAL vectors are carried in float32 (``config.compute_precision``): the reference would round
its coordinate vectors to half precision first, which this engine does not reproduce -- results at
precision 16 are the precision-32 results.
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
    if dtype.itemsize not in (2, 4, 8):
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

    @property
    def compute_precision(self):
        """Real dtype the device synthesises in: float32 for precision 16 and 32, float64 for 64."""
        return np.float32 if np.dtype(self._precision).itemsize < 8 else np.float64

    @precision.setter
    def precision(self, precision):
        self._precision = _coerce_real_dtype(precision)
        self._precision_complex = np.result_type(self._precision, 1j).type


config = Config()
