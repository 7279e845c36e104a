"""prysm_amd -- an MI355X-native engine for the pupil<->focus / free-space / matrix-DFT hot path of
brandondube/prysm, behind prysm's own Wavefront / propagation / fttools interface.

    from prysm_amd import propagation, fttools
    from prysm_amd.conf import config
    from prysm_amd.mathops import array_to_true_numpy

Arrays are torch tensors in HBM; all hot-path arithmetic runs in hand-written HIP kernels
(libprysm_amd.so, C ABI in include/prysm_amd.h).  There is no CPU fallback.
"""
from . import conf, mathops, fttools, propagation, otf, convolution   # noqa: F401
from .conf import config   # noqa: F401
from .propagation import Wavefront   # noqa: F401
from .npfacade import NumpyFacade, DeviceArray   # noqa: F401

__version__ = '0.1.0'
