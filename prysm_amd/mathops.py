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
from .npfacade import NumpyFacade, _wrap

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
    from .graph import active_sequence
    seq = active_sequence()
    if seq is not None:
        seq.join()      # inside a graph.sequence() block the results live on the ring's streams: a host copy waits for them first
    out = []
    for arg in args:
        if isinstance(arg, _scalar_types) or isinstance(arg, _np.ndarray):
            out.append(arg)
        elif isinstance(arg, torch.Tensor):
            out.append(arg.detach().cpu().numpy())
        elif hasattr(arg, 'get'):
            out.append(arg.get())
        elif isinstance(getattr(arg, 'data', None), (torch.Tensor, _np.ndarray)):
            # a Wavefront / RichData: convert its array.  (numpy would otherwise wrap the object in a 0-d object array
            # and every later arithmetic op would call the object's operators once per element of the other operand.)
            out.append(array_to_true_numpy(arg.data))
        elif isinstance(arg, (list, tuple)):
            out.append(_np.array(arg))
        else:
            raise TypeError(f'cannot convert {type(arg).__name__} to a numpy array')
    return out[0] if len(out) == 1 else out


def is_power_of_2(value):
    """prysm/mathops.py is_power_of_2."""
    if value == 1:
        return False
    return bool(value and not value & (value - 1))


# --------------------------------------------------------------------------------------------------------
# Installing the engine into an existing prysm (the reference's own plug mechanism: rebinding names on its
# modules at run time, as prysm/x/polarization.py:541-553 does for Jones propagation).
# --------------------------------------------------------------------------------------------------------
class FFTFacade:
    """Module-like ``fft`` object for prysm's BackendShim (``prysm.mathops.fft._srcmodule = FFTFacade()``).

    Exactly the surface the hot path touches: fft2 / ifft2 (norm None | 'backward' | 'ortho' | 'forward'),
    fft / ifft (n, axis), fftshift / ifftshift, fftfreq, next_fast_len -- numpy semantics on torch tensors in HBM,
    the transforms on libprysm_amd.so.  This is the UNFUSED level: prysm's own ``fftshift(fft2(ifftshift(x)))`` then
    runs as three device operations; the fused level is the function rebinding of set_backend_to_mi355x().
    """

    def __init__(self, device=None):
        self._device = device       # fftfreq only; the transforms run where the library runs

    @staticmethod
    def _scale(norm, count, inverse):
        if norm in (None, 'backward'):
            return 1.0 / count if inverse else 1.0
        if norm == 'ortho':
            return 1.0 / count ** 0.5
        if norm == 'forward':
            return 1.0 if inverse else 1.0 / count
        raise ValueError(f'invalid norm {norm!r}')

    def fft2(self, x, s=None, axes=(-2, -1), norm=None):
        from . import _ops
        x = L.as_field(x)
        if s is not None or tuple(a % x.dim() for a in axes) != (x.dim() - 2, x.dim() - 1):
            raise NotImplementedError('fft2 facade: last two axes, no resizing')
        M, N = x.shape[-2:]
        return _wrap(_ops.fft2(x, direction=-1, scale=self._scale(norm, M * N, False)))

    def ifft2(self, x, s=None, axes=(-2, -1), norm=None):
        from . import _ops
        x = L.as_field(x)
        if s is not None or tuple(a % x.dim() for a in axes) != (x.dim() - 2, x.dim() - 1):
            raise NotImplementedError('ifft2 facade: last two axes, no resizing')
        M, N = x.shape[-2:]
        return _wrap(_ops.fft2(x, direction=+1, scale=self._scale(norm, M * N, True)))

    def fft(self, x, n=None, axis=-1, norm=None):
        from . import _ops
        x = L.as_complex(x)
        length = n if n is not None else x.shape[axis]
        return _wrap(_ops.fft1(x, n, axis, -1, self._scale(norm, length, False)))

    def ifft(self, x, n=None, axis=-1, norm=None):
        from . import _ops
        x = L.as_complex(x)
        length = n if n is not None else x.shape[axis]
        return _wrap(_ops.fft1(x, n, axis, +1, self._scale(norm, length, True)))

    @staticmethod
    def fftshift(x, axes=None):
        x = L.as_device(x)
        axes = tuple(range(x.dim())) if axes is None else ((axes,) if isinstance(axes, int) else tuple(axes))
        return _wrap(torch.roll(x, [x.shape[a] // 2 for a in axes], axes))

    @staticmethod
    def ifftshift(x, axes=None):
        x = L.as_device(x)
        axes = tuple(range(x.dim())) if axes is None else ((axes,) if isinstance(axes, int) else tuple(axes))
        return _wrap(torch.roll(x, [-(x.shape[a] // 2) for a in axes], axes))

    def fftfreq(self, n, d=1.0):
        dev = self._device if self._device is not None else L.device()
        return _wrap(torch.fft.fftfreq(n, d, dtype=torch.float64, device=dev))

    @staticmethod
    def next_fast_len(n):
        from .fttools import next_fast_len
        return next_fast_len(n)


_PROPAGATION_NAMES = (
    'focus', 'focus_adjoint', 'unfocus', 'unfocus_adjoint',
    'angular_spectrum', 'angular_spectrum_adjoint', 'angular_spectrum_transfer_function',
    'coordinates_for_focus', 'prepare_executor', 'focus_dft', 'focus_dft_adjoint', 'unfocus_dft',
    'unfocus_dft_adjoint', 'to_fpm_and_back', 'to_fpm_and_back_adjoint', 'babinet', 'babinet_adjoint',
)
_FTTOOLS_NAMES = ('pad2d', 'crop_center', 'fftrange', 'MDFT', 'CZT', 'FFTDFT')
_saved = {}


def set_backend_to_mi355x(prysm=None, arrays=False):
    """Route prysm's pupil<->focus / free-space / matrix-DFT hot path to libprysm_amd.so.

    Rebinds the array-level functions and executor classes on ``prysm.propagation`` (and on the submodules and
    ``prysm.propagation.wavefront``, which bind them by name at import, prysm/propagation/wavefront.py:11-25),
    ``prysm.fttools`` and ``prysm.propagation.Wavefront``.  Arrays handed to these functions may be numpy (they are
    uploaded) or torch tensors in HBM; results are device tensors (use ``array_to_true_numpy``).  Undo with
    ``restore_prysm_backend()``.  ``config.precision`` of prysm is mirrored into ``prysm_amd.conf.config``.

    ``arrays=True`` also plugs ``prysm.mathops.np`` (NumpyFacade): prysm's own array code -- coordinates, geometry, the
    Wavefront constructors -- then builds its arrays in HBM and nothing crosses PCIe between model steps.
    """
    import importlib
    if prysm is None:
        prysm = importlib.import_module('prysm')
    from . import propagation as pa_prop, fttools as pa_ft
    from .conf import config as pa_config
    P = importlib.import_module(prysm.__name__ + '.propagation')
    F = importlib.import_module(prysm.__name__ + '.fttools')
    mods = [P] + [importlib.import_module(f'{prysm.__name__}.propagation.{m}')
                  for m in ('fft', 'angular_spectrum', 'dft', 'coronagraph', 'wavefront')]
    for name in _PROPAGATION_NAMES:
        fn = getattr(pa_prop, name)
        for mod in mods:
            if hasattr(mod, name):
                _saved.setdefault((mod, name), getattr(mod, name))
                setattr(mod, name, fn)
    for name in _FTTOOLS_NAMES:
        _saved.setdefault((F, name), getattr(F, name))
        setattr(F, name, getattr(pa_ft, name))
    for mod in (P, mods[-1]):
        _saved.setdefault((mod, 'Wavefront'), mod.Wavefront)
        mod.Wavefront = pa_prop.Wavefront
    try:
        pa_config.precision = importlib.import_module(prysm.__name__ + '.conf').config.precision
    except Exception:   # pragma: no cover
        pass
    # unfused level: prysm's own fft shim (anything that still calls fft.fft2 / fft.fftshift directly, e.g. otf.py)
    try:
        shim = importlib.import_module(prysm.__name__ + '.mathops').fft
        _saved.setdefault((shim, '_srcmodule'), shim._srcmodule)
        shim._srcmodule = FFTFacade()
        if arrays:
            npshim = importlib.import_module(prysm.__name__ + '.mathops').np
            _saved.setdefault((npshim, '_srcmodule'), npshim._srcmodule)
            npshim._srcmodule = NumpyFacade()
    except Exception:   # pragma: no cover
        pass


def restore_prysm_backend():
    """Undo set_backend_to_mi355x()."""
    for (mod, name), obj in _saved.items():
        setattr(mod, name, obj)
    _saved.clear()
