"""Array-level plug for prysm's ``np`` BackendShim (prysm/mathops.py:11-45) -- SURVEY 8(b), "np surface touched on the path".

``prysm.mathops.np._srcmodule = NumpyFacade()`` makes prysm's own array code (coordinates, geometry, pad2d, the
Wavefront constructors, angular_spectrum_transfer_function ...) build its arrays in MI355X HBM: the facade answers the
numpy names the hot-path files use with torch operations on the device and hands back ``DeviceArray`` -- a torch.Tensor
that also answers the few ndarray methods prysm calls (``astype``, ``copy``, ``get``).  No arithmetic of the path lives
here: transforms go through FFTFacade / the rebinding of set_backend_to_mi355x() to libprysm_amd.so.  Names the
facade does not provide raise AttributeError (loud, like a missing cupy function) instead of computing on the host.

Stock torch as ``_srcmodule`` fails on ``Tensor.astype`` (angular_spectrum.py:107), ``arange(..., dtype=<numpy type>)``
(fttools.py:15) and ``np.dtype`` (conf.py:13); those are the cases this module exists for.
"""
from numbers import Number

import numpy as _np
import torch

from . import _lib as L

_TORCH_DTYPE = {
    _np.dtype('float32'): torch.float32, _np.dtype('float64'): torch.float64, _np.dtype('float16'): torch.float16,
    _np.dtype('complex64'): torch.complex64, _np.dtype('complex128'): torch.complex128,
    _np.dtype('int8'): torch.int8, _np.dtype('int16'): torch.int16, _np.dtype('int32'): torch.int32,
    _np.dtype('int64'): torch.int64, _np.dtype('uint8'): torch.uint8, _np.dtype('bool'): torch.bool,
}
_NUMPY_DTYPE = {v: k for k, v in _TORCH_DTYPE.items()}


def torch_dtype(dt):
    """numpy dtype / type / string / torch dtype -> torch dtype (None passes through)."""
    if dt is None or isinstance(dt, torch.dtype):
        return dt
    if dt is float:
        return torch.float64
    if dt is complex:
        return torch.complex128
    if dt is int:
        return torch.int64
    if dt is bool:
        return torch.bool
    try:
        return _TORCH_DTYPE[_np.dtype(dt)]
    except KeyError:
        raise TypeError(f'dtype {dt!r} has no device counterpart') from None


def numpy_dtype(dt):
    """torch dtype (or anything numpy understands) -> numpy dtype."""
    return _NUMPY_DTYPE[dt] if isinstance(dt, torch.dtype) else _np.dtype(dt)


class DeviceArray(torch.Tensor):
    """torch.Tensor in HBM that also answers the ndarray methods prysm's hot path calls."""

    __array_priority__ = 1000      # numpy operands defer to this type's reflected operators
    __array_ufunc__ = None

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        # numpy arrays mixed into tensor arithmetic (x_numpy * field) are uploaded beside the tensor operand
        if any(isinstance(a, _np.ndarray) for a in args):
            with torch._C.DisableTorchFunctionSubclass():
                dev = next((a.device for a in args if isinstance(a, torch.Tensor)), None)
                if dev is not None:
                    args = tuple(torch.as_tensor(a, device=dev) if isinstance(a, _np.ndarray) else a for a in args)
        if func in _MATMULS and len(args) == 2 and not kwargs:
            with torch._C.DisableTorchFunctionSubclass():
                a, b = (args[1], args[0]) if func is torch.Tensor.__rmatmul__ else args
                out = _device_matmul(a, b)
            if out is not None:
                return out
        return super().__torch_function__(func, types, args, kwargs or {})

    def astype(self, dtype, copy=True):
        dt = torch_dtype(dtype)
        if dt == self.dtype and not copy:
            return self
        if self.is_complex() and not dt.is_complex:
            return self.real.to(dt)            # numpy discards the imaginary part (with a warning)
        return self.to(dt, copy=copy)

    def copy(self):
        return self.clone()

    def get(self):
        """Download to a numpy array (the cupy spelling; array_to_true_numpy uses it)."""
        return self.detach().cpu().as_subclass(torch.Tensor).resolve_conj().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype)


def _binary(name):
    base = getattr(torch.Tensor, name)

    def op(self, other):
        if isinstance(other, _np.ndarray):       # torch's operators refuse ndarray operands; upload beside the tensor
            other = torch.as_tensor(other, device=self.device)
        return base(self, other)
    op.__name__ = name
    return op


for _name in ('add', 'sub', 'mul', 'truediv', 'pow', 'matmul', 'floordiv', 'mod'):
    setattr(DeviceArray, f'__{_name}__', _binary(f'__{_name}__'))
    setattr(DeviceArray, f'__r{_name}__', _binary(f'__r{_name}__'))
for _name in ('lt', 'le', 'gt', 'ge', 'eq', 'ne', 'and', 'or', 'xor', 'iadd', 'isub', 'imul', 'itruediv'):
    setattr(DeviceArray, f'__{_name}__', _binary(f'__{_name}__'))
DeviceArray.__hash__ = torch.Tensor.__hash__


_MATMULS = (torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__, torch.Tensor.__rmatmul__, torch.mm)


def _gemm_operand(t):
    """(row-major storage, op code) of a 2-D complex operand: op bit 0 = conj (torch's lazy conj view), bit 1 = transpose."""
    op = 0
    if t.is_conj():
        t, op = torch.conj(t), 1          # drop the lazy flag: the stored bytes, conj applied by the GEMM
    if t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t, op
    if t.stride(0) == 1 and t.stride(1) >= t.shape[0]:
        return t.T, op | 2
    return t.contiguous(), op


def _device_matmul(a, b):
    """``Ey @ ary @ Ex.T`` of the matrix DFT (prysm/fttools.py:201-228) written with the array operators: complex 2-D
    products on the MI355X go to pm_cgemm (transposed / conjugated views are operand flags, not copies)."""
    if not (isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.is_cuda and b.is_cuda):
        return None
    if a.dim() != 2 or b.dim() != 2 or not (a.is_complex() or b.is_complex()):
        return None
    from . import _ops
    dt = torch.promote_types(a.dtype, b.dtype)
    a = a.as_subclass(torch.Tensor).to(dt)
    b = b.as_subclass(torch.Tensor).to(dt)
    (sa, opa), (sb, opb) = _gemm_operand(a), _gemm_operand(b)
    return _wrap(_ops.cgemm(sa, sb, opa, opb))


def _wrap(t):
    if isinstance(t, torch.Tensor) and not isinstance(t, DeviceArray):
        return t.as_subclass(DeviceArray)
    if isinstance(t, (tuple, list)) and t and isinstance(t[0], torch.Tensor):
        return type(t)(_wrap(v) for v in t)
    return t


def _is_scalar(a):
    return isinstance(a, (Number, _np.generic)) or a is None


# numpy name -> torch name, for elementwise functions with identical argument meaning
_ELEMENTWISE = {
    'exp': 'exp', 'log': 'log', 'sqrt': 'sqrt', 'sin': 'sin', 'cos': 'cos', 'tan': 'tan', 'arccos': 'acos', 'arcsin': 'asin',
    'arctan': 'atan', 'arctan2': 'atan2', 'hypot': 'hypot', 'abs': 'abs', 'absolute': 'abs', 'floor': 'floor', 'ceil': 'ceil',
    'isnan': 'isnan', 'isfinite': 'isfinite', 'sign': 'sign', 'radians': 'deg2rad', 'degrees': 'rad2deg', 'conj': 'conj',
    'conjugate': 'conj', 'angle': 'angle', 'maximum': 'maximum', 'minimum': 'minimum', 'sinc': 'sinc', 'square': 'square',
}
_PASS_THROUGH = ('pi', 'e', 'nan', 'inf', 'newaxis', 'float16', 'float32', 'float64', 'complex64', 'complex128', 'int8',
                 'int16', 'int32', 'int64', 'uint8', 'bool_', 'dtype', 'generic', 'ndarray', 'integer', 'floating',
                 'complexfloating', 'number', 'isscalar')


class NumpyFacade:
    """Module-like ``np`` for prysm's BackendShim: numpy semantics, arrays in MI355X HBM.

    ``device`` is the torch device the arrays live on; the default is the current MI355X and raises without one.
    (The CPU test-suite passes ``torch.device('cpu')`` to check the numpy semantics of this plumbing where no GPU
    exists; the transforms themselves have no such switch.)
    """

    def __init__(self, device=None):
        self._device = device

    # ------------------------------------------------------------------ plumbing
    @property
    def device(self):
        return self._device if self._device is not None else L.device()

    def _t(self, a, dtype=None):
        """Operand -> tensor on the device."""
        if isinstance(a, torch.Tensor):
            t = a if a.device == self.device else a.to(self.device)
        elif isinstance(a, _np.ndarray):
            t = torch.from_numpy(_np.ascontiguousarray(a)).to(self.device)
        elif isinstance(a, (list, tuple)) and any(isinstance(v, torch.Tensor) for v in a):
            t = torch.stack([self._t(v) for v in a])
        else:
            t = torch.as_tensor(_np.asarray(a), device=self.device)
        return t if dtype is None else t.to(torch_dtype(dtype))

    def __getattr__(self, name):
        if name in _PASS_THROUGH:
            return getattr(_np, name)
        if name in _ELEMENTWISE:
            fn = getattr(torch, _ELEMENTWISE[name])
            npfn = getattr(_np, name)

            def elementwise(*args, **kwargs):
                if all(_is_scalar(a) for a in args):
                    return npfn(*args, **kwargs)           # host scalars stay host scalars, as in numpy
                ref = next(a for a in args if not _is_scalar(a))
                ref = self._t(ref)
                ops = []
                for a in args:
                    if _is_scalar(a):
                        dt = ref.dtype if ref.dtype.is_floating_point or ref.dtype.is_complex else torch.float64
                        if isinstance(a, complex) and not dt.is_complex:
                            dt = torch.complex128 if dt == torch.float64 else torch.complex64
                        a = torch.as_tensor(a, dtype=dt, device=self.device)
                    else:
                        a = self._t(a)
                    ops.append(a)
                if name in ('exp', 'log', 'sqrt', 'sin', 'cos', 'tan', 'arccos', 'arcsin', 'arctan', 'radians', 'degrees', 'sinc'):
                    ops = [o.to(torch.float64) if not (o.dtype.is_floating_point or o.dtype.is_complex) else o for o in ops]
                return _wrap(fn(*ops))
            elementwise.__name__ = name
            return elementwise
        raise AttributeError(f'prysm_amd.NumpyFacade has no {name!r}: outside the hot path surface (SURVEY 8b); '
                             'convert with array_to_true_numpy and use numpy')

    # ------------------------------------------------------------------ creation
    def arange(self, *args, dtype=None, **kw):
        if dtype is None:
            dtype = _np.result_type(*[a for a in args]) if not all(isinstance(a, int) for a in args) else _np.int64
            if _np.dtype(dtype).kind == 'f':
                dtype = _np.float64
        return _wrap(torch.arange(*args, dtype=torch_dtype(dtype), device=self.device))

    def linspace(self, start, stop, num=50, endpoint=True, dtype=None):
        if not endpoint:
            step = (stop - start) / num
            stop = stop - step
        return _wrap(torch.linspace(start, stop, int(num), dtype=torch_dtype(dtype) or torch.float64, device=self.device))

    def _shape(self, shape):
        return (int(shape),) if isinstance(shape, (int, _np.integer)) else tuple(int(s) for s in shape)

    def zeros(self, shape, dtype=None):
        return _wrap(torch.zeros(self._shape(shape), dtype=torch_dtype(dtype) or torch.float64, device=self.device))

    def ones(self, shape, dtype=None):
        return _wrap(torch.ones(self._shape(shape), dtype=torch_dtype(dtype) or torch.float64, device=self.device))

    def empty(self, shape, dtype=None):
        return _wrap(torch.empty(self._shape(shape), dtype=torch_dtype(dtype) or torch.float64, device=self.device))

    def full(self, shape, fill_value, dtype=None):
        dt = torch_dtype(dtype) or torch_dtype(_np.result_type(fill_value))
        return _wrap(torch.full(self._shape(shape), fill_value, dtype=dt, device=self.device))

    def zeros_like(self, a, dtype=None):
        return _wrap(torch.zeros_like(self._t(a), dtype=torch_dtype(dtype)))

    def ones_like(self, a, dtype=None):
        return _wrap(torch.ones_like(self._t(a), dtype=torch_dtype(dtype)))

    def empty_like(self, a, dtype=None):
        return _wrap(torch.empty_like(self._t(a), dtype=torch_dtype(dtype)))

    def eye(self, n, dtype=None):
        return _wrap(torch.eye(int(n), dtype=torch_dtype(dtype) or torch.float64, device=self.device))

    def asarray(self, a, dtype=None):
        return _wrap(self._t(a, dtype))

    def array(self, a, dtype=None, copy=True):
        t = self._t(a, dtype)
        return _wrap(t.clone() if copy and t is a else t)

    def ascontiguousarray(self, a, dtype=None):
        return _wrap(self._t(a, dtype).contiguous())

    def asnumpy(self, a):
        return a.get() if isinstance(a, DeviceArray) else _np.asarray(a)

    # ------------------------------------------------------------------ dtype algebra
    def result_type(self, *args):
        return _np.result_type(*[numpy_dtype(a.dtype) if isinstance(a, torch.Tensor) else
                                 (numpy_dtype(a) if isinstance(a, torch.dtype) else a) for a in args])

    def finfo(self, dt):
        return _np.finfo(numpy_dtype(dt.dtype if isinstance(dt, torch.Tensor) else dt))

    def iscomplexobj(self, a):
        return a.is_complex() if isinstance(a, torch.Tensor) else _np.iscomplexobj(a)

    def isrealobj(self, a):
        return not self.iscomplexobj(a)

    # ------------------------------------------------------------------ shape / layout
    def pad(self, array, pad_width, mode='constant', constant_values=0, **kw):
        t = self._t(array)
        pw = _np.broadcast_to(_np.asarray(pad_width, dtype=int), (t.dim(), 2))
        if mode != 'constant':
            raise NotImplementedError(f'NumpyFacade.pad: mode {mode!r}')
        flat = [int(v) for pair in reversed(pw.tolist()) for v in pair]     # torch pads from the last axis backwards
        return _wrap(torch.nn.functional.pad(t, flat, mode='constant', value=constant_values))

    def meshgrid(self, *xi, indexing='xy', **kw):
        return tuple(_wrap(g) for g in torch.meshgrid(*[self._t(x) for x in xi], indexing=indexing))

    def broadcast_to(self, a, shape):
        return _wrap(torch.broadcast_to(self._t(a), self._shape(shape)))

    def reshape(self, a, shape):
        return _wrap(self._t(a).reshape(self._shape(shape)))

    def squeeze(self, a, axis=None):
        t = self._t(a)
        return _wrap(t.squeeze() if axis is None else t.squeeze(axis))

    def stack(self, arrays, axis=0):
        return _wrap(torch.stack([self._t(a) for a in arrays], dim=axis))

    def column_stack(self, arrays):
        return _wrap(torch.column_stack([self._t(a) for a in arrays]))

    def concatenate(self, arrays, axis=0):
        return _wrap(torch.cat([self._t(a) for a in arrays], dim=axis))

    def roll(self, a, shift, axis=None):
        t = self._t(a)
        if axis is None:
            return _wrap(torch.roll(t.reshape(-1), shift).reshape(t.shape))
        return _wrap(torch.roll(t, shift, axis))

    def flipud(self, a):
        return _wrap(torch.flipud(self._t(a)))

    def fliplr(self, a):
        return _wrap(torch.fliplr(self._t(a)))

    # ------------------------------------------------------------------ arithmetic with numpy-only spellings
    def real(self, a):
        if _is_scalar(a):
            return _np.real(a)
        t = self._t(a)
        return _wrap(t.real if t.is_complex() else t)

    def imag(self, a):
        if _is_scalar(a):
            return _np.imag(a)
        t = self._t(a)
        return _wrap(t.imag if t.is_complex() else torch.zeros_like(t))

    def outer(self, a, b):
        return _wrap(torch.outer(self._t(a).reshape(-1), self._t(b).reshape(-1)))

    def where(self, cond, x=None, y=None):
        c = self._t(cond)
        if x is None and y is None:
            return tuple(_wrap(v) for v in torch.where(c))
        xs, ys = _is_scalar(x), _is_scalar(y)
        if xs and ys:
            dt = torch_dtype(_np.result_type(x, y))
            return _wrap(torch.where(c, torch.as_tensor(x, dtype=dt, device=self.device), torch.as_tensor(y, dtype=dt, device=self.device)))
        xt = None if xs else self._t(x)
        yt = None if ys else self._t(y)
        dt = torch.promote_types(xt.dtype if xt is not None else yt.dtype, yt.dtype if yt is not None else xt.dtype)
        if (xs and isinstance(x, complex)) or (ys and isinstance(y, complex)):
            dt = torch.promote_types(dt, torch.complex64)
        xt = torch.as_tensor(x, dtype=dt, device=self.device) if xs else xt.to(dt)
        yt = torch.as_tensor(y, dtype=dt, device=self.device) if ys else yt.to(dt)
        return _wrap(torch.where(c, xt, yt))

    def clip(self, a, a_min=None, a_max=None):
        return _wrap(torch.clamp(self._t(a), min=a_min, max=a_max))

    def diff(self, a, n=1, axis=-1):
        return _wrap(torch.diff(self._t(a), n=n, dim=axis))

    def allclose(self, a, b, rtol=1e-5, atol=1e-8):
        a, b = self._t(a), self._t(b)
        dt = torch.promote_types(a.dtype, b.dtype)
        return bool(torch.allclose(a.to(dt), b.to(dt), rtol=rtol, atol=atol))

    def isclose(self, a, b, rtol=1e-5, atol=1e-8):
        a, b = self._t(a), self._t(b)
        dt = torch.promote_types(a.dtype, b.dtype)
        return _wrap(torch.isclose(a.to(dt), b.to(dt), rtol=rtol, atol=atol))

    def _reduce(self, fn, a, axis, **kw):
        t = self._t(a)
        if axis is None:
            return _wrap(fn(t, **kw))
        return _wrap(fn(t, dim=axis, **kw))

    def sum(self, a, axis=None, dtype=None):
        t = self._t(a)
        if t.dtype == torch.bool:
            t = t.to(torch.int64)
        return self._reduce(torch.sum, t, axis, **({'dtype': torch_dtype(dtype)} if dtype is not None else {}))

    def mean(self, a, axis=None):
        return self._reduce(torch.mean, a, axis)

    def max(self, a, axis=None):
        return self._reduce(torch.amax, a, axis) if axis is not None else _wrap(self._t(a).max())

    def min(self, a, axis=None):
        return self._reduce(torch.amin, a, axis) if axis is not None else _wrap(self._t(a).min())

    amax, amin = max, min

    def argmin(self, a, axis=None):
        return self._reduce(torch.argmin, a, axis)

    def argmax(self, a, axis=None):
        return self._reduce(torch.argmax, a, axis)

    def any(self, a, axis=None):
        return self._reduce(torch.any, a, axis)

    def all(self, a, axis=None):
        return self._reduce(torch.all, a, axis)

    def tensordot(self, a, b, axes=2):
        return _wrap(torch.tensordot(self._t(a), self._t(b), dims=axes))

    def dot(self, a, b):
        return _wrap(torch.matmul(self._t(a), self._t(b)))

    matmul = dot
