"""Angular-spectrum (plane-to-plane) free space propagation (prysm/propagation/angular_spectrum.py).

ifft2(fft2(pad(field)) * H) with the separable Fresnel transfer function
    H[i, j] = exp(-i pi (wvl/1e3) z ky[i]^2) * exp(-i pi (wvl/1e3) z kx[j]^2)
never materialised: two length-N vectors are synthesised on the device.  Power-of-two sizes run THREE
passes (row FFT; column FFT x H x column IFFT in registers; row IFFT -- 6 N^2 s bytes instead of the 8 N^2 s
of two full transforms); other sizes compose two fused transforms.  The 1/(MN) of ifft2 rides on the last store.
"""
import math

import torch

from .. import _lib as L
from .. import _ops
from ..conf import config
from ._kernels import _padded_shape, _shape_before_pad


def _cdtype():
    return L.torch_dtype(config.precision_complex)


def angular_spectrum_transfer_function(samples, wvl, dx, z):
    """Precompute the transfer function of free space (angular_spectrum.py:82-114).

    Returns the materialised (rows, cols) array in config.precision_complex, for API parity;
    angular_spectrum() itself uses the two factors directly.
    """
    if isinstance(samples, int):
        samples = (samples, samples)
    hy, hx = _ops.as_tf_vectors(tuple(samples), wvl, dx, z, _cdtype(), cache=False)
    return _ops.outer(hy, hx)


def _field(field, other_dtype):
    """numpy result type of field * transfer_function."""
    f = L.as_field(field)
    if other_dtype == torch.complex128 and L.cdtype_of(f) == torch.complex64:
        f = f.to(torch.complex128 if f.is_complex() else torch.float64)
    return f


def _tf_vectors(shape, wvl, dx, z, dtype, batch):
    """(hy, hx) of the separable transfer function; one pair per field when wvl / z are sequences (stack input)."""
    multi = [v for v in (wvl, z) if hasattr(v, '__len__')]
    if not multi:
        return _ops.as_tf_vectors(shape, wvl, dx, z, dtype)
    if batch is None or any(len(v) != batch for v in multi):
        raise ValueError('per-field wvl / z need a (batch, rows, cols) stack with one entry per field')
    wv = list(wvl) if hasattr(wvl, '__len__') else [wvl] * batch
    zz = list(z) if hasattr(z, '__len__') else [z] * batch
    pairs = [_ops.as_tf_vectors(shape, float(w), dx, float(zi), dtype) for w, zi in zip(wv, zz)]
    return torch.stack([p[0] for p in pairs]), torch.stack([p[1] for p in pairs])


def angular_spectrum(field, wvl, dx, z, Q=2, tf=None):
    """Propagate a field via the angular spectrum method (angular_spectrum.py:9-42).

    Extension: `field` may be a (batch, rows, cols) stack, with `wvl` / `z` scalars or one value per field."""
    if tf is not None:
        tf = L.as_complex(tf)
        f = _field(field, tf.dtype)
        if tf.dtype != L.cdtype_of(f):
            tf = tf.to(L.cdtype_of(f))
        M, N = f.shape[-2:]
        return _ops.fft2_mul_ifft2(f, scale=1.0 / (M * N), mul=tf.contiguous())
    f = _field(field, _cdtype())
    m, n = f.shape[-2:]
    M, N = _padded_shape((m, n), Q)
    in_off = (math.ceil((M - m) / 2), math.ceil((N - n) / 2))
    hy, hx = _tf_vectors((M, N), wvl, dx, z, L.cdtype_of(f), f.shape[0] if f.dim() == 3 else None)
    return _ops.fft2_mul_ifft2(f, scale=1.0 / (M * N), mul=hy, mul_x=hx, shape=(M, N), in_off=in_off)


def angular_spectrum_adjoint(field, wvl, dx, z, Q=2, tf=None):
    """Apply the adjoint of angular_spectrum (angular_spectrum.py:45-79): conj(tf), then crop."""
    if tf is not None:
        tf = L.as_complex(tf)
        f = _field(field, tf.dtype)
        if tf.dtype != L.cdtype_of(f):
            tf = tf.to(L.cdtype_of(f))
        M, N = f.shape[-2:]
        return _ops.fft2_mul_ifft2(f, scale=1.0 / (M * N), mul=tf.contiguous(), mul_conj=True)
    f = _field(field, _cdtype())
    M, N = f.shape[-2:]
    out_shape = _shape_before_pad((M, N), Q)
    hy, hx = _tf_vectors((M, N), wvl, dx, z, L.cdtype_of(f), f.shape[0] if f.dim() == 3 else None)
    if out_shape == (M, N):
        return _ops.fft2_mul_ifft2(f, scale=1.0 / (M * N), mul=hy, mul_x=hx, mul_conj=True)
    out_off = (math.ceil((M - out_shape[0]) / 2), math.ceil((N - out_shape[1]) / 2))
    return _ops.fft2_mul_ifft2(f, scale=1.0 / (M * N), mul=hy, mul_x=hx, mul_conj=True, out_shape=out_shape,
                               out_off=out_off)


def fresnel_number(a, L_, lambda_):
    """Compute the Fresnel number (angular_spectrum.py:117-137)."""
    return a**2 / (L_ * lambda_)


def talbot_distance(a, lambda_):
    """Compute the talbot distance (angular_spectrum.py:140-160)."""
    num = lambda_
    den = 1 - math.sqrt(1 - lambda_**2 / a**2)
    return num / den
