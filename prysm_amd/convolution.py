"""Numerical convolution (prysm/convolution.py) -- SURVEY 8(f) ranks 1 and 4.

conv = fftshift(ifft2(fft2(ifftshift(o)) * fft2(ifftshift(h)))): one fused pm_fft2 for H, then the fused
3-pass fft2 -> x H -> ifft2 chain; the shifts are index rotations, the product with H happens in registers between
the two column transforms and the 1/(MN) rides on the last store.  Real objects / PSFs / actuator maps are read as
they are (PM_FLAG_REAL_INPUT), and a real object's result -- the reference keeps its real part -- comes out of a chain that runs on
half spectra end to end (PM_FLAG_REAL_OUTPUT, csrc/fft_c2r.h) for unpadded power-of-two sizes.  apply_transfer_functions is the same chain with the product of the given
transfer functions as H (the DM surface render of prysm/x/dm.py:254,330).
"""
import inspect

import torch

from . import _lib as L
from . import _ops
from .fttools import forward_ft_unit


def _match(o, h):
    """Bring two fields to one precision (numpy result type)."""
    co, ch = L.cdtype_of(o), L.cdtype_of(h)
    if co != ch:
        if torch.complex128 in (co, ch):
            o = o.to(torch.complex128 if o.is_complex() else torch.float64)
            h = h.to(torch.complex128 if h.is_complex() else torch.float64)
    return o, h


def conv(obj, psf):
    """Convolve an object and psf (arrays of the same shape) (convolution.py:9-31)."""
    o = L.as_field(obj)
    real = not o.is_complex()
    h = L.as_field(psf)
    o, h = _match(o, h)
    M, N = o.shape
    shift = (M // 2, N // 2)
    H = _ops.fft2(h, direction=-1, scale=1.0, in_shift=shift)
    # a real object keeps the real part: the chain then runs on half spectra end to end where the shapes allow (PM_FLAG_REAL_OUTPUT)
    return _ops.fft2_mul_ifft2(o, scale=1.0 / (M * N), mul=H, in_shift=shift, out_shift=shift, real_out=real)


def _ifftshift2(t):
    return torch.roll(t, shifts=(-(t.shape[0] // 2), -(t.shape[1] // 2)), dims=(0, 1))


def apply_transfer_functions(obj, dx, tfs, fx=None, fy=None, ft=None, fr=None, shift=False):
    """Blur an object by N transfer functions (convolution.py:34-113).

    tfs: arrays, or callables taking any of fx, fy, fr, ft (frequency grids on the device).  With shift=False the
    transfer functions have their origin at sample [0, 0]; with shift=True at the centre of the array.
    """
    o = L.as_field(obj)
    real = not o.is_complex()
    M, N = o.shape
    if any(callable(tf) for tf in tfs):
        if fx is None or fy is None:
            uy, ux = [L.as_device(forward_ft_unit(dx, n, shift=shift)) for n in o.shape]
            fx = ux if fx is None else fx
            fy = uy if fy is None else fy
        fx, fy = L.as_device(fx), L.as_device(fy)
        if fx.dim() == 1 or (fx.dim() == 2 and fx.shape[0] != 1 and fy.dim() == 1):   # optimize_xy_separable
            fx, fy = fx.reshape(1, -1), fy.reshape(-1, 1)
        computed_fr, computed_ft = torch.hypot(fx, fy), torch.atan2(fy + 0 * fx, fx + 0 * fy)   # cart_to_polar
        fr = computed_fr if fr is None else L.as_device(fr)
        ft = computed_ft if ft is None else L.as_device(ft)
    cd = L.cdtype_of(o)
    H = None
    for tf in tfs:
        if callable(tf):
            params = inspect.signature(tf).parameters
            kwargs = {k: v for k, v in (('fx', fx), ('fy', fy), ('fr', fr), ('ft', ft)) if k in params}
            if not kwargs:
                raise ValueError(f'{tf} accepts none of fx, fy, fr, ft; a transfer function must accept at least one')
            tf = tf(**kwargs)
        tf = L.as_device(tf)
        if tf.dtype == torch.complex128 or tf.dtype == torch.float64:
            if cd == torch.complex64:
                cd = torch.complex128
        tf = torch.broadcast_to(tf, (M, N)) if tuple(tf.shape) != (M, N) else tf
        H = tf if H is None else H * tf
    if H is None:
        H = torch.ones((M, N), dtype=cd, device=o.device)
    H = H.to(cd)
    if cd == torch.complex128 and L.cdtype_of(o) != cd:
        o = o.to(torch.complex128 if o.is_complex() else torch.float64)
    if shift:
        H = _ifftshift2(H)      # centred transfer functions -> origin at [0, 0]
    sh = (M // 2, N // 2)
    return _ops.fft2_mul_ifft2(o, scale=1.0 / (M * N), mul=H.contiguous(), in_shift=sh, out_shift=sh, real_out=real)
