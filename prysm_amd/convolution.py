"""Numerical convolution (prysm/convolution.py:9-31) -- SURVEY 8(f) rank 1.

conv = fftshift(ifft2(fft2(ifftshift(o)) * fft2(ifftshift(h)))): three fused pm_fft2 calls; the shifts are
index rotations, the product with H rides on the store of the object's forward transform and the 1/(MN) on
the store of the inverse.
"""
from . import _lib as L
from . import _ops


def conv(obj, psf):
    """Convolve an object and psf (arrays of the same shape)."""
    o = L.as_device(obj)
    real = not o.is_complex()
    o = L.as_complex(o)
    h = L.as_complex(psf)
    if h.dtype != o.dtype:
        big = o.dtype if o.element_size() > h.element_size() else h.dtype
        o, h = o.to(big), h.to(big)
    M, N = o.shape
    shift = (M // 2, N // 2)
    H = _ops.fft2(h, direction=-1, scale=1.0, in_shift=shift)
    i = _ops.fft2_mul_ifft2(o, scale=1.0 / (M * N), mul=H, in_shift=shift, out_shift=shift)
    return i.real if real else i
