"""Fourier-transform toolbox on the MI355X: the counterpart of prysm/fttools.py.

Same names, arguments and conventions as the reference (everything is
"FFT-centred": the origin sits at index n//2); arrays are torch tensors in HBM.
"""
import math

import numpy as truenp
import torch

from . import _lib as L
from . import _ops
from .conf import config


def _rdtype():
    return L.torch_dtype(config.compute_precision)


def _cdtype():
    return L.torch_dtype(config.precision_complex)


def fftrange(n, dtype=None):
    """FFT-aligned coordinate grid for n samples (prysm/fttools.py:13-15)."""
    dt = torch.int64 if dtype is None else L.torch_dtype(dtype)
    return torch.arange(-(n // 2), -(n // 2) + n, dtype=dt, device=L.device())


def _next_power_of_2(n):
    return 1 << math.ceil(math.log2(n))


def next_fast_len(n):
    """The next fast FFT size (prysm/fttools.py:23-31: scipy's next_fast_len, a power of two as the fallback).

    Fast lengths here are powers of two up to 32768, and from 96 the lengths 3 / 5 / 7 x 2^k with 2^k <= 8192 (the mixed-radix kernel
    up to 8192, one radix-3 / 5 / 7 step around engine transforms above: csrc/fft_mixed.h, csrc/bigfft.hip), e.g. 2560 for 2559
    where a power of two gives 4096.

    Trade-off (these are not scipy's values, which also admit 2^a 3^b 5^c ... lengths): composite lengths up to 8192 do run on their
    own factors (the mixed-radix kernel, csrc/fft_mixed.h: 3000^2 complex64 in 92 us), but at a little over half the memory-bound rate
    of the engine (2000^2 39 us against 32 us for 2048^2), so the lengths returned here remain the safe choice.  A non power of
    two also leaves the paths that need engine lengths -- the fused fft2 -> multiply -> ifft2 chain, pupil synthesis inside the row
    load, the Hermitian real-input path, grouped wavelengths -- for the composed routes.  Callers that want those paths pad to a power
    of two (``1 << ceil(log2(n))``).
    """
    n = int(n)
    best = _next_power_of_2(n)
    for R in (3, 5, 7):
        q = max(16, _next_power_of_2(-(-n // R)))
        if q <= 8192 and R * q >= 96 and R * q <= 32768:
            best = min(best, R * q)
    return best


def _czt_len(n):
    """Convolution length of one chirp-Z axis: a power of two while the whole axis fits one fused kernel (pm_czt_axis, up to 8192
    points), else the next fast length (16384-point convolutions take the radix-R path of pm_fft1: 10240 for 8703)."""
    p2 = _next_power_of_2(n)
    return p2 if p2 <= 8192 else next_fast_len(n)


def fftfreq(n, d=1.0):
    """FFT frequency vector in config.precision (prysm/fttools.py:34-40)."""
    out = truenp.fft.fftfreq(n, d).astype(config.compute_precision)
    return L.as_device(out)


def pad2d(array, Q=2, value=0, mode='constant', out_shape=None):
    """Symmetrically pad a 2-D array (prysm/fttools.py:43-100); data lands at offset ceil(d/2)."""
    if Q == 1 and out_shape is None:
        return array
    array = L.as_device(array)
    in_shape = tuple(array.shape)
    if out_shape is None:
        out_shape = [math.ceil(s * Q) for s in in_shape]
    elif isinstance(out_shape, int):
        out_shape = [out_shape] * array.dim()
    shape_diff = [o - i for o, i in zip(out_shape, in_shape)]
    off = [math.ceil(d / 2) for d in shape_diff]      # np.pad widths (d - d // 2, d // 2): the same offset
    if mode == 'constant':
        return _ops.embed(array, tuple(out_shape), off, fill=value)
    if mode in _ops._PAD_MODES and array.dim() == 2 and all(d >= 0 for d in shape_diff):
        return _ops.pad_index(array, tuple(out_shape), off, mode)
    if mode == 'empty':
        return _ops.embed(array, tuple(out_shape), off, fill=0)     # np.pad leaves the border undefined; zeros are a valid instance
    if mode in _STAT_PADS and array.dim() == 2 and all(d >= 0 for d in shape_diff):
        return _pad_stat(array, [(d - d // 2, d // 2) for d in shape_diff], mode)
    raise NotImplementedError(f"pad2d: np.pad mode {mode!r} is not implemented (constant, edge, reflect, symmetric, wrap, empty, mean, "
                              "maximum, minimum, median and linear_ramp are)")


_STAT_PADS = ('mean', 'maximum', 'minimum', 'median', 'linear_ramp')


def _pad_stat(a, widths, mode):
    """np.pad's statistical modes and linear_ramp (defaults: statistics over the whole axis, end value 0), axis by axis like numpy
    -- the second axis sees the rows the first one added.  Off the hot path: tensor reductions and concatenations on the device."""
    if mode in ('maximum', 'minimum', 'median') and a.is_complex():
        raise TypeError(f'pad2d: mode {mode!r} needs a real array')
    for axis, (before, after) in enumerate(widths):
        if before == 0 and after == 0:
            continue
        if mode == 'linear_ramp':
            first, last = a.narrow(axis, 0, 1), a.narrow(axis, a.shape[axis] - 1, 1)
            work = a if a.is_floating_point() or a.is_complex() else a.to(torch.float64)
            first, last = first.to(work.dtype), last.to(work.dtype)
            shp = [1, 1]
            shp[axis] = -1
            kb = (torch.arange(before, device=a.device, dtype=torch.float64) / max(before, 1)).reshape(shp)
            ka = ((torch.arange(after, device=a.device, dtype=torch.float64) + 1) / max(after, 1)).reshape(shp)
            rd = work.real.dtype if work.is_complex() else work.dtype
            pre = first * kb.to(rd)                       # end value 0 -> edge, outermost sample first
            post = last * (1 - ka).to(rd)                 # edge -> end value 0
            out = torch.cat((pre, work, post), dim=axis)
            a = out if work is a else torch.floor(out).to(a.dtype)     # integer arrays: numpy builds the ramp with linspace(dtype=int), which FLOORS
            continue
        if mode == 'mean':
            if a.is_floating_point() or a.is_complex():
                stat = a.mean(dim=axis, keepdim=True)
            else:       # integer arrays: np.pad rounds the statistic (half to even) before casting back
                stat = torch.round(a.to(torch.float64).mean(dim=axis, keepdim=True)).to(a.dtype)
        elif mode == 'maximum':
            stat = a.amax(dim=axis, keepdim=True)
        elif mode == 'minimum':
            stat = a.amin(dim=axis, keepdim=True)
        else:
            # median = mean of the two middle order statistics (torch.quantile refuses arrays beyond ~16M elements: 4096^2)
            srt = torch.sort(a.to(torch.float64), dim=axis).values
            n = a.shape[axis]
            mid = (srt.narrow(axis, (n - 1) // 2, 1) + srt.narrow(axis, n // 2, 1)) * 0.5
            stat = (mid if a.is_floating_point() else torch.round(mid)).to(a.dtype)
        reps = [1, 1]
        reps[axis] = before
        pre = stat.repeat(reps)
        reps[axis] = after
        a = torch.cat((pre, a, stat.repeat(reps)), dim=axis)
    return a


def crop_center(img, out_shape):
    """Crop the central out_shape window, FFT aligned (prysm/fttools.py:103-125).

    Returns a view, like the reference.
    """
    if isinstance(out_shape, int):
        out_shape = (out_shape, out_shape)
    padding = [i - o for i, o in zip(img.shape, out_shape)]
    left = [math.ceil(p / 2) for p in padding]
    slcs = tuple(slice(l, l + o) for l, o in zip(left, out_shape))  # NOQA
    return img[slcs]


def forward_ft_unit(dx, samples, shift=True):
    """Frequency axis of an FFT (prysm/fttools.py:128-152)."""
    unit = fftfreq(samples, dx)
    if shift:
        return torch.roll(unit, samples // 2)
    return unit


def _as_real_vec(v):
    t = L.as_device(v)
    if t.is_complex():
        raise TypeError('coordinate vectors must be real')
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(_rdtype())
    return t


def _promote_input(ary, cdtype):
    """numpy-style result type of (input array) x (operator of dtype cdtype)."""
    t = L.as_complex(ary)
    if cdtype == torch.complex128 and t.dtype == torch.complex64:
        t = t.to(torch.complex128)
    return t


class MDFT:
    """Matrix DFT: out = norm * Ey @ ary @ Ex.T on the MFMA matrix cores.

    Mirror of prysm.fttools.MDFT (prysm/fttools.py:155-232): same constructor, ``__call__``,
    ``adjoint``, ``nbytes`` and the same attributes (``Ex``, ``Ey``, ``norm``,
    ``_forward_left_first``, ``_adjoint_left_first``).  The bases are generated on the device
    (phase reduced in fp64, one rounding) and held by the instance -- holding the instance is the
    caching mechanism, as in the reference.
    """

    def __init__(self, x, y, fx, fy, sign=-1, norm=1.0):
        x, y, fx, fy = (_as_real_vec(v) for v in (x, y, fx, fy))
        rd = torch.float64 if torch.float64 in (x.dtype, fx.dtype, y.dtype, fy.dtype) else torch.float32
        x, y, fx, fy = (v.to(rd) for v in (x, y, fx, fy))
        cd = L._COMPLEX_OF[rd]
        self.Ex = _ops.mdft_basis(fx, x, sign, cd)   # (len(fx), len(x))
        self.Ey = _ops.mdft_basis(fy, y, sign, cd)   # (len(fy), len(y))
        self.norm = norm
        Nx, Ny, Mx, My = x.numel(), y.numel(), fx.numel(), fy.numel()
        self._forward_left_first = My * Nx * (Ny + Mx) <= Ny * Mx * (Nx + My)
        self._adjoint_left_first = Ny * Mx * (My + Nx) <= My * Nx * (Mx + Ny)

    @classmethod
    def _for_focus_grids(cls, pupil_samples, focal_samples, pupil_dx, focal_dx, focal_shift, inv_lz, rdtype, sign, norm):
        """The operator prepare_executor builds -- MDFT(*coordinates_for_focus(...)) -- without materialising the four coordinate
        vectors: the bases are generated from the grid parameters (pm_mdft_basis_grid), bit for bit the same matrices."""
        pny, pnx = pupil_samples
        fny, fnx = focal_samples
        fsx, fsy = focal_shift
        cd = L._COMPLEX_OF[rdtype]
        self = cls.__new__(cls)
        self.Ex = _ops.mdft_basis_grid(fnx, pnx, focal_dx, fsx, inv_lz, pupil_dx, sign, cd)
        self.Ey = _ops.mdft_basis_grid(fny, pny, focal_dx, fsy, inv_lz, pupil_dx, sign, cd)
        self.norm = norm
        self._forward_left_first = fny * pnx * (pny + fnx) <= pny * fnx * (pnx + fny)
        self._adjoint_left_first = pny * fnx * (fny + pnx) <= fny * pnx * (fnx + pny)
        return self

    def _cast(self, ary):
        a = _promote_input(ary, self.Ex.dtype)
        if a.dtype != self.Ex.dtype:   # complex128 data through float32 bases: numpy promotes the bases
            return a, self.Ex.to(a.dtype), self.Ey.to(a.dtype)
        return a, self.Ex, self.Ey

    def __call__(self, ary):
        """Apply the forward DFT to ary (prysm/fttools.py:201-207)."""
        a, Ex, Ey = self._cast(ary)
        if not self._forward_left_first:
            out = _ops.cgemm(a, Ex, 0, 2)                      # ary @ Ex.T
            return _ops.cgemm(Ey, out, 0, 0, alpha=self.norm)  # Ey @ .
        out = _ops.cgemm(Ey, a, 0, 0)
        return _ops.cgemm(out, Ex, 0, 2, alpha=self.norm)

    def intensity(self, ary, out=None, weight=1.0):
        """weight * |self(ary)|^2 as a real image, added to `out` when given: focus_dft + Wavefront.intensity
        (prysm/propagation/wavefront.py:146-151) + the weighted accumulate of the polychromatic recipe, with the modulus in the
        epilogue of the second product (pm_cgemm_abs2) -- the complex focal field is not written.  Falls back to the composed form
        for shapes / precisions that kernel does not take."""
        a, Ex, Ey = self._cast(ary)
        # the first product once; if the fused epilogue does not take this shape / precision (None), the second one is finished as a
        # plain product + |.|^2 -- not by starting over with self(ary), which would run three products instead of two
        if not self._forward_left_first:
            t = _ops.cgemm(a, Ex, 0, 2)
            res = _ops.cgemm_abs2(Ey, t, 0, 0, alpha=self.norm, out=out, weight=weight)
            E = _ops.cgemm(Ey, t, 0, 0, alpha=self.norm) if res is None else None
        else:
            t = _ops.cgemm(Ey, a, 0, 0)
            res = _ops.cgemm_abs2(t, Ex, 0, 2, alpha=self.norm, out=out, weight=weight)
            E = _ops.cgemm(t, Ex, 0, 2, alpha=self.norm) if res is None else None
        if res is not None:
            return res
        if out is None:
            I = _ops.abs2(E)
            return I if weight == 1.0 else I * weight
        return _ops.abs2(E, out=out, weight=weight)

    def adjoint(self, grad):
        """Apply the conjugate transpose (prysm/fttools.py:209-228): norm * Ey^H @ grad @ conj(Ex)."""
        g, Ex, Ey = self._cast(grad)
        if not self._adjoint_left_first:
            out = _ops.cgemm(g, Ex, 0, 1)                      # grad @ conj(Ex)
            return _ops.cgemm(Ey, out, 3, 0, alpha=self.norm)  # Ey^H @ .
        out = _ops.cgemm(Ey, g, 3, 0)
        return _ops.cgemm(out, Ex, 0, 1, alpha=self.norm)

    def nbytes(self):
        """Total size in memory of the basis matrices, bytes."""
        return self.Ex.numel() * self.Ex.element_size() + self.Ey.numel() * self.Ey.element_size()


def _cexp_turns(turns):
    """exp(2 pi i t) for a real float64 vector t, in config.precision_complex (device kernel)."""
    t = turns.to(torch.float64)
    one = torch.ones(1, dtype=torch.float64, device=t.device)
    E = _ops.mdft_basis(t, one, +1, torch.complex128)[:, 0].contiguous()
    return E.to(_cdtype())


class CZT:
    """Chirp-Z transform with the same external API as MDFT (prysm/fttools.py:235-361).

    Bluestein factorisation per axis: chirp multiply, zero-padded FFT of length K, multiply by the
    transformed chirp, inverse FFT, slice, chirp multiply.  The FFTs are the library's batched 1-D
    passes; zero padding and slicing are folded into their load / store windows.
    """

    def __init__(self, x, y, fx, fy, sign=-1, norm=1.0):
        if sign not in (-1, 1):
            raise ValueError(f'sign must be -1 or +1, got {sign}')
        self.sign = sign
        self.norm = norm
        x, y, fx, fy = (_as_real_vec(v) for v in (x, y, fx, fy))
        Nx, Mx, Ny, My = x.numel(), fx.numel(), y.numel(), fy.numel()
        xs, ys, fxs, fys = (v.to(torch.float64) for v in (x, y, fx, fy))
        dx = float(xs[1] - xs[0])
        dfx = float(fxs[1] - fxs[0])
        dy = float(ys[1] - ys[0])
        dfy = float(fys[1] - fys[0])
        alpha_x, alpha_y = dx * dfx, dy * dfy
        shift_x = float(fxs[Mx // 2]) / dfx
        shift_y = float(fys[My // 2]) / dfy
        Kx = _czt_len(Nx + Mx - 1)
        Ky = _czt_len(Ny + My - 1)
        Hx, bx, ax = _prepare_czt_basis(Nx, Mx, Kx, shift_x, alpha_x, sign)
        Hy, by, ay = _prepare_czt_basis(Ny, My, Ky, shift_y, alpha_y, sign)
        self._brow, self._Hrow, self._arow = by, Hy, ay        # vectors along axis 0
        self._bcol, self._Hcol, self._acol = bx, Hx, ax        # vectors along axis 1
        self._x_phase = _cexp_turns(sign * float(xs[Nx // 2]) * fxs)
        self._y_phase = _cexp_turns(sign * float(ys[Ny // 2]) * fys)
        # post-slice factors a * phase, applied in one sweep
        self._post_col = (self._acol * self._x_phase).contiguous()
        self._post_row = (self._arow * self._y_phase).contiguous()
        self._finish(Nx, Ny, Mx, My, Kx, Ky)

    @classmethod
    def _for_focus_grids(cls, pupil_samples, focal_samples, pupil_dx, focal_dx, focal_shift, inv_lz, rdtype, sign, norm):
        """The operator prepare_executor builds -- CZT(*coordinates_for_focus(...)) -- from the grid parameters: the handful of
        coordinate samples the constructor reads (x[0], x[1], f[0], f[1], f[M/2]) are formed on the host in the precision the
        vectors would have, and the chirps of each axis come from one kernel (pm_czt_vectors) + one transform, where the constructor
        issues a dozen small array operations and five device-to-host reads per axis (750 -> ~100 us per executor; the polychromatic
        recipe builds one per wavelength).  Pupil grids are FFT-centred, so x[N/2] = 0 and the centre phase ramp is 1."""
        if sign not in (-1, 1):
            raise ValueError(f'sign must be -1 or +1, got {sign}')
        dt = truenp.float32 if rdtype == torch.float32 else truenp.float64
        cd = L._COMPLEX_OF[rdtype]
        self = cls.__new__(cls)
        self.sign, self.norm = sign, norm

        def axis(N, M, fshift):
            xv = lambda j: float(dt(j - N // 2) * dt(pupil_dx))                                     # noqa: E731
            fv = lambda i: float((dt(i - M // 2) * dt(focal_dx) + dt(fshift)) * dt(inv_lz))          # noqa: E731
            dxs = xv(1) - xv(0)
            dfs = fv(1) - fv(0)
            K = _czt_len(N + M - 1)
            b, a, h = _ops.czt_vectors(N, M, K, fv(M // 2) / dfs, sign * dxs * dfs / 2.0, cd)
            return _ops.fft1(h, K, axis=-1), b, a, K

        (pny, pnx), (fny, fnx) = pupil_samples, focal_samples
        fsx, fsy = focal_shift
        Hx, bx, ax, Kx = axis(pnx, fnx, fsx)
        Hy, by, ay, Ky = axis(pny, fny, fsy)
        self._brow, self._Hrow, self._arow = by, Hy, ay
        self._bcol, self._Hcol, self._acol = bx, Hx, ax
        self._x_phase = self._y_phase = None
        self._post_col, self._post_row = ax, ay
        self._finish(pnx, pny, fnx, fny, Kx, Ky)
        return self

    def _finish(self, Nx, Ny, Mx, My, Kx, Ky):
        self._Nx, self._Ny, self._Mx, self._My = Nx, Ny, Mx, My
        self._Kx, self._Ky = Kx, Ky
        self._sx, self._sy = Nx - 1, Ny - 1
        # which axis first: the reference minimises flops (fttools.py:283-289); here a pass is also slower when it has too few
        # sequences to fill the chip -- a row pass runs one or two rows per workgroup, a column pass a tile of 64 B of columns --
        # e.g. 2048^2 -> 512^2: x first leaves the y pass 512 columns = 64 workgroups on 256 CUs (73 us), y first 37 us
        def _cost(nrow_seq, krow, ncol_seq, kcol):
            tc = 8 if _cdtype() == torch.complex64 else 4
            row_wgs = nrow_seq / (2 if krow >= 4096 and _cdtype() == torch.complex64 else max(1, 4096 // max(krow, 16) // 16))
            col_wgs = ncol_seq / tc
            return (nrow_seq * krow * math.log2(krow) / min(1.0, max(row_wgs, 1) / 256.0) +
                    ncol_seq * kcol * math.log2(kcol) / min(1.0, max(col_wgs, 1) / 256.0))
        x_first_cost = _cost(Ny, Kx, Mx, Ky)      # rows of the (Ny x Nx) input first, then the columns of the (Ny x Mx) result
        y_first_cost = _cost(My, Kx, Nx, Ky)      # columns of the input first, then the rows of the (My x Nx) result
        self._x_first = x_first_cost <= y_first_cost

    # Engine lengths: one kernel per axis (pm_czt_axis: chirp multiply, K-point transform, x H, inverse, slice, chirp multiply -- the
    # K-point sequence never leaves the registers of its workgroup); otherwise the composition of pm_fft1 and pm_scale_sep.
    def _fused(self, K):
        return _ops._is_pow2_engine(K) and K >= 16

    def _xaxis(self, out, scale=1.0, pre=None):
        if self._fused(self._Kx):
            return _ops.czt_axis(out, self._Kx, 1, self._Hcol, pre=pre, post=self._post_col, out_len=self._Mx, out_off=self._sx,
                                 scale=scale)
        if pre is not None:
            out = _ops.scale_sep(out, col_vec=pre)
        out = _ops.fft1(out, self._Kx, axis=1)
        out = _ops.scale_sep(out, col_vec=self._Hcol)
        out = _ops.fft1(out, axis=1, direction=+1, scale=1.0 / self._Kx, out_len=self._Mx, out_off=self._sx)
        return _ops.scale_sep(out, col_vec=self._post_col, scale=scale)

    def _yaxis(self, out, scale=1.0, pre=None):
        if self._fused(self._Ky):
            return _ops.czt_axis(out, self._Ky, 0, self._Hrow, pre=pre, post=self._post_row, out_len=self._My, out_off=self._sy,
                                 scale=scale)
        if pre is not None:
            out = _ops.scale_sep(out, row_vec=pre)
        out = _ops.fft1(out, self._Ky, axis=0)
        out = _ops.scale_sep(out, row_vec=self._Hrow)
        out = _ops.fft1(out, axis=0, direction=+1, scale=1.0 / self._Ky, out_len=self._My, out_off=self._sy)
        return _ops.scale_sep(out, row_vec=self._post_row, scale=scale)

    def __call__(self, ary):
        a = _promote_input(ary, _cdtype()).to(_cdtype())
        if a.stride(-1) != 1:
            a = a.contiguous()
        # the input chirps b ride on the load of each axis' kernel (they commute with the other axis' transform)
        if self._x_first:
            return self._yaxis(self._xaxis(a, pre=self._bcol), scale=self.norm, pre=self._brow)
        return self._xaxis(self._yaxis(a, pre=self._brow), scale=self.norm, pre=self._bcol)

    def _xadj(self, out, pre=None, post=None, scale=1.0):
        # zero-embed at [sx, sx+Mx) of length Kx, fft, * conj(H), ifft, keep [:Nx]
        if self._fused(self._Kx):
            return _ops.czt_axis(out, self._Kx, 1, self._Hcol, pre=pre, post=post, out_len=self._Nx, in_off=self._sx, conj=True,
                                 scale=scale)
        if pre is not None:
            out = _ops.scale_sep(out, col_vec=pre, col_conj=True)
        out = self._xadj_composed(out)
        return _ops.scale_sep(out, col_vec=post, col_conj=True, scale=scale) if (post is not None or scale != 1.0) else out

    def _xadj_composed(self, out):
        out = _ops.fft1(out, self._Kx, axis=1, in_off=self._sx)
        out = _ops.scale_sep(out, col_vec=self._Hcol, col_conj=True)
        return _ops.fft1(out, axis=1, direction=+1, scale=1.0 / self._Kx, out_len=self._Nx)

    def _yadj(self, out, pre=None, post=None, scale=1.0):
        if self._fused(self._Ky):
            return _ops.czt_axis(out, self._Ky, 0, self._Hrow, pre=pre, post=post, out_len=self._Ny, in_off=self._sy, conj=True,
                                 scale=scale)
        if pre is not None:
            out = _ops.scale_sep(out, row_vec=pre, row_conj=True)
        out = self._yadj_composed(out)
        return _ops.scale_sep(out, row_vec=post, row_conj=True, scale=scale) if (post is not None or scale != 1.0) else out

    def _yadj_composed(self, out):
        out = _ops.fft1(out, self._Ky, axis=0, in_off=self._sy)
        out = _ops.scale_sep(out, row_vec=self._Hrow, row_conj=True)
        return _ops.fft1(out, axis=0, direction=+1, scale=1.0 / self._Ky, out_len=self._Ny)

    def adjoint(self, grad):
        g = _promote_input(grad, _cdtype()).to(_cdtype())
        if g.stride(-1) != 1:
            g = g.contiguous()
        # conj(a * phase) on the way in, conj(b) on the way out, per axis (each commutes with the other axis' transform)
        if self._x_first:
            out = self._yadj(g, pre=self._post_row, post=self._brow)
            return self._xadj(out, pre=self._post_col, post=self._bcol, scale=self.norm)
        out = self._xadj(g, pre=self._post_col, post=self._bcol)
        return self._yadj(out, pre=self._post_row, post=self._brow, scale=self.norm)

    def nbytes(self):
        total = 0
        for arr in (self._brow, self._bcol, self._Hrow, self._Hcol, self._arow, self._acol):
            total += arr.numel() * arr.element_size()
        total += (self._Mx + self._My) * self._arow.element_size()      # the two centre phase ramps (all ones for a centred pupil grid)
        return total


def _prepare_czt_basis(N, M, K, shift, alpha, sign=-1):
    """prysm/fttools.py:364-389; phases formed in fp64 on the device, FFT of the chirp by pm_fft1."""
    n = fftrange(N, dtype=torch.float64)
    m = fftrange(M, dtype=torch.float64)
    q = m + shift
    half = sign * alpha / 2.0           # exp(sign i pi alpha q^2) = exp(2 pi i * half * q^2)
    a = _cexp_turns(half * q * q)
    b = _cexp_turns(half * n * n)
    d_min = float(m[0] - n[-1])
    d_max = float(m[-1] - n[0])
    d = torch.arange(d_min, d_max + 1, dtype=torch.float64, device=L.device())
    h = torch.zeros(K, dtype=_cdtype(), device=L.device())
    h[:d.numel()] = _cexp_turns(-half * (d + shift) * (d + shift))
    H = _ops.fft1(h, K, axis=-1)
    return H, b, a


class FFTDFT:
    """DFT accelerated by FFTs for compatible uniform grids (prysm/fttools.py:392-535).

    Requires |dx * dfx| = 1/K with integer K >= max(N, M) per axis; one zero-padded FFT per axis,
    the axis producing the smaller intermediate first.
    """

    def __init__(self, x, y, fx, fy, sign=-1, norm=1.0):
        if sign not in (-1, 1):
            raise ValueError(f'sign must be -1 or +1, got {sign}')
        x, y, fx, fy = (_as_real_vec(v) for v in (x, y, fx, fy))
        Nx, Ny, Mx, My = x.numel(), y.numel(), fx.numel(), fy.numel()
        dx = _uniform_spacing(x, 'x')
        dy = _uniform_spacing(y, 'y')
        dfx = _uniform_spacing(fx, 'fx')
        dfy = _uniform_spacing(fy, 'fy')
        Kx = _fft_compatible_length(dx * dfx, Nx, Mx, 'x/fx')
        Ky = _fft_compatible_length(dy * dfy, Ny, My, 'y/fy')
        xs, ys, fxs, fys = (v.to(torch.float64) for v in (x, y, fx, fy))
        nx = torch.arange(Nx, dtype=torch.float64, device=L.device())
        ny = torch.arange(Ny, dtype=torch.float64, device=L.device())
        self._pre_x = _cexp_turns(sign * nx * dx * float(fxs[0]))
        self._pre_y = _cexp_turns(sign * ny * dy * float(fys[0]))
        self._post_x = _cexp_turns(sign * float(xs[0]) * fxs)
        self._post_y = _cexp_turns(sign * float(ys[0]) * fys)
        self._Nx, self._Ny, self._Mx, self._My = Nx, Ny, Mx, My
        self._Kx, self._Ky = Kx, Ky
        self._x_direction = sign if dx * dfx > 0 else -sign
        self._y_direction = sign if dy * dfy > 0 else -sign
        self.norm = norm
        x_first_cost = Ny * Kx * math.log2(Kx) + Mx * Ky * math.log2(Ky)
        y_first_cost = Nx * Ky * math.log2(Ky) + My * Kx * math.log2(Kx)
        self._x_first = x_first_cost <= y_first_cost

    # Engine lengths K (powers of two up to 8192): ONE kernel per axis with the phase ramps in the transform's load and store
    # (pm_fft1_ramp: ramp multiply, zero pad, K-point transform, crop, ramp multiply); otherwise the composition of pm_scale_sep and
    # pm_fft1_ws (four launches, the intermediate through memory).  Round 2 always composed.
    def _fused(self):
        return all(_ops._is_pow2_engine(K) and K >= 16 for K in (self._Kx, self._Ky))

    def __call__(self, ary):
        a = _promote_input(ary, _cdtype()).to(_cdtype())
        if self._fused() and a.dim() == 2:
            if self._x_first:
                out = _ops.fft1_ramp(a, self._Kx, 1, self._x_direction, pre=self._pre_x, post=self._post_x, out_len=self._Mx)
                return _ops.fft1_ramp(out, self._Ky, 0, self._y_direction, pre=self._pre_y, post=self._post_y, out_len=self._My,
                                      scale=self.norm)
            out = _ops.fft1_ramp(a, self._Ky, 0, self._y_direction, pre=self._pre_y, post=self._post_y, out_len=self._My)
            return _ops.fft1_ramp(out, self._Kx, 1, self._x_direction, pre=self._pre_x, post=self._post_x, out_len=self._Mx,
                                  scale=self.norm)
        out = _ops.scale_sep(a, row_vec=self._pre_y, col_vec=self._pre_x)
        # fft(ary, K) or ifft(ary, K) * K == the unnormalised transform of either sign
        if self._x_first:
            out = _ops.fft1(out, self._Kx, axis=1, direction=self._x_direction, out_len=self._Mx)
            out = _ops.fft1(out, self._Ky, axis=0, direction=self._y_direction, out_len=self._My)
        else:
            out = _ops.fft1(out, self._Ky, axis=0, direction=self._y_direction, out_len=self._My)
            out = _ops.fft1(out, self._Kx, axis=1, direction=self._x_direction, out_len=self._Mx)
        return _ops.scale_sep(out, row_vec=self._post_y, col_vec=self._post_x, scale=self.norm)

    def adjoint(self, grad):
        g = _promote_input(grad, _cdtype()).to(_cdtype())
        if self._fused() and g.dim() == 2:
            # adjoint of (ramp, pad, transform, crop, ramp) per axis: conj(post) . g, zero pad to K, opposite sign, keep the first N, conj(pre)
            if self._x_first:
                out = _ops.fft1_ramp(g, self._Ky, 0, -self._y_direction, pre=self._post_y, post=self._pre_y, out_len=self._Ny, conj=True)
                return _ops.fft1_ramp(out, self._Kx, 1, -self._x_direction, pre=self._post_x, post=self._pre_x, out_len=self._Nx, conj=True,
                                      scale=self.norm)
            out = _ops.fft1_ramp(g, self._Kx, 1, -self._x_direction, pre=self._post_x, post=self._pre_x, out_len=self._Nx, conj=True)
            return _ops.fft1_ramp(out, self._Ky, 0, -self._y_direction, pre=self._post_y, post=self._pre_y, out_len=self._Ny, conj=True,
                                  scale=self.norm)
        out = _ops.scale_sep(g, row_vec=self._post_y, col_vec=self._post_x, row_conj=True, col_conj=True)
        # adjoint of a cropped unnormalised transform: zero pad to K, opposite sign, keep the first N
        if self._x_first:
            out = _ops.fft1(out, self._Ky, axis=0, direction=-self._y_direction, out_len=self._Ny)
            out = _ops.fft1(out, self._Kx, axis=1, direction=-self._x_direction, out_len=self._Nx)
        else:
            out = _ops.fft1(out, self._Kx, axis=1, direction=-self._x_direction, out_len=self._Nx)
            out = _ops.fft1(out, self._Ky, axis=0, direction=-self._y_direction, out_len=self._Ny)
        return _ops.scale_sep(out, row_vec=self._pre_y, col_vec=self._pre_x, row_conj=True, col_conj=True,
                              scale=self.norm)

    def nbytes(self):
        return sum(a.numel() * a.element_size() for a in (self._pre_x, self._pre_y, self._post_x, self._post_y))


def _uniform_spacing(values, name):
    """prysm/fttools.py:538-552."""
    if values.numel() < 2:
        raise ValueError(f'{name} must contain at least two samples')
    v = values.to(torch.float64)
    spacing = float(v[1] - v[0])
    if spacing == 0:
        raise ValueError(f'{name} must have nonzero spacing')
    tolerance = 32 * float(truenp.finfo(config.precision).eps)
    scale = max(1.0, abs(float(v[0])), abs(float(v[-1])), abs(spacing))
    diffs = torch.diff(v)
    if not bool(torch.all(torch.abs(diffs - spacing) <= tolerance * scale + tolerance * abs(spacing))):
        raise ValueError(f'{name} must be uniformly spaced')
    return spacing


def _fft_compatible_length(alpha, N, M, name):
    """prysm/fttools.py:555-571."""
    inv_alpha = 1 / abs(alpha)
    K = round(inv_alpha)
    tolerance = 32 * float(truenp.finfo(config.precision).eps)
    if not math.isclose(inv_alpha, K, rel_tol=tolerance, abs_tol=tolerance):
        raise ValueError(f'{name} spacings are not FFT-compatible: '
                         'abs(input spacing * output spacing) must be 1/integer')
    if K < max(N, M):
        raise ValueError(f'{name} requires FFT length {K}, smaller than input/output length {max(N, M)}')
    return K


def fourier_resample(f, zoom):
    """Resample f via Fourier methods (truncated sinc interpolation) (prysm/fttools.py:538-593).

    One fused centred fft2 (real input read as it is) and one matrix-DFT GEMM pair onto the zoomed grid.
    """
    if zoom == 1:
        return f
    if isinstance(zoom, (float, int)):
        zoom = (zoom, zoom)
    elif not isinstance(zoom, tuple):
        zoom = tuple(float(z) for z in zoom)
    if len(zoom) != 2 or any(z <= 0 for z in zoom):
        raise ValueError('zoom must contain two positive values')
    x = L.as_field(f)
    real = not x.is_complex()
    m, n = x.shape
    M = int(m * zoom[0])
    N = int(n * zoom[1])
    if M < 1 or N < 1:
        raise ValueError('zoom produces an empty output')
    sh = (m // 2, n // 2)
    F = _ops.fft2(x, direction=-1, scale=1.0, in_shift=sh, out_shift=sh)
    xx = fftrange(n, dtype=config.compute_precision)
    yy = fftrange(m, dtype=config.compute_precision)
    fx = fftrange(N, dtype=config.compute_precision) * (1.0 / zoom[1] / n)
    fy = fftrange(M, dtype=config.compute_precision) * (1.0 / zoom[0] / m)
    fprime = MDFT(xx, yy, fx, fy, sign=+1, norm=1.0 / (m * n))(F)
    return fprime.real if real else fprime
