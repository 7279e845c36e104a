"""Wavefront: the metadata-carrying object API (prysm/propagation/wavefront.py).

Same constructor, attributes (data, wavelength, dx, space), methods, unit conventions
(wavelength um, pupil dx mm, focal dx um, efl / z mm, OPD nm) and error behaviour as the
reference; ``data`` is a torch tensor in MI355X HBM.
"""
import copy
import math
import numbers
import operator

import numpy as np
import torch

from .. import _lib as L
from .. import _ops
from ..graph import sequenced
from ._kernels import field_multiply
from .._richdata import RichData
from ._kernels import phase_prefix
from .fft import (
    focus, focus_adjoint, unfocus, unfocus_adjoint, focus_intensity, focus_from_amp_and_phase,
    pupil_sample_to_psf_sample, psf_sample_to_pupil_sample,
)
from .dft import (
    prepare_executor, prepare_multiresolution, focus_dft, focus_dft_adjoint, unfocus_dft, unfocus_dft_adjoint,
)
from .angular_spectrum import angular_spectrum, angular_spectrum_adjoint
from .coronagraph import (to_fpm_and_back, to_fpm_and_back_adjoint, to_fpm_and_back_multiresolution,
                          to_fpm_and_back_multiresolution_adjoint, babinet, babinet_adjoint)
from ..fttools import pad2d, crop_center


def _field_data(field):
    """Return array data from a Wavefront-like field (pass through otherwise)."""
    if isinstance(field, Wavefront):
        return field.data
    return field


def _real_opd(phase):
    """OPD array -> real device tensor (fp32 stays fp32, everything else fp64)."""
    t = L.as_device(phase)
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    return t


def _synth_args(amplitude, phase):
    """(amplitude, OPD, complex dtype) of from_amp_and_phase with numpy's result-type rules applied."""
    opd = _real_opd(phase)
    amp = None if amplitude is None else L.as_device(amplitude)
    cd = L._COMPLEX_OF[opd.dtype]
    if amp is not None and amp.dtype == torch.float64 and cd == torch.complex64:
        cd, opd = torch.complex128, opd.to(torch.float64)
    return amp, opd, cd


class Wavefront:
    """(Complex) representation of a wavefront (wavefront.py:35-56)."""

    __array_ufunc__ = None   # numpy_array <op> wavefront defers to the reflected operators below (one device op, not one per element)

    def __init__(self, cmplx_field, wavelength, dx, space='pupil'):
        """cmplx_field: array (numpy is uploaded); wavelength um; dx mm (pupil) or um (psf)."""
        self._synth = None
        if cmplx_field is None or isinstance(cmplx_field, numbers.Number):
            self.data = cmplx_field
        else:
            self.data = L.as_device(cmplx_field)
        self.wavelength = wavelength
        self.dx = dx
        self.space = space

    # `data` of a wavefront made by from_amp_and_phase is materialised on first use: focus / focus_intensity of such a
    # wavefront synthesise amp * exp(i k opd) inside the transform instead (PM_FLAG_SYNTH_INPUT) and never touch it.
    @property
    def data(self):
        if self._data is None and self._synth is not None:
            amp, opd, k, cd = self._synth
            self._data = _ops.pupil_synth(amp, opd, k, cd)
        return self._data

    @data.setter
    def data(self, value):
        self._data = value
        self._synth = None

    def __array__(self, dtype=None, copy=None):
        """numpy conversion = the field (keeps np.asarray(wavefront) from building an object array)."""
        if isinstance(self.data, torch.Tensor):
            from ..mathops import array_to_true_numpy
            a = array_to_true_numpy(self.data)      # (joins an open graph.sequence() block first)
        else:
            a = np.asarray(self.data)
        return a.astype(dtype) if dtype is not None else a

    def _lazy(self):
        """(amp, opd, k, complex dtype) while the field has not been materialised, else None"""
        return self._synth if self._data is None else None

    def _shape(self):
        lz = self._lazy()
        return tuple(lz[1].shape) if lz is not None else tuple(self.data.shape)

    def _fusable(self, Q):
        """(amp, opd, k) when the pupil can be synthesised inside the FFT (not yet materialised, power-of-two padded width),
        else None."""
        if self._data is not None or self._synth is None:
            return None
        amp, opd, k, cd = self._synth
        N = math.ceil(opd.shape[1] * Q)
        if cd == L._COMPLEX_OF[opd.dtype] and _ops.synth_supported(opd, amp, N):
            return amp, opd, k
        return None

    @classmethod
    def from_amp_and_phase(cls, amplitude, phase, wavelength, dx):
        """P = amplitude * exp(i 2 pi / (wavelength 1e3) * phase_nm) (wavefront.py:58-79).

        One fused synthesis kernel; with phase None the amplitude is returned unchanged, as in the
        reference.
        """
        if phase is not None:
            amp, opd, cd = _synth_args(amplitude, phase)
            k = 2 * math.pi / wavelength / 1e3
            if amp is not None and amp.is_complex():
                P = _ops.cmul(amp.to(cd), _ops.pupil_synth(None, opd, k, cd))
            else:
                wf = cls(None, wavelength, dx)       # lazy: see the `data` property
                wf._synth = (amp, opd, k, cd)
                return wf
        else:
            P = amplitude
        return cls(P, wavelength, dx)

    @classmethod
    def phase_screen(cls, phase, wavelength, dx):
        """exp(i 2 pi / (wavelength 1e3) * phase_nm) (wavefront.py:81-96)."""
        opd = _real_opd(phase)
        wf = cls(None, wavelength, dx)           # lazy, like from_amp_and_phase: a screen is usually multiplied into a pupil at once,
        wf._synth = (None, opd, 2 * math.pi / wavelength / 1e3, L._COMPLEX_OF[opd.dtype])     # see __numerical_operation__
        return wf

    @classmethod
    def thin_lens(cls, f, wavelength, x, y):
        """Quadratic phase screen exp(-i 2 pi/(wavelength/1e3) r^2/(2f)) (wavefront.py:98-144)."""
        w = wavelength / 1e3
        xt, yt = _real_opd(x), _real_opd(y)
        if xt.dtype != yt.dtype:
            xt, yt = xt.to(torch.float64), yt.to(torch.float64)
        c = -2 * math.pi / w / (2 * f)
        screen = _ops.quadratic_phase(xt, yt, c, L._COMPLEX_OF[xt.dtype])
        dx = float(xt[0, 1] - xt[0, 0])
        return cls(cmplx_field=screen, wavelength=wavelength, dx=dx, space='pupil')

    @classmethod
    def thin_lens_adjoint(cls, f, wavelength, x, y, wf_bar):
        """Adjoint of thin_lens with respect to the focal length f (wavefront.py:244-279): a scalar."""
        L_bar = L.as_complex(_field_data(wf_bar))
        screen = cls.thin_lens(f, wavelength, x, y).data
        if L_bar.dtype != screen.dtype:
            L_bar = L_bar.to(torch.complex128)
            screen = screen.to(torch.complex128)
        w = wavelength / 1e3
        xt, yt = _real_opd(x), _real_opd(y)
        rsq = xt * xt + yt * yt
        coeff = math.pi / (w * f * f)
        return coeff * torch.sum(rsq * _ops.cmul(L_bar, screen, conj_b=True).imag)

    @property
    @sequenced
    def intensity(self):
        """Intensity, abs(w)^2 (wavefront.py:146-151)."""
        data = self.data
        if data.is_complex():
            out = _ops.abs2(data)
        else:
            d = data.to(torch.float64) if data.dtype == torch.bool else data
            out = d * d
        return RichData(out, self.dx, self.wavelength)

    @property
    @sequenced
    def phase(self):
        """Phase, angle(w)."""
        return RichData(torch.angle(self.data), self.dx, self.wavelength)

    @property
    @sequenced
    def real(self):
        """re(w)."""
        return RichData(self.data.real if self.data.is_complex() else self.data, self.dx, self.wavelength)

    @property
    @sequenced
    def imag(self):
        """im(w)."""
        return RichData(self.data.imag if self.data.is_complex() else torch.zeros_like(self.data), self.dx,
                        self.wavelength)

    def copy(self):
        """Return a (deep) copy of this instance."""
        return copy.deepcopy(self)

    def from_amp_and_phase_adjoint_phase(self, wf_bar):
        """Adjoint of from_amp_and_phase with respect to phase (wavefront.py:172-188)."""
        k = phase_prefix(self.wavelength)
        return k * _ops.cmul(L.as_complex(wf_bar.data).to(self.data.dtype), self.data, conj_b=True).imag

    def from_amp_and_phase_adjoint_amp(self, wf_bar, phase=None):
        """Adjoint of from_amp_and_phase with respect to amplitude (wavefront.py:190-222)."""
        if phase is not None:
            opd = _real_opd(phase)
            S = _ops.pupil_synth(None, opd, 2 * math.pi / self.wavelength / 1e3, self.data.dtype)
            return _ops.cmul(L.as_complex(wf_bar.data).to(S.dtype), S, conj_b=True).real
        absP = torch.abs(self.data)
        nonzero = absP > 0
        grad = _ops.cmul(L.as_complex(wf_bar.data).to(self.data.dtype), self.data, conj_b=True).real
        return torch.where(nonzero, grad / torch.where(nonzero, absP, torch.ones_like(absP)), torch.zeros_like(grad))

    def phase_screen_adjoint_phase(self, wf_bar):
        """Adjoint of phase_screen with respect to phase (wavefront.py:224-240)."""
        return self.from_amp_and_phase_adjoint_phase(wf_bar)

    def intensity_adjoint(self, intensity_bar):
        """Adjoint of intensity: 2 * Ibar * E (wavefront.py:282-298)."""
        ibar = _field_data(intensity_bar)
        if isinstance(ibar, RichData):
            ibar = ibar.data
        ibar = L.as_device(ibar)
        E = self.data
        if (ibar.dim() == 2 and E.dim() == 2 and E.is_complex() and ibar.dtype == L._REAL_OF[E.dtype] and ibar.shape == E.shape and
                ibar.stride(1) == 1 and E.stride(1) == 1):
            Gbar = _ops.rmul(ibar, E, 2.0)        # one sweep (pm_rmul) instead of two torch ones
        else:
            Gbar = 2 * ibar * E
        return Wavefront(Gbar, self.wavelength, self.dx, self.space)

    def pad2d(self, Q, value=0, mode='constant', out_shape=None, inplace=True):
        """Pad the wavefront (wavefront.py:300-332)."""
        padded = pad2d(self.data, Q=Q, value=value, mode=mode, out_shape=out_shape)
        if inplace:
            self.data = padded
            return self
        return Wavefront(padded, self.wavelength, self.dx, self.space)

    def crop(self, out_shape, inplace=True):
        """Crop the wavefront to the centermost out_shape (wavefront.py:334-358)."""
        cropped = crop_center(self.data, out_shape)
        if inplace:
            self.data = cropped
            return self
        return Wavefront(cropped, self.wavelength, self.dx, self.space)

    @sequenced
    def __numerical_operation__(self, other, op, reverse=False):
        """Apply an operation to this wavefront with another piece of data (wavefront.py:360-383)."""
        func = getattr(operator, op)
        if isinstance(other, Wavefront):
            criteria = [
                abs(self.dx - other.dx) / self.dx * 100 < 0.1,
                self._shape() == other._shape(),
                self.wavelength == other.wavelength,
                self.space == other.space,
            ]
            if not all(criteria):
                raise ValueError('all physicality criteria not met: sample spacing, shape, wavelength, or space different.')
            if op == 'mul':
                la, lb = self._lazy(), other._lazy()
                if la is not None and lb is not None and la[2] == lb[2] and la[3] == lb[3] and la[1].dtype == lb[1].dtype:
                    # A1 exp(i k W1) * A2 exp(i k W2) = (A1 A2) exp(i k (W1 + W2)): two wavefronts that have not been materialised
                    # (a pupil from from_amp_and_phase times a phase_screen -- a deformable mirror, an aberration, a lens) stay ONE
                    # lazy wavefront: a real addition instead of two synthesis sweeps and a complex product (round 6)
                    amp = la[0] if lb[0] is None else (lb[0] if la[0] is None else la[0] * lb[0])
                    wf = Wavefront(None, self.wavelength, self.dx, self.space)
                    wf._synth = (amp, la[1] + lb[1], la[2], la[3])
                    return wf
                data = field_multiply(self.data, other.data)   # the hot pointwise products run in the HIP kernels (pm_cmul, pm_rmul)
            else:
                a, b = self.data, other.data
                data = func(b, a) if reverse else func(a, b)
        elif isinstance(other, (torch.Tensor, np.ndarray)):
            o = L.as_device(other)
            if op == 'mul' and isinstance(self.data, torch.Tensor):
                data = field_multiply(self.data, o)
            else:
                data = func(o, self.data) if reverse else func(self.data, o)
        elif isinstance(other, numbers.Number):
            data = func(other, self.data) if reverse else func(self.data, other)
        else:
            raise TypeError(f'unsupported operand type(s) for {op}: \'Wavefront\' and {type(other)}')
        return Wavefront(dx=self.dx, wavelength=self.wavelength, cmplx_field=data, space=self.space)

    def __mul__(self, other):
        return self.__numerical_operation__(other, 'mul')

    def __rmul__(self, other):
        return self.__numerical_operation__(other, 'mul', reverse=True)

    def __truediv__(self, other):
        return self.__numerical_operation__(other, 'truediv')

    def __rtruediv__(self, other):
        return self.__numerical_operation__(other, 'truediv', reverse=True)

    def __add__(self, other):
        return self.__numerical_operation__(other, 'add')

    def __radd__(self, other):
        return self.__numerical_operation__(other, 'add', reverse=True)

    def __sub__(self, other):
        return self.__numerical_operation__(other, 'sub')

    def __rsub__(self, other):
        return self.__numerical_operation__(other, 'sub', reverse=True)

    def free_space(self, dz=np.nan, Q=1, tf=None):
        """Plane-to-plane free space propagation by angular spectrum (wavefront.py:413-443)."""
        if np.isnan(dz) and tf is None:
            raise ValueError('dz must be provided if tf is None')
        out = angular_spectrum(self.data, wvl=self.wavelength, dx=self.dx, z=dz, Q=Q, tf=tf)
        return Wavefront(out, self.wavelength, self.dx, self.space)

    def free_space_adjoint(self, dz=np.nan, Q=1, tf=None):
        """Apply the adjoint of free_space (wavefront.py:445-476)."""
        if np.isnan(dz) and tf is None:
            raise ValueError('dz must be provided if tf is None')
        out = angular_spectrum_adjoint(self.data, wvl=self.wavelength, dx=self.dx, z=dz, Q=Q, tf=tf)
        return Wavefront(out, self.wavelength, self.dx, self.space)

    def focus(self, efl, Q=2):
        """Pupil to PSF plane propagation by FFT (wavefront.py:478-504)."""
        if self.space != 'pupil':
            raise ValueError('can only propagate from a pupil to psf plane')
        fus = self._fusable(Q)
        if fus is not None:
            data = focus_from_amp_and_phase(fus[0], fus[1], fus[2], Q)
        else:
            data = focus(self.data, Q=Q)
        dx = pupil_sample_to_psf_sample(self.dx, data.shape[1], self.wavelength, efl)
        return Wavefront(data, self.wavelength, dx, space='psf')

    def focus_intensity(self, efl, Q=2):
        """``self.focus(efl, Q).intensity`` with |.|^2 fused into the transform (no complex PSF in memory)."""
        if self.space != 'pupil':
            raise ValueError('can only propagate from a pupil to psf plane')
        fus = self._fusable(Q)
        if fus is not None:
            data = focus_intensity(fus[1], Q=Q, synth=(fus[0], fus[2]))
        else:
            data = focus_intensity(self.data, Q=Q)
        dx = pupil_sample_to_psf_sample(self.dx, data.shape[1], self.wavelength, efl)
        return RichData(data, dx, self.wavelength)

    def focus_adjoint(self, efl, Q=2):
        """Apply the adjoint of focus (wavefront.py:506-532)."""
        if self.space != 'psf':
            raise ValueError('can only apply adjoint from a psf to pupil plane')
        samples = self.data.shape[1]
        data = focus_adjoint(self.data, Q=Q)
        dx = psf_sample_to_pupil_sample(self.dx, samples, self.wavelength, efl)
        return Wavefront(data, self.wavelength, dx, space='pupil')

    def unfocus(self, efl, Q=2):
        """PSF to pupil plane propagation by FFT (wavefront.py:534-560)."""
        if self.space != 'psf':
            raise ValueError('can only propagate from a psf to pupil plane')
        data = unfocus(self.data, Q=Q)
        dx = psf_sample_to_pupil_sample(self.dx, data.shape[1], self.wavelength, efl)
        return Wavefront(data, self.wavelength, dx, space='pupil')

    def unfocus_adjoint(self, efl, Q=2):
        """Apply the adjoint of unfocus (wavefront.py:562-588)."""
        if self.space != 'pupil':
            raise ValueError('can only apply adjoint from a pupil to psf plane')
        samples = self.data.shape[1]
        data = unfocus_adjoint(self.data, Q=Q)
        dx = pupil_sample_to_psf_sample(self.dx, samples, self.wavelength, efl)
        return Wavefront(data, self.wavelength, dx, space='psf')

    def prepare_executor(self, efl, dx, samples, shift=(0, 0), kind='mdft'):
        """Build a reusable MDFT, CZT, or FFTDFT focus executor (wavefront.py:590-641)."""
        if isinstance(samples, int):
            samples = (samples, samples)
        if self.space == 'pupil':
            return prepare_executor(pupil_dx=self.dx, pupil_samples=tuple(self.data.shape), focal_dx=dx,
                                    focal_samples=samples, wavelength=self.wavelength, efl=efl, focal_shift=shift,
                                    kind=kind)
        elif self.space == 'psf':
            return prepare_executor(pupil_dx=dx, pupil_samples=samples, focal_dx=self.dx,
                                    focal_samples=tuple(self.data.shape), wavelength=self.wavelength, efl=efl,
                                    focal_shift=shift, kind=kind)
        raise ValueError(f"unknown space {self.space!r}")

    def prepare_multiresolution(self, efl, focal_dx, focal_samples, num_levels, scaling=4.0, fine_samples=None,
                                window=(0.2, 0.7), kind='mdft'):
        """Build a MultiResolutionExecutor for this pupil-plane wavefront (wavefront.py:643-677)."""
        if self.space != 'pupil':
            raise ValueError('multiresolution propagation begins at a pupil plane')
        return prepare_multiresolution(pupil_dx=self.dx, pupil_samples=tuple(self.data.shape), focal_dx=focal_dx,
                                       focal_samples=focal_samples, wavelength=self.wavelength, efl=efl,
                                       num_levels=num_levels, scaling=scaling, fine_samples=fine_samples, window=window,
                                       kind=kind)

    def focus_dft(self, executor):
        """Pupil -> PSF propagation via a precomputed executor (wavefront.py:679-696)."""
        if self.space != 'pupil':
            raise ValueError('can only propagate from a pupil to psf plane')
        data = focus_dft(self.data, executor)
        return Wavefront(dx=executor.focal_dx, cmplx_field=data, wavelength=self.wavelength, space='psf')

    def focus_dft_intensity(self, executor, out=None, weight=1.0):
        """``self.focus_dft(executor).intensity`` with the modulus (and an optional weighted accumulate into `out`) in the epilogue of
        the executor's last product -- no complex focal field in memory (MDFT executors: pm_cgemm_abs2; others compose).  The
        counterpart of focus_intensity for the fixed-sampling focus; the polychromatic driver's variant M uses it per wavelength."""
        if self.space != 'pupil':
            raise ValueError('can only propagate from a pupil to psf plane')
        if hasattr(executor, 'intensity'):
            data = executor.intensity(self.data, out=out, weight=weight)
        else:
            E = focus_dft(self.data, executor)
            if out is None:
                data = _ops.abs2(E)
                data = data if weight == 1.0 else data * weight
            else:
                data = _ops.abs2(E, out=out, weight=weight)
        return RichData(data, executor.focal_dx, self.wavelength)

    def focus_dft_adjoint(self, executor):
        """Apply the adjoint of focus_dft (wavefront.py:698-718)."""
        if self.space != 'psf':
            raise ValueError('can only apply adjoint from a psf to pupil plane')
        data = focus_dft_adjoint(self.data, executor)
        return Wavefront(dx=executor.pupil_dx, cmplx_field=data, wavelength=self.wavelength, space='pupil')

    def unfocus_dft(self, executor):
        """PSF -> pupil propagation via a precomputed executor (wavefront.py:720-737)."""
        if self.space != 'psf':
            raise ValueError('can only propagate from a psf to pupil plane')
        data = unfocus_dft(self.data, executor)
        return Wavefront(dx=executor.pupil_dx, cmplx_field=data, wavelength=self.wavelength, space='pupil')

    def unfocus_dft_adjoint(self, executor):
        """Apply the adjoint of unfocus_dft (wavefront.py:739-757)."""
        if self.space != 'pupil':
            raise ValueError('can only apply adjoint from a pupil to psf plane')
        data = unfocus_dft_adjoint(self.data, executor)
        return Wavefront(dx=executor.focal_dx, cmplx_field=data, wavelength=self.wavelength, space='psf')

    def to_fpm_and_back(self, fpm, executor, return_more=False):
        """Propagate to a focal plane mask, apply it, and return (wavefront.py:759-789)."""
        fpm = _field_data(fpm)
        pak = to_fpm_and_back(self.data, fpm=fpm, executor=executor, return_more=return_more)
        if return_more:
            at_next_pupil, at_fpm, after_fpm = pak
            return (Wavefront(at_next_pupil, self.wavelength, self.dx, self.space),
                    Wavefront(at_fpm, self.wavelength, executor.focal_dx, 'psf'),
                    Wavefront(after_fpm, self.wavelength, executor.focal_dx, 'psf'))
        return Wavefront(pak, self.wavelength, self.dx, self.space)

    def to_fpm_and_back_adjoint(self, fpm, executor, return_more=False, return_fpm_grad=False, field_at_fpm=None):
        """Apply the adjoint of to_fpm_and_back (wavefront.py:789-842): self is the gradient at the next pupil.

        Returns the gradient at the input pupil; with return_more also the two focal-plane intermediates, with
        return_fpm_grad (needs field_at_fpm of the forward pass) also the mask gradient -- focal-plane quantities come
        back as psf-space Wavefronts sampled at executor.focal_dx, like the reference.
        """
        pak = to_fpm_and_back_adjoint(self.data, fpm=_field_data(fpm), executor=executor, return_more=return_more,
                                      return_fpm_grad=return_fpm_grad, field_at_fpm=_field_data(field_at_fpm))
        if not (return_more or return_fpm_grad):
            return Wavefront(pak, self.wavelength, self.dx, self.space)
        first, *focal = pak
        return (Wavefront(first, self.wavelength, self.dx, self.space),
                *(Wavefront(f, self.wavelength, executor.focal_dx, 'psf') for f in focal))

    def to_fpm_and_back_multiresolution(self, fpm, executor, return_more=False):
        """Propagate to a focal plane mask and back at multiple resolutions (wavefront.py:852-885)."""
        if self.space != 'pupil':
            raise ValueError('can only propagate from a pupil to psf plane')
        pak = to_fpm_and_back_multiresolution(self.data, fpm, executor, return_more=return_more)
        if not return_more:
            return Wavefront(pak, self.wavelength, self.dx, self.space)
        out, at_fpm, after_fpm = pak
        out = Wavefront(out, self.wavelength, self.dx, self.space)
        at_fpm = [Wavefront(f, self.wavelength, ex.focal_dx, 'psf') for f, ex in zip(at_fpm, executor.executors)]
        after_fpm = [Wavefront(f, self.wavelength, ex.focal_dx, 'psf') for f, ex in zip(after_fpm, executor.executors)]
        return out, at_fpm, after_fpm

    def to_fpm_and_back_multiresolution_adjoint(self, fpm, executor, return_more=False, return_fpm_grad=False,
                                                field_at_fpm=None):
        """Apply the adjoint of to_fpm_and_back_multiresolution (wavefront.py:887-944)."""
        if field_at_fpm is not None:
            field_at_fpm = [_field_data(f) for f in field_at_fpm]
        pak = to_fpm_and_back_multiresolution_adjoint(self.data, fpm, executor, return_more=return_more,
                                                      return_fpm_grad=return_fpm_grad, field_at_fpm=field_at_fpm)

        def _psf_wrap(fields):
            return [Wavefront(f, self.wavelength, ex.focal_dx, 'psf') for f, ex in zip(fields, executor.executors)]

        if return_more:
            if return_fpm_grad:
                Eabar, Ebbars, intermediates, fpm_bars = pak
            else:
                Eabar, Ebbars, intermediates = pak
            Eabar = Wavefront(Eabar, self.wavelength, self.dx, self.space)
            Ebbars = _psf_wrap(Ebbars)
            intermediates = _psf_wrap(intermediates)
            if return_fpm_grad:
                return Eabar, Ebbars, intermediates, _psf_wrap(fpm_bars)
            return Eabar, Ebbars, intermediates
        elif return_fpm_grad:
            Eabar, fpm_bars = pak
            return Wavefront(Eabar, self.wavelength, self.dx, self.space), _psf_wrap(fpm_bars)
        return Wavefront(pak, self.wavelength, self.dx, self.space)

    def babinet(self, lyot, fpm, executor, return_more=False):
        """Propagate through a Lyot-style coronagraph using Babinet's principle (wavefront.py:952-1000)."""
        fpm, lyot = _field_data(fpm), _field_data(lyot)
        pak = babinet(self.data, lyot=lyot, fpm=fpm, executor=executor, return_more=return_more)
        if return_more:
            after_lyot, at_fpm, after_fpm, at_lyot = pak
            return (Wavefront(after_lyot, self.wavelength, self.dx, self.space),
                    Wavefront(at_fpm, self.wavelength, executor.focal_dx, 'psf'),
                    Wavefront(after_fpm, self.wavelength, executor.focal_dx, 'psf'),
                    Wavefront(at_lyot, self.wavelength, self.dx, self.space))
        return Wavefront(pak, self.wavelength, self.dx, self.space)

    def babinet_adjoint(self, lyot, fpm, executor, field_at_fpm=None, field_at_lyot=None, return_fpm_grad=False,
                        return_lyot_grad=False):
        """Apply the adjoint of babinet (wavefront.py:987-1048): self is the gradient after the Lyot stop.

        Returns the gradient at the input pupil, followed -- in this order, when asked for -- by the focal-plane-mask
        gradient (psf space, executor.focal_dx; needs field_at_fpm) and the Lyot-stop gradient (this wavefront's plane;
        needs field_at_lyot).
        """
        pak = babinet_adjoint(self.data, lyot=_field_data(lyot), fpm=_field_data(fpm), executor=executor,
                              field_at_fpm=_field_data(field_at_fpm), field_at_lyot=_field_data(field_at_lyot),
                              return_fpm_grad=return_fpm_grad, return_lyot_grad=return_lyot_grad)
        if not (return_fpm_grad or return_lyot_grad):
            return Wavefront(pak, self.wavelength, self.dx, self.space)
        grads = list(pak)
        out = [Wavefront(grads.pop(0), self.wavelength, self.dx, self.space)]
        if return_fpm_grad:
            out.append(Wavefront(grads.pop(0), self.wavelength, executor.focal_dx, 'psf'))
        if return_lyot_grad:
            out.append(Wavefront(grads.pop(0), self.wavelength, self.dx, self.space))
        return tuple(out)
