"""Numerical optical propagation on the MI355X -- the counterpart of prysm/propagation/__init__.py.

Files:
- fft               FFT-based pupil <-> focal propagation (one fused device call per propagation)
- dft               matrix-DFT / chirp-Z / FFT-DFT propagation with arbitrary sampling (MFMA GEMMs)
- angular_spectrum  plane-to-plane free space propagation
- wavefront         the Wavefront type, object oriented interface
"""
from . import fft as _fft, dft as _dft, angular_spectrum as _as, coronagraph as _cor

# public names, grouped by the module that implements them (same names as prysm.propagation)
_EXPORTS = {
    _fft: ('focus focus_adjoint unfocus unfocus_adjoint focus_intensity Q_for_sampling pupil_sample_to_psf_sample '
           'psf_sample_to_pupil_sample'),
    _dft: ('coordinates_for_focus prepare_executor prepare_multiresolution MultiResolutionExecutor unit_cell_focal_grid '
           'focus_dft focus_dft_adjoint unfocus_dft unfocus_dft_adjoint focus_fixed_sampling'),
    _as: 'angular_spectrum angular_spectrum_adjoint angular_spectrum_transfer_function fresnel_number talbot_distance',
    _cor: ('to_fpm_and_back to_fpm_and_back_adjoint to_fpm_and_back_multiresolution '
           'to_fpm_and_back_multiresolution_adjoint vortex_phase_mask prepare_measured_fpm babinet babinet_adjoint'),
}
for _mod, _names in _EXPORTS.items():
    for _n in _names.split():
        globals()[_n] = getattr(_mod, _n)
del _mod, _names, _n

from .wavefront import Wavefront  # noqa: E402
from ._kernels import phase_prefix  # noqa: E402
