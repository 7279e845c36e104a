"""Numerical optical propagation on the MI355X -- the counterpart of prysm/propagation/__init__.py.

Files:
- fft               FFT-based pupil <-> focal propagation (one fused device call per propagation)
- dft               matrix-DFT / chirp-Z / FFT-DFT propagation with arbitrary sampling (MFMA GEMMs)
- angular_spectrum  plane-to-plane free space propagation
- wavefront         the Wavefront type, object oriented interface
"""
from .fft import (
    focus,
    focus_adjoint,
    unfocus,
    unfocus_adjoint,
    focus_intensity,
    Q_for_sampling,
    pupil_sample_to_psf_sample,
    psf_sample_to_pupil_sample,
)
from .dft import (
    coordinates_for_focus,
    prepare_executor,
    prepare_multiresolution,
    MultiResolutionExecutor,
    unit_cell_focal_grid,
    focus_dft,
    focus_dft_adjoint,
    unfocus_dft,
    unfocus_dft_adjoint,
    focus_fixed_sampling,
)
from .angular_spectrum import (
    angular_spectrum,
    angular_spectrum_adjoint,
    angular_spectrum_transfer_function,
    fresnel_number,
    talbot_distance,
)
from .coronagraph import (
    to_fpm_and_back,
    to_fpm_and_back_adjoint,
    to_fpm_and_back_multiresolution,
    to_fpm_and_back_multiresolution_adjoint,
    vortex_phase_mask,
    babinet,
    babinet_adjoint,
)
from .wavefront import Wavefront
from ._kernels import phase_prefix

__all__ = [
    'focus', 'focus_adjoint', 'unfocus', 'unfocus_adjoint', 'focus_intensity', 'Q_for_sampling',
    'pupil_sample_to_psf_sample', 'psf_sample_to_pupil_sample', 'coordinates_for_focus',
    'prepare_executor', 'unit_cell_focal_grid', 'focus_dft', 'focus_dft_adjoint', 'unfocus_dft',
    'unfocus_dft_adjoint', 'focus_fixed_sampling', 'angular_spectrum', 'angular_spectrum_adjoint',
    'angular_spectrum_transfer_function', 'fresnel_number', 'talbot_distance', 'Wavefront', 'phase_prefix',
    'to_fpm_and_back', 'to_fpm_and_back_adjoint', 'to_fpm_and_back_multiresolution', 'vortex_phase_mask',
    'babinet', 'babinet_adjoint',
]
