// complex64 arithmetic on the packed fp32 instructions (pm_common.h PM_PACKED_F32: v_pk_add / mul / fma_f32 on the register pair of a
// complex value -- 8 instructions per radix-4 butterfly instead of 16, 2 per complex multiply instead of 4; 38 % fewer floating-point
// instructions in a kernel).  Measured per translation unit against the scalar build (tools/exp_ab_libs.py, profiles/r04/
// exp_ab_packed_*.log): in THIS one `focus` 2048^2 28.55 -> 27.8 us, 1024^2 17.0 -> 16.3, 8192^2 476 -> 472, 4096^2 unchanged; in the
// row kernels 4096^2 93.0 -> 94.3 and the pupil-synthesising row pass 102 -> 106 per wavelength (they stay scalar), in the real-input
// kernels mtf 74.7 -> 78.2 (scalar), in the mixed-radix kernels nothing (their stages wait on loads, not on butterflies).
#define PM_PACKED_F32
// Column-pass FFT kernels, float precision (explicit instantiation; see fft_kernels.h).
#include "fft_kernels.h"
namespace pm {
template <> int launch_col_tiled<float>(int logm, int var, const ColLoadTiled<float>& l, const ColStoreNat<float>& s, const cx<float>* tw, int ntiles, int log_g, hipStream_t st, int nbatch) {
    return launch_fft<float, true>(logm, var, l, s, tw, ntiles, log_g, st, nbatch);
}
template <> int launch_col_nat<float>(int logm, int var, const ColLoadNat<float>& l, const ColStoreNat<float>& s, const cx<float>* tw, int ntiles, int log_g, hipStream_t st, int nbatch) {
    return launch_fft<float, true>(logm, var, l, s, tw, ntiles, log_g, st, nbatch);
}
}  // namespace pm
