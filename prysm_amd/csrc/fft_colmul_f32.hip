// Middle pass of the fused chain (column FFT x H x column IFFT), float precision (explicit instantiation; see fft_kernels.h).  Its own
// translation unit since round 4: fft_col_f32.hip is compiled with the packed complex64 arithmetic (PM_PACKED_F32), which these
// 128-register kernels pay for in spills.
#include "fft_kernels.h"
namespace pm {
template <> int launch_col_mul<float>(int logm, const ColLoadTiled<float>& l, const MidMul<float>& m, const ColStoreTiled<float>& s, const cx<float>* tw, int ntiles, int log_g, hipStream_t st, int nbatch, int mode) {
    return launch_col_mul_impl<float>(logm, l, m, s, tw, ntiles, log_g, st, nbatch, mode);
}
template <> int launch_col_mul_crop<float>(int logm, const ColLoadTiled<float>& l, const MidMul<float>& m, const ColStoreTiledCrop<float>& s, const cx<float>* tw, int ntiles, int log_g, hipStream_t st) {
    return launch_col_mul_impl<float>(logm, l, m, s, tw, ntiles, log_g, st, 1);
}
}  // namespace pm
