// Column-pass FFT kernels, float precision (explicit instantiation; see fft_kernels.h).
#include "fft_kernels.h"
namespace pm {
template <> int launch_col_tiled<float>(int logm, int var, const ColLoadTiled<float>& l, const ColStoreNat<float>& s, const cx<float>* tw, int ntiles, int log_g, hipStream_t st, int nbatch) {
    return launch_fft<float, true>(logm, var, l, s, tw, ntiles, log_g, st, nbatch);
}
template <> int launch_col_nat<float>(int logm, int var, const ColLoadNat<float>& l, const ColStoreNat<float>& s, const cx<float>* tw, int ntiles, int log_g, hipStream_t st, int nbatch) {
    return launch_fft<float, true>(logm, var, l, s, tw, ntiles, log_g, st, nbatch);
}
template <> int launch_col_mul<float>(int logm, const ColLoadTiled<float>& l, const MidMul<float>& m, const ColStoreTiled<float>& s, const cx<float>* tw, int ntiles, int log_g, hipStream_t st, int nbatch, int mode) {
    return launch_col_mul_impl<float>(logm, l, m, s, tw, ntiles, log_g, st, nbatch, mode);
}
template <> int launch_col_mul_crop<float>(int logm, const ColLoadTiled<float>& l, const MidMul<float>& m, const ColStoreTiledCrop<float>& s, const cx<float>* tw, int ntiles, int log_g, hipStream_t st) {
    return launch_col_mul_impl<float>(logm, l, m, s, tw, ntiles, log_g, st, 1);
}
}  // namespace pm
