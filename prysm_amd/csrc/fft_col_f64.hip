// Column-pass FFT kernels, double precision (explicit instantiation; see fft_kernels.h).
#include "fft_kernels.h"
namespace pm {
template <> int launch_col_tiled<double>(int logm, int var, const ColLoadTiled<double>& l, const ColStoreNat<double>& s, const cx<double>* tw, int ntiles, int log_g, hipStream_t st, int nbatch) {
    return launch_fft<double, true>(logm, var, l, s, tw, ntiles, log_g, st, nbatch);
}
template <> int launch_col_nat<double>(int logm, int var, const ColLoadNat<double>& l, const ColStoreNat<double>& s, const cx<double>* tw, int ntiles, int log_g, hipStream_t st, int nbatch) {
    return launch_fft<double, true>(logm, var, l, s, tw, ntiles, log_g, st, nbatch);
}
template <> int launch_col_mul<double>(int logm, const ColLoadTiled<double>& l, const MidMul<double>& m, const ColStoreTiled<double>& s, const cx<double>* tw, int ntiles, int log_g, hipStream_t st, int nbatch, int mode) {
    return launch_col_mul_impl<double>(logm, l, m, s, tw, ntiles, log_g, st, nbatch, mode);
}
template <> int launch_col_mul_crop<double>(int logm, const ColLoadTiled<double>& l, const MidMul<double>& m, const ColStoreTiledCrop<double>& s, const cx<double>* tw, int ntiles, int log_g, hipStream_t st) {
    return launch_col_mul_impl<double>(logm, l, m, s, tw, ntiles, log_g, st, 1);
}
}  // namespace pm
