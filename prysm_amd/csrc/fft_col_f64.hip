// Column-pass FFT kernels, double precision (explicit instantiation; see fft_kernels.h).
#include "fft_kernels.h"
namespace pm {
template <> int launch_col_tiled<double>(int logm, const ColLoadTiled<double>& l, const ColStoreNat<double>& s, const cx<double>* tw, int ntiles, hipStream_t st) {
    return launch_fft<double, true>(logm, l, s, tw, ntiles, st);
}
template <> int launch_col_nat<double>(int logm, const ColLoadNat<double>& l, const ColStoreNat<double>& s, const cx<double>* tw, int ntiles, hipStream_t st) {
    return launch_fft<double, true>(logm, l, s, tw, ntiles, st);
}
}  // namespace pm
