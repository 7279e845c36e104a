// Column-pass FFT kernels, float precision (explicit instantiation; see fft_kernels.h).
#include "fft_kernels.h"
namespace pm {
template <> int launch_col_tiled<float>(int logm, const ColLoadTiled<float>& l, const ColStoreNat<float>& s, const cx<float>* tw, int ntiles, hipStream_t st) {
    return launch_fft<float, true>(logm, l, s, tw, ntiles, st);
}
template <> int launch_col_nat<float>(int logm, const ColLoadNat<float>& l, const ColStoreNat<float>& s, const cx<float>* tw, int ntiles, hipStream_t st) {
    return launch_fft<float, true>(logm, l, s, tw, ntiles, st);
}
}  // namespace pm
