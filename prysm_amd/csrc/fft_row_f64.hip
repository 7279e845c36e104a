// Row-pass FFT kernels, double precision (explicit instantiation; see fft_kernels.h).
#include "fft_kernels.h"
namespace pm {
template <> int launch_row_tiled<double>(int logn, const RowLoadNat<double>& l, const RowStoreTiled<double>& s, const cx<double>* tw, int nseq, hipStream_t st) {
    return launch_fft<double, false>(logn, l, s, tw, nseq, st);
}
template <> int launch_row_nat<double>(int logn, const RowLoadNat<double>& l, const RowStoreNat<double>& s, const cx<double>* tw, int nseq, hipStream_t st) {
    return launch_fft<double, false>(logn, l, s, tw, nseq, st);
}
}  // namespace pm
