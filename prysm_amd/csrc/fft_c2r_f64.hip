// Real-in / real-out spectral multiply, double precision (explicit instantiation; see fft_c2r.h).
#include "fft_c2r.h"
namespace pm {
template <> int launch_col_mul_herm<double>(int logm, const ColLoadTiled<double>& l, const HermMul<double>& h, const ColStoreTiled<double>& s, const cx<double>* tw, int ntiles, int log_g, hipStream_t st) {
    return launch_col_mul_herm_impl<double>(logm, l, h, s, tw, ntiles, log_g, st);
}
template <> int launch_row_c2r<double>(int logn2, const RowLoadTiled<double>& l, const RowStoreNat<double>& s, const cx<double>* tw2, const cx<double>* twn, int nseq, hipStream_t st) {
    return launch_row_c2r_impl<double>(logn2, l, s, tw2, twn, nseq, st);
}
template <> int launch_row_c2r_fold<double>(int logn2, const RowLoadFold<double>& l, const RowStoreNat<double>& s, const cx<double>* tw2, const cx<double>* twn, int npairs, hipStream_t st) {
    return launch_row_c2r_fold_impl<double>(logn2, l, s, tw2, twn, npairs, st);
}
}  // namespace pm
