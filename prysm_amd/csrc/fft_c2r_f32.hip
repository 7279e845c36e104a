// Real-in / real-out spectral multiply, float precision (explicit instantiation; see fft_c2r.h).
#include "fft_c2r.h"
namespace pm {
template <> int launch_col_mul_herm<float>(int logm, const ColLoadTiled<float>& l, const HermMul<float>& h, const ColStoreTiled<float>& s, const cx<float>* tw, int ntiles, int log_g, hipStream_t st) {
    return launch_col_mul_herm_impl<float>(logm, l, h, s, tw, ntiles, log_g, st);
}
template <> int launch_row_c2r<float>(int logn2, const RowLoadTiled<float>& l, const RowStoreNat<float>& s, const cx<float>* tw2, const cx<float>* twn, int nseq, hipStream_t st) {
    return launch_row_c2r_impl<float>(logn2, l, s, tw2, twn, nseq, st);
}
template <> int launch_row_c2r_fold<float>(int logn2, const RowLoadFold<float>& l, const RowStoreNat<float>& s, const cx<float>* tw2, const cx<float>* twn, int npairs, hipStream_t st) {
    return launch_row_c2r_fold_impl<float>(logn2, l, s, tw2, twn, npairs, st);
}
}  // namespace pm
