// Real-input (Hermitian) 2-D transform kernels, float precision (explicit instantiation; see fft_r2c.h).
#include "fft_r2c.h"
namespace pm {
template <> int launch_row_r2c<float>(int logn2, const RowLoadNat<float>& l, const R2CRowStore<float>& s, const cx<float>* tw, int nseq, int log_g, hipStream_t st) {
    return launch_row_r2c_impl<float>(logn2, l, s, tw, nseq, log_g, st);
}
template <> int launch_col_herm<float>(int logm, const ColLoadTiled<float>& l, const HermStore<float>& s, const cx<float>* tw, int ntiles, int log_g, hipStream_t st) {
    return launch_col_herm_impl<float>(logm, l, s, tw, ntiles, log_g, st);
}
}  // namespace pm
