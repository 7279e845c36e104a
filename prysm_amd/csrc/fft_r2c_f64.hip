// Real-input (Hermitian) 2-D transform kernels, double precision (explicit instantiation; see fft_r2c.h).
#include "fft_r2c.h"
namespace pm {
template <> int launch_row_r2c<double>(int logn2, const RowLoadNat<double>& l, const R2CRowStore<double>& s, const cx<double>* tw, int nseq, int log_g, hipStream_t st) {
    return launch_row_r2c_impl<double>(logn2, l, s, tw, nseq, log_g, st);
}
template <> int launch_col_herm<double>(int logm, const ColLoadTiled<double>& l, const HermStore<double>& s, const cx<double>* tw, int ntiles, int log_g, hipStream_t st) {
    return launch_col_herm_impl<double>(logm, l, s, tw, ntiles, log_g, st);
}
}  // namespace pm
