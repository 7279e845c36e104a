// Transposed real-input (Hermitian) transform, double precision (explicit instantiation; see fft_hermt.h).
#include "fft_hermt.h"
namespace pm {
template <> int launch_col_hermt<double>(int logm, const ColLoadNat<double>& l, const HermTColStore<double>& s, const cx<double>* tw, int ntiles, int log_g, hipStream_t st) {
    return launch_col_hermt_impl<double>(logm, l, s, tw, ntiles, log_g, st);
}
template <> int launch_row_hermt<double>(int logn, int var, const RowLoadNat<double>& l, const HermTRowStore<double>& s, const cx<double>* tw, int log_g, hipStream_t st) {
    return launch_row_hermt_impl<double>(logn, var, l, s, tw, log_g, st);
}
}  // namespace pm
