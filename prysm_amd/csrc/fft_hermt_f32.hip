// Transposed real-input (Hermitian) transform, float precision (explicit instantiation; see fft_hermt.h).
#include "fft_hermt.h"
namespace pm {
template <> int launch_col_hermt<float>(int logm, const ColLoadNat<float>& l, const HermTColStore<float>& s, const cx<float>* tw, int ntiles, int log_g, hipStream_t st) {
    return launch_col_hermt_impl<float>(logm, l, s, tw, ntiles, log_g, st);
}
template <> int launch_row_hermt<float>(int logn, int var, const RowLoadNat<float>& l, const HermTRowStore<float>& s, const cx<float>* tw, int log_g, hipStream_t st) {
    return launch_row_hermt_impl<float>(logn, var, l, s, tw, log_g, st);
}
}  // namespace pm
