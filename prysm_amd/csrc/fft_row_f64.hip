// Row-pass FFT kernels, double precision (explicit instantiation; see fft_kernels.h).
#include "fft_kernels.h"
namespace pm {
template <> int launch_row_tiled<double>(int logn, int var, const RowLoadNat<double>& l, const RowStoreTiled<double>& s, const cx<double>* tw, int nseq, int log_g, hipStream_t st, int nbatch) {
    return launch_fft<double, false>(logn, var, l, s, tw, nseq, log_g, st, nbatch);
}
template <> int launch_row_nat<double>(int logn, int var, const RowLoadNat<double>& l, const RowStoreNat<double>& s, const cx<double>* tw, int nseq, int log_g, hipStream_t st, int nbatch) {
    return launch_fft<double, false>(logn, var, l, s, tw, nseq, log_g, st, nbatch);
}
template <> int launch_row_from_tiled<double>(int logn, int var, const RowLoadTiled<double>& l, const RowStoreNat<double>& s, const cx<double>* tw, int nseq, hipStream_t st, int nbatch) {
    return launch_fft<double, false>(logn, var, l, s, tw, nseq, 0, st, nbatch);
}
template <> int launch_row_fold<double>(int logn, const RowLoadNat<double>& l, const RowStoreFold<double>& s, const cx<double>* tw, int npairs, int log_g, hipStream_t st, int nbatch) {
    return launch_fold_impl<double>(logn, l, s, tw, npairs, log_g, st, nbatch);
}
template <> int launch_row_unfold<double>(int logn, const RowLoadFold<double>& l, const RowStoreNat<double>& s, const cx<double>* tw, int npairs, hipStream_t st, int nbatch) {
    return launch_unfold_impl<double>(logn, l, s, tw, npairs, st, nbatch);
}
template <> int launch_row_chirp_tiled<double>(int logn, int var, const RowLoadChirp<double>& l, const RowStoreTiled<double>& s, const cx<double>* tw, int nseq, int log_g, hipStream_t st) {
    return launch_fft<double, false>(logn, var, l, s, tw, nseq, log_g, st, 1);
}
template <> int launch_row_tiled_chirp<double>(int logn, int var, const RowLoadTiled<double>& l, const RowStoreChirp<double>& s, const cx<double>* tw, int nseq, hipStream_t st) {
    return launch_fft<double, false>(logn, var, l, s, tw, nseq, 0, st, 1);
}
}  // namespace pm
