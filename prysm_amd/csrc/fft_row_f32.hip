// Row-pass FFT kernels, float precision (explicit instantiation; see fft_kernels.h).
#include "fft_kernels.h"
namespace pm {
template <> int launch_row_tiled<float>(int logn, int var, const RowLoadNat<float>& l, const RowStoreTiled<float>& s, const cx<float>* tw, int nseq, int log_g, hipStream_t st, int nbatch) {
    return launch_fft<float, false>(logn, var, l, s, tw, nseq, log_g, st, nbatch);
}
template <> int launch_row_nat<float>(int logn, int var, const RowLoadNat<float>& l, const RowStoreNat<float>& s, const cx<float>* tw, int nseq, int log_g, hipStream_t st, int nbatch) {
    return launch_fft<float, false>(logn, var, l, s, tw, nseq, log_g, st, nbatch);
}
template <> int launch_row_from_tiled<float>(int logn, int var, const RowLoadTiled<float>& l, const RowStoreNat<float>& s, const cx<float>* tw, int nseq, hipStream_t st, int nbatch) {
    return launch_fft<float, false>(logn, var, l, s, tw, nseq, 0, st, nbatch);
}
template <> int launch_row_fold<float>(int logn, const RowLoadNat<float>& l, const RowStoreFold<float>& s, const cx<float>* tw, int npairs, int log_g, hipStream_t st, int nbatch) {
    return launch_fold_impl<float>(logn, l, s, tw, npairs, log_g, st, nbatch);
}
template <> int launch_row_unfold<float>(int logn, const RowLoadFold<float>& l, const RowStoreNat<float>& s, const cx<float>* tw, int npairs, hipStream_t st, int nbatch) {
    return launch_unfold_impl<float>(logn, l, s, tw, npairs, st, nbatch);
}
template <> int launch_row_chirp_tiled<float>(int logn, int var, const RowLoadChirp<float>& l, const RowStoreTiled<float>& s, const cx<float>* tw, int nseq, int log_g, hipStream_t st) {
    return launch_fft<float, false>(logn, var, l, s, tw, nseq, log_g, st, 1);
}
template <> int launch_row_tiled_chirp<float>(int logn, int var, const RowLoadTiled<float>& l, const RowStoreChirp<float>& s, const cx<float>* tw, int nseq, hipStream_t st) {
    return launch_fft<float, false>(logn, var, l, s, tw, nseq, 0, st, 1);
}
}  // namespace pm
