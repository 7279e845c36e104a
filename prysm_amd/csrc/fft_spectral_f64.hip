// Spectral (multi-wavelength) transform pair, complex128 (explicit instantiation; see fft_spectral.h).  Rows of up to 2048 samples
// (longer complex128 rows exchange real and imaginary parts separately and fold differently); unfolded column tiles up to 2048 points.
#include "fft_spectral.h"
namespace pm {

template <>
int launch_row_spectral<double>(int logn, int var, const RowLoadNat<double>& l, const RowStoreTiled<double>& s, const cx<double>* tw, int nseq,
                                int log_g, const Spectral& w, hipStream_t st) {
    using T = double;
    using S = RowStoreTiled<T>;
    switch (logn) {
#define PM_CASE(k) \
    case k:        \
        return launch_row_spectral_one<T, k, 0, S>(l, s, tw, nseq, log_g, w, st);
        PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10)
#undef PM_CASE
        case 11: return var == 1 ? launch_row_spectral_one<T, 11, 1, S>(l, s, tw, nseq, log_g, w, st)
                                 : launch_row_spectral_one<T, 11, 0, S>(l, s, tw, nseq, log_g, w, st);
        default: return -2;
    }
}

template <>
int launch_row_spectral_fold<double>(int, const RowLoadNat<double>&, const RowStoreFold<double>&, const cx<double>*, int, const Spectral&,
                                     hipStream_t) {
    return -2;
}

template <>
int launch_col_spectral<double>(int logm, const ColLoadTiled<double>& l, const ColStoreNat<double>& s, const cx<double>* tw, int ntiles,
                                int log_g, const Spectral& w, hipStream_t st, int nplanes) {
    switch (logm) {
#define PM_CASE(k) \
    case k:        \
        return launch_col_spectral_one<double, k>(l, s, tw, ntiles, log_g, w, st, nplanes);
        PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASE(11)
#undef PM_CASE
        default: return -2;
    }
}

}  // namespace pm
