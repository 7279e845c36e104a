// Spectral (multi-wavelength) transform pair, complex64 (explicit instantiation; see fft_spectral.h).
#include "fft_spectral.h"
namespace pm {

template <>
int launch_row_spectral<float>(int logn, int var, const RowLoadNat<float>& l, const RowStoreTiled<float>& s, const cx<float>* tw, int nseq,
                               int log_g, const Spectral& w, hipStream_t st) {
    using T = float;
    using S = RowStoreTiled<T>;
    switch (logn) {
#define PM_CASE(k) \
    case k:        \
        return launch_row_spectral_one<T, k, 0, S>(l, s, tw, nseq, log_g, w, st);
        PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10)
#undef PM_CASE
        case 11: return var == 5 ? launch_row_spectral_one<T, 11, 5, S>(l, s, tw, nseq, log_g, w, st)
                                 : launch_row_spectral_one<T, 11, 0, S>(l, s, tw, nseq, log_g, w, st);
        case 12: return var == 4 ? launch_row_spectral_one<T, 12, 4, S>(l, s, tw, nseq, log_g, w, st)
                                 : launch_row_spectral_one<T, 12, 0, S>(l, s, tw, nseq, log_g, w, st);
        default: return -2;     // 8192-point rows take the per-wavelength loop (capi.hip spectral_fast): this kernel spilled there
    }
}

template <>
int launch_row_spectral_fold<float>(int logn, const RowLoadNat<float>& l, const RowStoreFold<float>& s, const cx<float>* tw, int npairs,
                                    const Spectral& w, hipStream_t st) {
    using T = float;
    using S = RowStoreFold<T>;
    switch (logn) {
        case 11: return launch_row_spectral_one<T, 11, 4, S>(l, s, tw, npairs, 0, w, st);
        case 12: return launch_row_spectral_one<T, 12, 4, S>(l, s, tw, npairs, 0, w, st);
        default: return -2;
    }
}

template <>
int launch_col_spectral<float>(int logm, const ColLoadTiled<float>& l, const ColStoreNat<float>& s, const cx<float>* tw, int ntiles, int log_g,
                               const Spectral& w, hipStream_t st, int nplanes) {
    switch (logm) {
#define PM_CASE(k) \
    case k:        \
        return launch_col_spectral_one<float, k>(l, s, tw, ntiles, log_g, w, st, nplanes);
        PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASE(11)
#undef PM_CASE
        default: return -2;
    }
}

}  // namespace pm
