// Parameter block and entry points of the spectral (multi-wavelength) transform pair, fft_spectral.h.
#pragma once
#include "fft_io.h"

namespace pm {

constexpr int kSpectralMax = 8;   // wavelengths per launch pair

struct Spectral {
    int nb;                     // wavelengths of this launch pair (<= kSpectralMax)
    double w[kSpectralMax];     // spectral weights
    double k2[kSpectralMax];    // phase in turns per OPD unit: k / (2 pi), per wavelength
    int64_t fstride;            // elements between the intermediates of consecutive wavelengths
    int mode;                   // bit 0: rows keep the packed map in registers; bit 1: columns accumulate in registers (fft_spectral.h)
};

// rows: VAR as launch_row_tiled; the store's bstride must be sp.fstride
template <typename T>
int launch_row_spectral(int logn, int var, const RowLoadNat<T>&, const RowStoreTiled<T>&, const cx<T>* tw, int nseq, int log_g, const Spectral&,
                        hipStream_t);
template <typename T>
int launch_row_spectral_fold(int logn, const RowLoadNat<T>&, const RowStoreFold<T>&, const cx<T>* tw, int npairs, const Spectral&, hipStream_t);
// columns: nplanes = 2 for the planes of a folded transform (blockIdx.y, the load's / store's own bstride), else 1
template <typename T>
int launch_col_spectral(int logm, const ColLoadTiled<T>&, const ColStoreNat<T>&, const cx<T>* tw, int ntiles, int log_g, const Spectral&,
                        hipStream_t, int nplanes);

}  // namespace pm
