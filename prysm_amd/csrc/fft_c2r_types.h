// Parameter blocks and launcher declarations of the real-in / real-out spectral multiply (fft_c2r.h) -- the part capi.hip needs.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_io.h"

namespace pm {

// full multiplier H (M x N, indexed by the unshifted bin) as the half spectrum sees it: the result is the REAL part of
// ifft2(X H), i.e. ifft2(X Hh) with Hh(u, k) = (H(u, k) + conj H(-u, -k)) / 2 -- for the transfer function of a real PSF Hh = H
template <typename T>
struct HermMul {
    const cx<T>* H;
    int64_t ld;
    int M, N;
    int conj;       // multiply by conj(H)
    int planes;     // folded column transform: blockIdx.y = plane b holds the bins u = 2 u' + b of the length-M transform (M stays the FULL length)
};

template <typename T> int launch_col_mul_herm(int logm, const ColLoadTiled<T>&, const HermMul<T>&, const ColStoreTiled<T>&, const cx<T>* tw,
                                              int ntiles, int log_g, hipStream_t);
// twn: W_N^k of the FULL row length
template <typename T> int launch_row_c2r(int logn2, const RowLoadTiled<T>&, const RowStoreNat<T>&, const cx<T>* tw2, const cx<T>* twn, int nseq,
                                         hipStream_t);
// folded form: the load rebuilds the row pair (n, n + M/2) from the two planes (RowLoadFold); the store's eoff must be M/2
template <typename T> int launch_row_c2r_fold(int logn2, const RowLoadFold<T>&, const RowStoreNat<T>&, const cx<T>* tw2, const cx<T>* twn, int npairs,
                                              hipStream_t);

}  // namespace pm
