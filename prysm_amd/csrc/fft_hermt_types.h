// Parameter blocks and launcher declarations of the TRANSPOSED real-input (Hermitian) 2-D transform -- the part capi.hip needs; the
// kernels are in fft_hermt.h.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_io.h"
#include "fft_r2c_types.h"

namespace pm {

// pass A (column transforms of the real array read as M x N/2 complex): the separated column spectra X_j(u), u < M/2, of the N real
// columns go out as rows u of a natural (row-major) M/2 x N complex intermediate; row 0 carries X_j(0) + i X_j(M/2) (both real)
template <typename T>
struct HermTColStore {
    cx<T>* dst;
    int64_t ld;         // complex elements between rows of the intermediate
    int ncols2;         // N / 2 packed columns
    int neg_odd;        // the input rows came rotated by M/2 (ifftshift): bins of odd u change sign
    int fold;           // the radix-2 step of the column transform is taken in the load: workgroups (tile, plane), M/2-point transforms
    const cx<T>* twm;   // ... W_M^k
    int ntiles;
};

// pass B (N-point row transforms of rows u < M/2): row u is stored twice -- F[u][k] at ((u + sy) mod M, (k + sx) mod N) and its
// conjugate image at ((M - u + sy) mod M, (N - k + sx) mod N) --, row 0 separates into F[0][.] and F[M/2][.]
template <typename T>
struct HermTRowStore {
    void* dst;          // cx<T>* (EPI_NONE) or T*
    int64_t ld;         // output elements between rows
    int M, N;
    int sy, sx;         // output rotations (0 or half a length)
    int epilogue;       // EPI_NONE, EPI_ABS2, EPI_ABS, EPI_ARG
    T scale;
    int norm_dc;        // divide by F[0][0] first
    const cx<T>* i0;    // row 0 of the intermediate: re = X_j(0), whose sum over j is F[0][0]
    int neg_odd;        // the input columns came rotated by N/2: bins of odd k change sign
    int nseq;           // M / 2 rows
};

template <typename T> int launch_col_hermt(int logm, const ColLoadNat<T>&, const HermTColStore<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t);
template <typename T> int launch_row_hermt(int logn, int var, const RowLoadNat<T>&, const HermTRowStore<T>&, const cx<T>* tw, int log_g, hipStream_t);

}  // namespace pm
