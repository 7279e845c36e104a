// Parameter blocks and launcher declarations of the real-input (Hermitian) 2-D transform -- the part capi.hip needs; the kernels are
// in fft_r2c.h.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_io.h"

namespace pm {

enum : int { EPI_ABS = 3, EPI_ARG = 4 };   // beyond EPI_NONE / EPI_ABS2 / EPI_ABS2_ACCUM of fft_io.h

template <typename T>
struct R2CRowStore {
    cx<T>* dst;         // tiled intermediate: N/2 + 1 columns (rounded up to whole layout tiles) x nseq rows
    int nseq;           // rows
    int log_tc;         // log2 of the layout tile width
    const cx<T>* twn;   // W_N^k, k < N (the table of the FULL row length)
    // fold (see RowStoreFold in fft_io.h): the thread's two rows are the pair (i, i + M/2); plane 0 row i = y[i] + y[i + M/2],
    // plane 1 row i = (y[i] - y[i + M/2]) W_M^i, nseq = M/2 rows per plane
    int fold;
    int64_t plane_stride;   // elements between the planes
    const cx<T>* twm;       // W_M^k
    int swap;               // input rows were rotated by M/2: slot 0 holds logical row i + M/2
};

template <typename T>
struct HermStore {
    void* dst;          // cx<T>* (EPI_NONE) or T*
    int64_t ld;
    AxisMap ay, ax;     // output views of the full M x N spectrum (rotation / crop)
    int M, N;
    int epilogue;       // EPI_NONE, EPI_ABS2, EPI_ABS2_ACCUM, EPI_ABS, EPI_ARG
    T scale, weight;
    int norm_dc;        // divide by F[0][0] first
    const cx<T>* w0;    // column 0 of the intermediate (layout tile 0), element q at w0[q * w0_stride]
    int64_t w0_stride;
    int nrows_w;        // rows stored in the intermediate
    int plane;          // folded column transform: -1 none, else the planes are told apart by blockIdx.y (0: even bins, 1: odd bins);
                        // M is then the length of ONE plane's transform, dst / ld / ay describe the plane's rows (ld doubled)
    int64_t plane_dst;  // elements (of the output type) from plane 0's first row to plane 1's
    int fast;           // unwindowed output, rotations by 0 or half a length, 16-byte alignable rows: the map-free store applies
    int wide;           // 2048-point tiles of 16 complex64 / 8 complex128 columns (ColCfgSel variant 2): 64 B pieces of a real output
};

template <typename T> int launch_row_r2c(int logn2, const RowLoadNat<T>&, const R2CRowStore<T>&, const cx<T>* tw, int nseq, int log_g, hipStream_t);
template <typename T> int launch_col_herm(int logm, const ColLoadTiled<T>&, const HermStore<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t);

}  // namespace pm
