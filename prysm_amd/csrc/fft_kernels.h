// HIP kernels wrapping the FFT engine + host launchers (explicitly instantiated
// per precision / mode in fft_row_f32.hip, fft_row_f64.hip, fft_col_f32.hip,
// fft_col_f64.hip so the four translation units compile in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "fft_io.h"

namespace pm {

constexpr int kMaxLogN = 13;  // 8192: the largest length one workgroup holds on chip

// ---- per-length configurations -------------------------------------------------------------
// Row pass: one sequence per N/16 threads, 256 threads per workgroup (512 at N = 8192); LDS is
// 8.5 B per point (complex64) so 4 workgroups / CU stay resident at N = 4096.
// VAR (row_variant() in pm_internal.h picks it per length / precision):
//   1: real and imaginary parts exchanged separately -> half the LDS per workgroup (complex128 rows of 2048 points)
//   4: two rows per thread -- stage twiddles, their products and the LDS addressing are shared by the pair (complex64 from
//      4096 points; the folded row pass pairs rows (i, i + M/2) this way)
//   5: one sequence per (small) workgroup -- twice the workgroups at 2048 points
template <typename T, int LOGN, int VAR>
struct RowCfgSel {
    static constexpr int N = 1 << LOGN, P = N >= 16 ? 16 : N, TPS = N / P;
    static constexpr int BO = (TPS >= 256 || VAR == 5) ? 1 : 256 / TPS;
    static constexpr int COMP = ((sizeof(T) == 8 && LOGN >= 12) || VAR == 1) ? 2 : 1;
    static constexpr int E = (VAR == 4) ? 2 : 1;
    using type = FftCfg<T, LOGN, 1, E, BO, COMP>;
};
// Column pass: a tile of 64 B rows (8 complex64 / 4 complex128 columns) per workgroup; at
// M = 4096 that is 256 KiB of field in the registers of 1024 threads, exchanged through LDS in
// two 136 KiB chunks (complex64: the thread's two columns; complex128: real then imaginary).
// VAR = 2: tiles of 128 B rows (8 complex128 columns, 1024 threads) for 2048-point columns -- the planes of a folded 4096-row
// complex128 transform store whole cache lines (216 -> 209 us at 4096^2, profiles/r02/exp_wide_col_tiles.log; complex64 loses
// the two-workgroups-per-CU overlap instead: 96 -> 108 us, so it keeps the 64 B tiles).
template <typename T, int LOGN, int VAR>
struct ColCfgSel {
    static constexpr int N = 1 << LOGN, P = N >= 16 ? 16 : N, TPS = N / P;
    static constexpr int E = sizeof(T) == 4 ? 2 : 1;
    static constexpr int CI = (VAR == 2 && LOGN == 11) ? 8 : (LOGN <= 12 ? 4 : 2);
    static constexpr int BO = (CI * TPS >= 256) ? 1 : 256 / (CI * TPS);
    static constexpr int COMP = (sizeof(T) == 8 && LOGN >= 11) ? 2 : 1;
    using type = FftCfg<T, LOGN, CI, E, BO, COMP>;
};

template <typename C, bool COL, int VAR, typename L, typename S>
__global__ void __launch_bounds__(C::NT) fft_kernel(const L lp, const S sp, const cx<typename C::T>* __restrict__ tw, const int log_g) {
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    int unit = group_remap(blockIdx.x, gridDim.x, log_g);
    if (COL) unit = unit * C::BO + pos.bo;
    cx<typename C::T> v[C::E][C::P];
    const L lpb = at_batch(lp, blockIdx.y);   // blockIdx.y: field of a batch
    const S spb = at_batch(sp, blockIdx.y);
    load<C>(lpb, unit, pos, v);
    // two sequences per thread (complex64 columns, paired rows): exchanges pipelined against the butterflies of the other
    // sequence (measured: 4096^2 column pass 57.2 -> 56.1 us, 8192^2 419 -> 358 us); row variant 5 keeps the plain order
    if constexpr (VAR != 5 && C::E == 2 && C::COMP == 1 && C::NSTAGE > 1)
        fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else
        fft_run<C>(v, pos, pm_smem, tw);
    store<C>(spb, unit, pos, v);
}

// Fused spectral-multiply column pass: forward transform, multiply by H, inverse transform -- all on the
// registers of the workgroup (the engine returns natural order, so the inverse starts where the forward ended);
// reads the tiled intermediate of the row pass and writes a tiled buffer for the inverse row pass.  This is the
// middle pass of ifft2(fft2(x) * H) in 3 passes / 6 N^2 s bytes instead of 4 passes / 8 N^2 s.
// Measured (profiles/r01/fused_as.log): 4096^2 complex128 448 vs 473 us, complex64 210 vs 239 us, 2048^2 complex64
// 55 vs 75 us against two pm_fft2 calls.  (Built without the SLP vectorizer -- with it this kernel spilled > 150
// VGPRs under the 128-register cap of the 1024-thread workgroup and lost.)
#ifndef PM_COLMUL_MINWG
#define PM_COLMUL_MINWG 1
#endif
template <typename C, typename S = ColStoreTiled<typename C::T>>
__global__ void __launch_bounds__(C::NT, (C::NT == 512 ? PM_COLMUL_MINWG : 1))   // 2nd argument: min waves per SIMD
    fft_col_mul_kernel(const ColLoadTiled<typename C::T> lp0, const MidMul<typename C::T> mp0,
                                                            const S sp0,
                                                            const cx<typename C::T>* __restrict__ tw, const int log_g) {
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = group_remap(blockIdx.x, gridDim.x, log_g) * C::BO + pos.bo;
    const auto lp = at_batch(lp0, blockIdx.y);
    const auto mp = at_batch(mp0, blockIdx.y);
    const auto sp = at_batch(sp0, blockIdx.y);
    cx<typename C::T> v[C::E][C::P];
    load<C>(lp, unit, pos, v);
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else fft_run<C>(v, pos, pm_smem, tw);
    mid_multiply_conj<C>(mp, unit, pos, v);
    __syncthreads();   // LDS of the forward exchange is reused by the inverse
    // opaque copy of the slot: otherwise the twiddles (and their products) of the first transform are CSE'd
    // with the second and kept live across it -- hundreds of spilled registers at 1024 threads
    ThreadPos pos2 = pos;
    asm volatile("" : "+v"(pos2.t), "+v"(pos2.cl), "+v"(pos2.bo));
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos2, pm_smem, tw);
    else fft_run<C>(v, pos2, pm_smem, tw);
#pragma unroll
    for (int e = 0; e < C::E; ++e)
#pragma unroll
        for (int m = 0; m < C::P; ++m) v[e][m].y = -v[e][m].y;
    store<C>(sp, unit, pos, v);
}

template <typename T, int LOGN, typename S>
int launch_col_mul_one(const ColLoadTiled<T>& lp, const MidMul<T>& mp, const S& sp, const cx<T>* tw, int ntiles,
                       int log_g, hipStream_t st, int nbatch) {
    using C = typename ColCfgSel<T, LOGN, 0>::type;
    auto kern = fft_col_mul_kernel<C, S>;
    if (C::LDS_BYTES > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           int(C::LDS_BYTES));
        if (e != hipSuccess) return int(e);
    }
    const int grid = (ntiles + C::BO - 1) / C::BO;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid, nbatch), dim3(C::NT), C::LDS_BYTES, st, lp, mp, sp, tw, log_g);
    return int(hipGetLastError());
}

template <typename T, typename S>
int launch_col_mul_impl(int logm, const ColLoadTiled<T>& lp, const MidMul<T>& mp, const S& sp, const cx<T>* tw,
                        int ntiles, int log_g, hipStream_t st, int nbatch) {
    switch (logm) {
#define PM_CASE(k) \
    case k:        \
        return launch_col_mul_one<T, k, S>(lp, mp, sp, tw, ntiles, log_g, st, nbatch);
        PM_CASE(1) PM_CASE(2) PM_CASE(3) PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7)
        PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASE(11) PM_CASE(12) PM_CASE(13)
#undef PM_CASE
        default:
            return -2;
    }
}

template <typename T, bool COL, int LOGN, int VAR, typename L, typename S>
int launch_one(const L& lp, const S& sp, const cx<T>* tw, int units, int log_g, hipStream_t st, int nbatch) {
    using Sel = typename std::conditional<COL, ColCfgSel<T, LOGN, VAR>, RowCfgSel<T, LOGN, VAR>>::type;
    using C = typename Sel::type;
    auto kern = fft_kernel<C, COL, VAR, L, S>;
    constexpr size_t LDSB = C::LDS_BYTES;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int per_wg = C::BO * (COL ? 1 : C::E);     // row mode: a thread owns E consecutive rows
    const int grid = (units + per_wg - 1) / per_wg;
    if (grid <= 0 || nbatch <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid, nbatch), dim3(C::NT), LDSB, st, lp, sp, tw, log_g);
    return int(hipGetLastError());
}


// folded row pass (RowStoreFold): units are row PAIRS (i, i + M/2); complex64 / complex128 rows of 2048 .. 8192 points
template <typename T, int LOGN>
int launch_fold_one(const RowLoadNat<T>& lp, const RowStoreFold<T>& sp, const cx<T>* tw, int npairs, int log_g, hipStream_t st,
                    int nbatch) {
    using C = typename RowCfgSel<T, LOGN, 4>::type;
    auto kern = fft_kernel<C, false, 4, RowLoadNat<T>, RowStoreFold<T>>;
    constexpr size_t LDSB = C::LDS_BYTES;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int grid = (npairs + C::BO - 1) / C::BO;
    if (grid <= 0 || nbatch <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid, nbatch), dim3(C::NT), LDSB, st, lp, sp, tw, log_g);
    return int(hipGetLastError());
}
template <typename T>
int launch_fold_impl(int logn, const RowLoadNat<T>& lp, const RowStoreFold<T>& sp, const cx<T>* tw, int npairs, int log_g,
                     hipStream_t st, int nbatch) {
    switch (logn) {
        case 11: return launch_fold_one<T, 11>(lp, sp, tw, npairs, log_g, st, nbatch);
        case 12: return launch_fold_one<T, 12>(lp, sp, tw, npairs, log_g, st, nbatch);
        case 13: return launch_fold_one<T, 13>(lp, sp, tw, npairs, log_g, st, nbatch);
        default: return -2;
    }
}

// last pass of the folded fused operation: rows rebuilt from the two planes (RowLoadFold), natural output
template <typename T, int LOGN>
int launch_unfold_one(const RowLoadFold<T>& lp, const RowStoreNat<T>& sp, const cx<T>* tw, int npairs, hipStream_t st, int nbatch) {
    using C = typename RowCfgSel<T, LOGN, 4>::type;
    auto kern = fft_kernel<C, false, 4, RowLoadFold<T>, RowStoreNat<T>>;
    constexpr size_t LDSB = C::LDS_BYTES;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int grid = (npairs + C::BO - 1) / C::BO;
    if (grid <= 0 || nbatch <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid, nbatch), dim3(C::NT), LDSB, st, lp, sp, tw, 0);
    return int(hipGetLastError());
}
template <typename T>
int launch_unfold_impl(int logn, const RowLoadFold<T>& lp, const RowStoreNat<T>& sp, const cx<T>* tw, int npairs, hipStream_t st,
                       int nbatch) {
    switch (logn) {
        case 11: return launch_unfold_one<T, 11>(lp, sp, tw, npairs, st, nbatch);
        case 12: return launch_unfold_one<T, 12>(lp, sp, tw, npairs, st, nbatch);
        case 13: return launch_unfold_one<T, 13>(lp, sp, tw, npairs, st, nbatch);
        default: return -2;
    }
}

template <typename T, bool COL, typename L, typename S>
int launch_fft(int logn, int var, const L& lp, const S& sp, const cx<T>* tw, int units, int log_g, hipStream_t st, int nbatch) {
    switch (logn) {
#define PM_CASE(k) \
    case k:        \
        return launch_one<T, COL, k, 0, L, S>(lp, sp, tw, units, log_g, st, nbatch);
// lengths with more than one tiling (see RowCfgSel / ColCfgSel)
#define PM_CASEV(k)                                                                                              \
    case k:                                                                                                      \
        if (!COL && var == 1) return launch_one<T, COL, k, (COL ? 0 : 1), L, S>(lp, sp, tw, units, log_g, st, nbatch); \
        if (!COL && var == 4) return launch_one<T, COL, k, (COL ? 0 : 4), L, S>(lp, sp, tw, units, log_g, st, nbatch); \
        if (!COL && var == 5) return launch_one<T, COL, k, (COL ? 0 : 5), L, S>(lp, sp, tw, units, log_g, st, nbatch); \
        if (COL && var == 2 && k == 11) return launch_one<T, COL, k, (COL ? 2 : 0), L, S>(lp, sp, tw, units, log_g, st, nbatch); \
        return launch_one<T, COL, k, 0, L, S>(lp, sp, tw, units, log_g, st, nbatch);
        PM_CASE(1) PM_CASE(2) PM_CASE(3) PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7)
        PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASEV(11) PM_CASEV(12) PM_CASEV(13)
#undef PM_CASE
#undef PM_CASEV
        default:
            return -2;
    }
}

// entry points, one explicit instantiation per .hip file
template <typename T> int launch_row_tiled(int logn, int var, const RowLoadNat<T>&, const RowStoreTiled<T>&, const cx<T>* tw, int nseq, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_row_nat(int logn, int var, const RowLoadNat<T>&, const RowStoreNat<T>&, const cx<T>* tw, int nseq, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_col_tiled(int logm, int var, const ColLoadTiled<T>&, const ColStoreNat<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_col_nat(int logm, int var, const ColLoadNat<T>&, const ColStoreNat<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_col_mul(int logm, const ColLoadTiled<T>&, const MidMul<T>&, const ColStoreTiled<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_col_mul_crop(int logm, const ColLoadTiled<T>&, const MidMul<T>&, const ColStoreTiledCrop<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t);
template <typename T> int launch_row_from_tiled(int logn, int var, const RowLoadTiled<T>&, const RowStoreNat<T>&, const cx<T>* tw, int nseq, hipStream_t, int nbatch = 1);
template <typename T> int launch_row_fold(int logn, const RowLoadNat<T>&, const RowStoreFold<T>&, const cx<T>* tw, int npairs, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_row_unfold(int logn, const RowLoadFold<T>&, const RowStoreNat<T>&, const cx<T>* tw, int npairs, hipStream_t, int nbatch = 1);
template <typename T> int launch_row_chirp_tiled(int logn, int var, const RowLoadChirp<T>&, const RowStoreTiled<T>&, const cx<T>* tw, int nseq, int log_g, hipStream_t);
template <typename T> int launch_row_tiled_chirp(int logn, int var, const RowLoadTiled<T>&, const RowStoreChirp<T>&, const cx<T>* tw, int nseq, hipStream_t);

}  // namespace pm
