// out = Re( ifft2( fft2(x) H ) ) for a REAL x, on half spectra end to end: the image-chain convolution of the reference
// (prysm/convolution.py:9-31 conv, :34-113 apply_transfer_functions take `.real` of the complex result when the object is real).
//   pass 1  fft_row_r2c_kernel (fft_r2c.h): the real rows as N/2 packed complex points -> N/2 columns of the tiled intermediate,
//           column 0 = X[0] + i X[N/2];
//   pass 2  fft_col_mul_herm_kernel: column transform, x Hh(u, k) = (H(u, k) + conj H(-u, -k)) / 2 (the Hermitian part of H: what
//           taking the real part of the result amounts to), inverse column transform -- in the registers of the workgroup, like the
//           complex chain's middle pass; the packed column is separated into its two real columns' transforms (partner exchange
//           through LDS), multiplied by Hh(u, 0) and Hh(u, N/2), and packed again;
//   pass 3  fft_row_c2r_kernel: each row's half spectrum Z[0 .. N/2] -> the packed spectrum E[k] + i O[k] of its even / odd samples
//           (E = (Z[k] + conj Z[N/2 - k]), O = (Z[k] - conj Z[N/2 - k]) conj W_N^k: the inverse of r2c_combine), N/2-point inverse
//           transform, stored as N real samples.
// Bytes per sample: 4 + 4, 8 (+ 8 of H), 4 + 4 = 32 against 8 + 8, 16 (+ 8), 8 + 8 = 56 for the complex chain on a real array.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_c2r_types.h"
#include "fft_r2c.h"

namespace pm {

template <typename T>
PM_HD cx<T> herm_part(const HermMul<T>& h, int u, int k) {      // 2 Hh(u, k); the caller halves
    const int um = (h.M - u) & (h.M - 1), km = (h.N - k) & (h.N - 1);
    const cx<T> a = h.H[int64_t(u) * h.ld + k], b = h.H[int64_t(um) * h.ld + km];
    const cx<T> s = {a.x + b.x, a.y - b.y};
    return h.conj ? cx<T>{s.x, -s.y} : s;
}

template <typename C>
__global__ void __launch_bounds__(C::NT) fft_col_mul_herm_kernel(const ColLoadTiled<typename C::T> lp0, const HermMul<typename C::T> hm,
                                                                const ColStoreTiled<typename C::T> sp0,
                                                                const cx<typename C::T>* __restrict__ tw, const int log_g) {
    using T = typename C::T;
    constexpr int TC = C::CI * C::E;
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = group_remap(blockIdx.x, gridDim.x, log_g) * C::BO + pos.bo;
    const ColLoadTiled<T> lp = at_batch(lp0, blockIdx.y);       // blockIdx.y: plane of a folded transform
    const ColStoreTiled<T> sp = at_batch(sp0, blockIdx.y);
    const int pb = hm.planes ? int(blockIdx.y) : 0;             // this plane's bins are u = 2 u' + pb (u = u' unfolded)
    const int ush = hm.planes ? 1 : 0;
    cx<T> v[C::E][C::P];
    load<C>(lp, unit, pos, v);
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else fft_run<C>(v, pos, pm_smem, tw);
    const int n2 = hm.N / 2;
    const int col0 = unit * TC + pos.cl * C::E;
    // the packed column: its transform P separates as X0[u] = (P[u] + conj P[-u]) / 2, XN[u] = (P[u] - conj P[-u]) / (2i); the
    // partner P[-u] comes through LDS in the workgroup that owns tile 0 (short columns: every workgroup passes the barriers)
    // (read from LDS where it is used, slot by slot: held in a register array, the 16 partners were 32 VGPRs beside the 64 of the
    // field through the whole multiply -- 83 spilled registers under the 128-register cap of the 1024-thread tiles, and a spill
    // reload waits for every outstanding load: 86 us = 0.39 of the HBM roofline for the middle pass of a real 4096^2 convolution)
    cx<T>* const ex = reinterpret_cast<cx<T>*>(pm_smem);
    if (C::BO > 1 || unit == 0) {
        __syncthreads();
        if (col0 == 0) {
#pragma unroll
            for (int m = 0; m < C::P; ++m) ex[pos.bo * C::N + pos.t + m * C::TPS] = v[0][m];
        }
        __syncthreads();
    }
    const T half = T(0.5);
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int u = ((pos.t + m * C::TPS) << ush) + pb;      // bin of the full length-M transform
#pragma unroll
        for (int e = 0; e < C::E; ++e) {
            const int k = col0 + e;
            cx<T> y = {T(0), T(0)};
            if (k == 0) {
                const int up = pos.t + m * C::TPS;      // the partner -u: (L - u') mod L in the plane of the even bins, L - 1 - u' among the odd ones
                const cx<T> p = v[e][m], q = ex[pos.bo * C::N + (pb ? C::N - 1 - up : ((C::N - up) & (C::N - 1)))];
                const cx<T> x0 = {half * (p.x + q.x), half * (p.y - q.y)};                 // (P + conj Pm) / 2
                const cx<T> xn = mul_mi(cx<T>{half * (p.x - q.x), half * (p.y + q.y)});    // (P - conj Pm) / (2 i)
                const cx<T> y0 = cmul(x0, cscale(herm_part(hm, u, 0), half));
                const cx<T> yn = cmul(xn, cscale(herm_part(hm, u, n2), half));
                y = y0 + mul_pi(yn);
            } else if (k < n2) {
                y = cmul(v[e][m], cscale(herm_part(hm, u, k), half));
            }
            v[e][m] = {y.x, -y.y};      // conj: the inverse transform is the forward one between two conjugations
        }
        __builtin_amdgcn_sched_barrier(0);      // one slot at a time: hoisted, the 4 x 16 multiplier loads and their 64-bit addresses spill
    }
    __syncthreads();     // LDS of the forward exchange (and of the partner exchange) is reused by the inverse
    ThreadPos pos2 = pos;   // opaque copy: see fft_col_mul_kernel
    asm volatile("" : "+v"(pos2.t), "+v"(pos2.cl), "+v"(pos2.bo));
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos2, pm_smem, tw);
    else fft_run<C>(v, pos2, pm_smem, tw);
#pragma unroll
    for (int e = 0; e < C::E; ++e)
#pragma unroll
        for (int m = 0; m < C::P; ++m) v[e][m].y = -v[e][m].y;
    store<C>(sp, unit, pos, v);
}

template <typename C, typename L>
__global__ void __launch_bounds__(C::NT) fft_row_c2r_kernel(const L lp, const RowStoreNat<typename C::T> sp,
                                                           const cx<typename C::T>* __restrict__ tw, const cx<typename C::T>* __restrict__ twn) {
    using T = typename C::T;
    static_assert(C::COMP == 1 && C::CI == 1 && C::P == 16, "row mode, complex exchange, rows of at least 32 samples");
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(pm_smem);
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = blockIdx.x;
    constexpr int N2 = C::N;
    cx<T> v[C::E][C::P];
    const cx<T> wt = twn[pos.t];
    load<C>(lp, unit, pos, v);
    // partner exchange, one slot e at a time through the same LDS region (see fft_row_r2c_kernel): the sequences of the workgroup in
    // natural order, then Z[(N2 - k) mod N2]
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        if (e > 0) __syncthreads();
#pragma unroll
        for (int m = 0; m < C::P; ++m) lds[lds_addr<C>(pos.bo, 0, pos.t + m * C::TPS)] = v[e][m];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const int k = pos.t + m * C::TPS;
            const cx<T> z = v[e][m];
            cx<T> zp;
            if (k == 0) {
                zp = {z.x + z.y, z.x - z.y};       // X[0] and X[N/2] (both real) share column 0: E[0] + i O[0], doubled like the rest
            } else {
                const cx<T> q = lds[lds_addr<C>(pos.bo, 0, N2 - k)];
                const cx<T> ev = {z.x + q.x, z.y - q.y};                                   // 2 E[k] = Z[k] + conj Z[N/2 - k]
                const cx<T> od = cmulc(cx<T>{z.x - q.x, z.y + q.y}, cmul(wt, w32<T>(m)));    // 2 O[k] = (Z[k] - conj Z[N/2 - k]) conj W_N^k
                zp = ev + mul_pi(od);
            }
            v[e][m] = {zp.x, -zp.y};
        }
    }
    __syncthreads();     // the transform's exchange reuses this LDS
    if constexpr (C::E == 2 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else fft_run<C>(v, pos, pm_smem, tw);
    store<C>(sp, unit, pos, v);      // conj-out and the scale ride on the store; a complex element is two real samples
}

template <typename T, int LOGM>
int launch_col_mul_herm_one(const ColLoadTiled<T>& lp, const HermMul<T>& hm, const ColStoreTiled<T>& sp, const cx<T>* tw, int ntiles, int log_g,
                            hipStream_t st) {
    using C = typename ColCfgSel<T, LOGM, 0>::type;
    auto kern = fft_col_mul_herm_kernel<C>;
    constexpr size_t part = size_t(C::BO) * C::N * sizeof(cx<T>);
    constexpr size_t LDSB = C::LDS_BYTES > part ? C::LDS_BYTES : part;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int grid = (ntiles + C::BO - 1) / C::BO;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid, hm.planes ? 2 : 1), dim3(C::NT), LDSB, st, lp, hm, sp, tw, log_g);
    return int(hipGetLastError());
}

template <typename T>
int launch_col_mul_herm_impl(int logm, const ColLoadTiled<T>& lp, const HermMul<T>& hm, const ColStoreTiled<T>& sp, const cx<T>* tw, int ntiles,
                             int log_g, hipStream_t st) {
    switch (logm) {
#define PM_CASE(k) \
    case k:        \
        return launch_col_mul_herm_one<T, k>(lp, hm, sp, tw, ntiles, log_g, st);
        PM_CASE(1) PM_CASE(2) PM_CASE(3) PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASE(11) PM_CASE(12)
        PM_CASE(13)
#undef PM_CASE
        default: return -2;
    }
}

template <typename T, int LOGN2, int VAR>
int launch_row_c2r_one(const RowLoadTiled<T>& lp, const RowStoreNat<T>& sp, const cx<T>* tw, const cx<T>* twn, int nseq, hipStream_t st) {
    using C = typename RowCfgSel<T, LOGN2, VAR>::type;
    auto kern = fft_row_c2r_kernel<C, RowLoadTiled<T>>;
    constexpr size_t part = size_t(C::LDS_ELEMS) * sizeof(cx<T>);
    constexpr size_t LDSB = C::LDS_BYTES > part ? C::LDS_BYTES : part;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int per_wg = C::BO * C::E;
    const int grid = (nseq + per_wg - 1) / per_wg;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), LDSB, st, lp, sp, tw, twn);
    return int(hipGetLastError());
}

template <typename T>
int launch_row_c2r_impl(int logn2, const RowLoadTiled<T>& lp, const RowStoreNat<T>& sp, const cx<T>* tw, const cx<T>* twn, int nseq,
                        hipStream_t st) {
    switch (logn2) {
#define PM_CASE(k) \
    case k:        \
        return launch_row_c2r_one<T, k, 0>(lp, sp, tw, twn, nseq, st);
        PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10)
#undef PM_CASE
        case 11:
            if constexpr (sizeof(T) == 4) return launch_row_c2r_one<T, 11, 5>(lp, sp, tw, twn, nseq, st);
            else return launch_row_c2r_one<T, 11, 0>(lp, sp, tw, twn, nseq, st);
        case 12:
            if constexpr (sizeof(T) == 4) return launch_row_c2r_one<T, 12, 4>(lp, sp, tw, twn, nseq, st);
            else return -2;     // complex128 rows of 4096 complex points exchange re / im separately: not on this path
        default: return -2;
    }
}

template <typename T, int LOGN2>
int launch_row_c2r_fold_one(const RowLoadFold<T>& lp, const RowStoreNat<T>& sp, const cx<T>* tw, const cx<T>* twn, int npairs, hipStream_t st) {
    using C = typename RowCfgSel<T, LOGN2, 4>::type;       // two rows per thread: the pair (n, n + M/2) the unfold rebuilds
    auto kern = fft_row_c2r_kernel<C, RowLoadFold<T>>;
    constexpr size_t part = size_t(C::LDS_ELEMS) * sizeof(cx<T>);
    constexpr size_t LDSB = C::LDS_BYTES > part ? C::LDS_BYTES : part;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int grid = (npairs + C::BO - 1) / C::BO;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), LDSB, st, lp, sp, tw, twn);
    return int(hipGetLastError());
}

template <typename T>
int launch_row_c2r_fold_impl(int logn2, const RowLoadFold<T>& lp, const RowStoreNat<T>& sp, const cx<T>* tw, const cx<T>* twn, int npairs,
                             hipStream_t st) {
    switch (logn2) {
        case 11: return launch_row_c2r_fold_one<T, 11>(lp, sp, tw, twn, npairs, st);
        case 12:
            if constexpr (sizeof(T) == 4) return launch_row_c2r_fold_one<T, 12>(lp, sp, tw, twn, npairs, st);
            else return -2;
        default: return -2;
    }
}

}  // namespace pm
