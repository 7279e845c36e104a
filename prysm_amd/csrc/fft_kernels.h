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
// VAR (tuning variant, PM_TUNE / pm_set_tuning): row pass VAR = 1 doubles the rows per workgroup (each
// workgroup then writes whole 128 B lines of the tiled intermediate); column pass VAR = 1 takes the stage
// twiddles from a per-workgroup LDS table instead of global gathers (measured 6 % slower at 4096^2: the
// extra registers spill under the 128-VGPR cap of the 1024-thread workgroup; kept as an A/B knob).
template <typename T, int LOGN, int VAR>
struct RowCfgSel {
    static constexpr int N = 1 << LOGN, P = N >= 16 ? 16 : N, TPS = N / P;
    // VAR = 5: one sequence per (small) workgroup -- twice the workgroups at 2048 points, finer overlap of their phases
    static constexpr int BO = (TPS >= 256 || VAR == 5) ? 1 : 256 / TPS;
    // VAR = 1: exchange real and imaginary parts separately -> half the LDS per workgroup, so the 72-VGPR
    // complex64 kernel fits 7 workgroups per CU instead of 4 (the row pass is latency / concurrency bound)
    static constexpr int COMP = ((sizeof(T) == 8 && LOGN >= 12) || VAR == 1) ? 2 : 1;   // VAR = 2: persistent kernel
    // VAR = 4: two consecutive rows per thread -- stage twiddles, their products and the LDS addressing are shared by
    // the pair (the row pass spends more VALU time per point than the column pass, which already works this way)
    static constexpr int E = (VAR == 4) ? 2 : 1;
    using type = FftCfg<T, LOGN, 1, E, BO, COMP>;
};
// Column pass: a tile of 64 B rows (8 complex64 / 4 complex128 columns) per workgroup; at
// M = 4096 that is 256 KiB of field in the registers of 1024 threads, exchanged through LDS in
// two 136 KiB chunks (complex64: the thread's two columns; complex128: real then imaginary).
template <typename T, int LOGN, int VAR>
struct ColCfgSel {
    static constexpr int N = 1 << LOGN, P = N >= 16 ? 16 : N, TPS = N / P;
    static constexpr int E = sizeof(T) == 4 ? 2 : 1;
    // VAR = 2: tiles of 128 B rows (16 complex64 / 8 complex128 columns) for 2048-point columns: whole cache lines per store
    static constexpr int CI = (VAR == 2 && LOGN == 11) ? 8 : (LOGN <= 12 ? 4 : 2);   // VAR = 1: stage twiddles from an LDS table (A/B knob; measured slower)
    static constexpr int BO = (CI * TPS >= 256) ? 1 : 256 / (CI * TPS);
    static constexpr int COMP = (sizeof(T) == 8 && LOGN >= 11) ? 2 : 1;
    using type = FftCfg<T, LOGN, CI, E, BO, COMP>;
};

// stage twiddles from an LDS table: column pass, complex64, when the table fits behind the exchange chunk
template <typename C, bool COL, int VAR = 0>
constexpr bool use_tw_lds() {
    return COL && VAR == 1 && sizeof(typename C::T) == 4 && C::NSTAGE > 1 &&
           (C::LDS_BYTES + size_t(tw_lds_entries<C>()) * sizeof(cx<typename C::T>) <= 160 * 1024);
}
template <typename C, bool COL, int VAR = 0>
constexpr size_t kernel_lds_bytes() {
    return C::LDS_BYTES + (use_tw_lds<C, COL, VAR>() ? size_t(tw_lds_entries<C>()) * sizeof(cx<typename C::T>) : 0);
}

template <typename C, bool COL, int VAR, typename L, typename S>
__global__ void __launch_bounds__(C::NT) fft_kernel(const L lp, const S sp,
                                                    const cx<typename C::T>* __restrict__ tw, const int log_g_skew) {
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    // log_g_skew: bits 0-7 sibling-group size, bits 8+ start skew (column pass, experiment): every other workgroup
    // of the first wave sleeps skew x ~0.85 us so the CUs do not run their load / compute / store phases in lockstep
    const int log_g = log_g_skew & 0xff;
    if (((log_g_skew >> 8) & 0xff) && ((blockIdx.x >> 3) & 1)) {
        for (int i = 0; i < ((log_g_skew >> 8) & 0xff); ++i) __builtin_amdgcn_s_sleep(32);
    }
    int unit = group_remap(blockIdx.x, gridDim.x, log_g);
    if (COL && (log_g_skew >> 16)) {
        // experiment: spread the sibling groups that run at the same time across the whole row (stride permutation
        // of the group index) instead of one contiguous span
        const int ls = log_g_skew >> 16, G = 1 << log_g;
        const int ngroups = int(gridDim.x) >> log_g, SP = 1 << ls;
        if ((int(gridDim.x) & (G - 1)) == 0 && (ngroups & (SP - 1)) == 0) {
            const int gi = unit >> log_g, within = unit & (G - 1);
            const int g2 = (gi & (SP - 1)) * (ngroups >> ls) + (gi >> ls);
            unit = (g2 << log_g) | within;
        }
    }
    if (COL) unit = unit * C::BO + pos.bo;
    cx<typename C::T> v[C::E][C::P];
    const L lpb = at_batch(lp, blockIdx.y);   // blockIdx.y: field of a batch
    const S spb = at_batch(sp, blockIdx.y);
    if constexpr (use_tw_lds<C, COL, VAR>()) {
        // table first (two gathers + two ds_write per thread), then the tile loads; the barrier that publishes
        // the table is passed while the tile loads are still in flight
        cx<typename C::T>* tab = reinterpret_cast<cx<typename C::T>*>(pm_smem + C::LDS_BYTES);
        fill_tw_lds<C>(tab, threadIdx.x, C::NT, tw);
        load<C>(lpb, unit, pos, v);
        __syncthreads();
        fft_run_twlds<C>(v, pos, pm_smem, tab);
    } else {
        if constexpr (VAR == 6 || VAR == 7) {
            // compute only (timing experiments): registers filled from the thread index, results stored only under a
            // condition that never holds
#pragma unroll
            for (int e = 0; e < C::E; ++e)
#pragma unroll
                for (int m = 0; m < C::P; ++m) v[e][m] = {typename C::T(threadIdx.x + m), typename C::T(unit + e)};
        } else {
            load<C>(lpb, unit, pos, v);
        }
        // two sequences per thread (complex64 columns): exchanges pipelined against the butterflies of the other
        // sequence (measured: 4096^2 column pass 57.2 -> 56.1 us, 8192^2 419 -> 358 us); VAR = 5 keeps the plain order
        // for A/B runs, VAR = 3 skips the transform (memory phases only: timing experiments, wrong results)
        if constexpr (VAR == 7) {   // butterflies only, no exchange (timing experiments; wrong results)
            stage_compute<C, 0>(v, pos.t, tw);
            if constexpr (C::NSTAGE > 1) stage_compute<C, 1>(v, pos.t, tw);
            if constexpr (C::NSTAGE > 2) stage_compute<C, 2>(v, pos.t, tw);
            if constexpr (C::NSTAGE > 3) stage_compute<C, 3>(v, pos.t, tw);
        } else if constexpr (VAR != 5 && VAR != 3 && C::E == 2 && C::COMP == 1 && C::NSTAGE > 1)
            fft_run_pipe2<C>(v, pos, pm_smem, tw);
        else if constexpr (VAR != 3)
            fft_run<C>(v, pos, pm_smem, tw);
    }
    if constexpr (VAR == 6 || VAR == 7) {
        if (v[0][0].x != typename C::T(-12345.5)) return;
    }
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

// Persistent row pass: each workgroup walks units g, g + G, g + 2G, ... and keeps TWO register sets: the loads
// of the next unit are issued before the current one is transformed and stay in flight under its three
// radix-16 stages (the wave only waits for them at the top of the next half-iteration); stores are fire and
// forget.  Twiddles are row-invariant and live in registers for the whole kernel.  This turns the
// load -> compute -> store chain of one workgroup into a pipeline without needing more resident workgroups.
template <typename C, typename L, typename S>
__global__ void __launch_bounds__(C::NT, (C::NT <= 256 ? 4 : 2)) fft_row_persistent_kernel(const L lp, const S sp,
                                                                   const cx<typename C::T>* __restrict__ tw,
                                                                   const int nunits, const int log_g) {
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int stride = gridDim.x;
    cx<typename C::T> va[C::E][C::P], vb[C::E][C::P];
    int u = blockIdx.x;
    if (u < nunits) load<C>(lp, group_remap(u, nunits, log_g), pos, va);
    while (u < nunits) {
        // An opaque copy of the thread slot per iteration: without it LICM hoists every loop-invariant
        // address, twiddle and twiddle product out of the loop (> 300 live registers, spills).
        ThreadPos p = pos;
        asm volatile("" : "+v"(p.t));
        // order matters for the in-order vmcnt: twiddles of THIS unit first, then the prefetch of the next
        // unit, so waiting for the twiddles never waits for the prefetch
        TwSet<C> ts;
        load_tw_set<C>(ts, p.t, tw);
        const int u1 = u + stride;
        if (u1 < nunits) load<C>(lp, group_remap(u1, nunits, log_g), p, vb);
        // pin the issue point: with predicate-free (full-window) loads nothing else stops the machine
        // scheduler from sinking the prefetch next to its first use to save registers, which would serialise
        // the row pipeline again
        __builtin_amdgcn_sched_barrier(0);
        fft_run_tw<C>(va, p, pm_smem, ts);
        store<C>(sp, group_remap(u, nunits, log_g), p, va);
        // rotate the register sets (32 moves); the wait for the prefetched loads lands here, after the
        // stores of this unit were issued (those loads are older than the stores)
#pragma unroll
        for (int e = 0; e < C::E; ++e)
#pragma unroll
            for (int m = 0; m < C::P; ++m) va[e][m] = vb[e][m];
        u = u1;
    }
}

template <typename T, bool COL, int LOGN, int VAR, typename L, typename S>
int launch_one(const L& lp, const S& sp, const cx<T>* tw, int units, int log_g, hipStream_t st, int nbatch) {
    using Sel = typename std::conditional<COL, ColCfgSel<T, LOGN, VAR>, RowCfgSel<T, LOGN, VAR>>::type;
    using C = typename Sel::type;
    auto kern = fft_kernel<C, COL, (COL || VAR == 3 || VAR == 4 || VAR == 5 || VAR == 6 || VAR == 7 ? VAR : 0), L, S>;
    constexpr size_t LDSB = kernel_lds_bytes<C, COL, (COL ? VAR : 0)>();
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int per_wg = C::BO * (COL ? 1 : C::E);     // row mode: a thread owns E consecutive rows
    const int grid = (units + per_wg - 1) / per_wg;
    if (grid <= 0) return 0;
    if (nbatch <= 0) return 0;
    if constexpr (!COL && VAR == 2) {
        if (nbatch == 1) {
        // persistent, double-buffered (row pass, tuning row_var = 2)
        auto pk = fft_row_persistent_kernel<C, L, S>;
        if (C::LDS_BYTES > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pk), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               int(C::LDS_BYTES));
            if (e != hipSuccess) return int(e);
        }
        const int lds_wg = C::LDS_BYTES ? int(160 * 1024 / C::LDS_BYTES) : 8;
        int per_cu = lds_wg < 4 ? lds_wg : 4;
        if (per_cu < 1) per_cu = 1;
        int pgrid = 256 * per_cu;
        if (pgrid > grid) pgrid = grid;
        hipLaunchKernelGGL(pk, dim3(pgrid), dim3(C::NT), C::LDS_BYTES, st, lp, sp, tw, grid, log_g);
        return int(hipGetLastError());
        }   // batches take the plain kernel (the grid already holds many workgroups per CU)
    }
    hipLaunchKernelGGL(kern, dim3(grid, nbatch), dim3(C::NT), LDSB, st, lp, sp, tw, log_g);
    return int(hipGetLastError());
}

int tuning_row_var_for_timing();   // capi.hip: the row_var knob (timing builds only)

// folded row pass (RowStoreFold): units are row PAIRS (i, i + M/2); complex64 / complex128 rows of 2048 .. 8192 points
template <typename T, int LOGN, int KVAR = 4>
int launch_fold_one(const RowLoadNat<T>& lp, const RowStoreFold<T>& sp, const cx<T>* tw, int npairs, int log_g, hipStream_t st,
                    int nbatch) {
    using C = typename RowCfgSel<T, LOGN, 4>::type;
    auto kern = fft_kernel<C, false, KVAR, RowLoadNat<T>, RowStoreFold<T>>;
    constexpr size_t LDSB = kernel_lds_bytes<C, false, 0>();
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
#ifdef PM_TIMING_VARIANTS   // row_var = 3 / 6 / 7 on the folded row pass (4096 points): memory only / arithmetic only / butterflies only
    if (logn == 12 && (tuning_row_var_for_timing() == 3 || tuning_row_var_for_timing() == 6 || tuning_row_var_for_timing() == 7)) {
        const int v = tuning_row_var_for_timing();
        if (v == 3) return launch_fold_one<T, 12, 3>(lp, sp, tw, npairs, log_g, st, nbatch);
        if (v == 6) return launch_fold_one<T, 12, 6>(lp, sp, tw, npairs, log_g, st, nbatch);
        return launch_fold_one<T, 12, 7>(lp, sp, tw, npairs, log_g, st, nbatch);
    }
#endif
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
    constexpr size_t LDSB = kernel_lds_bytes<C, false, 0>();
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
// timing variants (memory phases only / arithmetic only / butterflies only; WRONG results, used for the "where the time
// goes" analysis in DESIGN.md) exist only in builds with -DPM_TIMING_VARIANTS (make EXTRA=-DPM_TIMING_VARIANTS)
#ifdef PM_TIMING_VARIANTS
#define PM_TIMING_CASES(k)                                                                      \
    if (var == 3) return launch_one<T, COL, k, 3, L, S>(lp, sp, tw, units, log_g, st, nbatch); \
    if (var == 6) return launch_one<T, COL, k, 6, L, S>(lp, sp, tw, units, log_g, st, nbatch); \
    if (var == 7) return launch_one<T, COL, k, 7, L, S>(lp, sp, tw, units, log_g, st, nbatch);
#else
#define PM_TIMING_CASES(k)
#endif
#define PM_CASEV(k)                                                       \
    case k:                                                               \
        if (var == 1) return launch_one<T, COL, k, 1, L, S>(lp, sp, tw, units, log_g, st, nbatch); \
        if (var == 2 && !COL) return launch_one<T, COL, k, (COL ? 0 : 2), L, S>(lp, sp, tw, units, log_g, st, nbatch); \
        PM_TIMING_CASES(k) \
        if (var == 5 && COL) return launch_one<T, COL, k, (COL ? 5 : 0), L, S>(lp, sp, tw, units, log_g, st, nbatch); \
        if (var == 2 && COL && k == 11) return launch_one<T, COL, k, (COL ? 2 : 0), L, S>(lp, sp, tw, units, log_g, st, nbatch); \
        if (var == 4 && !COL) return launch_one<T, COL, k, (COL ? 0 : 4), L, S>(lp, sp, tw, units, log_g, st, nbatch); \
        if (var == 5 && !COL) return launch_one<T, COL, k, (COL ? 0 : 5), L, S>(lp, sp, tw, units, log_g, st, nbatch); \
        return launch_one<T, COL, k, 0, L, S>(lp, sp, tw, units, log_g, st, nbatch);
        PM_CASE(1) PM_CASE(2) PM_CASE(3) PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7)
        PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASEV(11) PM_CASEV(12) PM_CASEV(13)
#undef PM_CASE
#undef PM_CASEV
#undef PM_TIMING_CASES
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
