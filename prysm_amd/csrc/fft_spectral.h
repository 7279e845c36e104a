// The wavelength loop of a polychromatic PSF as ONE launch pair per group of wavelengths
// (docs/source/how-tos/Polychromatic Propagation.ipynb cell 3: for each wavelength, pupil = amp exp(i k opd), focus, |.|^2,
// weighted sum).  The loop's per-wavelength transform reads the packed (amplitude, OPD) map and read-modify-writes the
// accumulator once per wavelength: 8 + 8 + 8 + (4 + 4) = 32 bytes per sample and wavelength.  Here
//   rows     a workgroup reads its rows of the packed map ONCE, and for each wavelength b synthesises the pupil in registers,
//            transforms it and stores it to intermediate b;
//   columns  a workgroup owns its tile of columns for ALL wavelengths: transform of intermediate b, w_b |.|^2 summed in
//            registers, the accumulator read and written once at the end,
// i.e. 16 + 16 / B bytes per sample and wavelength for groups of B (18 at B = 8; twice that in complex128).
// The sum runs in wavelength order; against the loop only the association differs (old + (w_0 i_0 + w_1 i_1 + ...)).
#pragma once
#include <hip/hip_runtime.h>

#include "fft_kernels.h"
#include "fft_spectral_types.h"

namespace pm {

// KEEP: the (amplitude, OPD) pairs stay in registers across the wavelengths (twice the data registers); else each wavelength
// reloads them -- from the L2 of the XCD the workgroup runs on, where its first read left them
template <typename C, int VAR, typename S, bool KEEP>
__global__ void __launch_bounds__(C::NT) fft_row_spectral_kernel(const RowLoadNat<typename C::T> lp, const S sp,
                                                                 const cx<typename C::T>* __restrict__ tw, const int log_g, const Spectral w) {
    using T = typename C::T;
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = group_remap(blockIdx.x, gridDim.x, log_g);
    cx<T> raw[KEEP ? C::E : 1][KEEP ? C::P : 1];      // (amplitude, OPD) pairs; zero outside the window -> zero amplitude
    if constexpr (KEEP) load_sel<C, 0>(lp, unit, pos, raw);
#pragma unroll 1
    for (int b = 0; b < w.nb; ++b) {
        const double k2 = w.k2[b];
        cx<T> v[C::E][C::P];
        if constexpr (KEEP) {
#pragma unroll
            for (int e = 0; e < C::E; ++e)
#pragma unroll
                for (int m = 0; m < C::P; ++m) v[e][m] = synth_value<T>(raw[e][m].y, raw[e][m].x, k2);
        } else {
            RowLoadNat<T> lpb = lp;
            lpb.k2 = k2;
            lpb.nt = 0;     // the next wavelength reads the same rows again
            load_sel<C, 3>(lpb, unit, pos, v);
        }
        if (b) __syncthreads();     // the exchange buffer of the previous transform is still being read
        // complex128: an opaque copy of the slot per wavelength, or the transform's twiddle products are hoisted out of the loop and
        // held in registers across it (256 VGPRs, one wave per SIMD); complex64 has the room and keeps them
        ThreadPos pb = pos;
        if constexpr (sizeof(T) == 8) asm volatile("" : "+v"(pb.t), "+v"(pb.cl), "+v"(pb.bo));
        if constexpr (VAR != 5 && C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pb, pm_smem, tw);
        else fft_run<C>(v, pb, pm_smem, tw);
        S spb = sp;
        spb.dst += int64_t(b) * w.fstride;
        store<C>(spb, unit, pos, v);
    }
}

// acc (real, already weighted and scaled) += into the output view of the column store
template <typename C>
PM_HD void store_acc(const ColStoreNat<typename C::T>& p, int tile, ThreadPos pos, const typename C::T (&acc)[C::E][C::P]) {
    using T = typename C::T;
    constexpr int TC = C::CI * C::E;
    const int col0 = tile * TC + pos.cl * C::E;
    T* dst = reinterpret_cast<T*>(p.dst);
    int qx[C::E];
#pragma unroll
    for (int e = 0; e < C::E; ++e) qx[e] = (col0 + e < p.ax.n) ? p.ax.map(col0 + e) : -1;
    bool pair = false;
    if constexpr (C::E == 2) pair = (p.vec_ok & 2) && qx[0] >= 0 && qx[1] == qx[0] + 1 && (qx[0] & 1) == 0;
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int qy = p.ay.map(pos.t + m * C::TPS);
        if (qy < 0) continue;
        T* a = dst + int64_t(qy) * p.ld;
        if constexpr (C::E == 2) {
            if (pair) {
                cx<T>* a2 = reinterpret_cast<cx<T>*>(a + qx[0]);
                const cx<T> old = *a2;
                *a2 = cx<T>{old.x + acc[0][m], old.y + acc[1][m]};
                continue;
            }
        }
#pragma unroll
        for (int e = 0; e < C::E; ++e)
            if (qx[e] >= 0) a[qx[e]] += acc[e][m];
    }
}

// REGACC: w_b |.|^2 summed in registers, the accumulator touched once (32 more registers: one 512-thread workgroup per CU where the
// plain column pass runs two); else each wavelength read-modify-writes the accumulator through the ordinary epilogue -- the
// workgroup's own lines, so from the second wavelength on that is L2 / Infinity Cache traffic
template <typename C, bool REGACC>
__global__ void __launch_bounds__(C::NT) fft_col_spectral_kernel(const ColLoadTiled<typename C::T> lp0, const ColStoreNat<typename C::T> sp0,
                                                                 const cx<typename C::T>* __restrict__ tw, const int log_g, const Spectral w) {
    using T = typename C::T;
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = group_remap(blockIdx.x, gridDim.x, log_g) * C::BO + pos.bo;
    ColLoadTiled<T> lp = at_batch(lp0, blockIdx.y);     // blockIdx.y: plane of a folded transform
    const ColStoreNat<T> sp = at_batch(sp0, blockIdx.y);
    const T s2 = sp.scale * sp.scale;
    T acc[REGACC ? C::E : 1][REGACC ? C::P : 1];
    if constexpr (REGACC) {
#pragma unroll
        for (int e = 0; e < C::E; ++e)
#pragma unroll
            for (int m = 0; m < C::P; ++m) acc[e][m] = T(0);
    }
#pragma unroll 1
    for (int b = 0; b < w.nb; ++b) {
        cx<T> v[C::E][C::P];
        load<C>(lp, unit, pos, v);
        lp.src += w.fstride;
        if (b) __syncthreads();
        ThreadPos pb = pos;         // see the row kernel
        if constexpr (sizeof(T) == 8) asm volatile("" : "+v"(pb.t), "+v"(pb.cl), "+v"(pb.bo));
        if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pb, pm_smem, tw);
        else fft_run<C>(v, pb, pm_smem, tw);
        const T wb = T(w.w[b]);
        if constexpr (REGACC) {
#pragma unroll
            for (int e = 0; e < C::E; ++e)
#pragma unroll
                for (int m = 0; m < C::P; ++m) acc[e][m] += wb * ((v[e][m].x * v[e][m].x + v[e][m].y * v[e][m].y) * s2);
        } else {
            ColStoreNat<T> spb = sp;
            spb.weight = wb;
            spb.nt = 0;
            store<C>(spb, unit, pos, v);
        }
    }
    if constexpr (REGACC) store_acc<C>(sp, unit, pos, acc);
}

template <typename T, int LOGN, int VAR, typename S>
int launch_row_spectral_one(const RowLoadNat<T>& lp, const S& sp, const cx<T>* tw, int units, int log_g, const Spectral& w,
                            hipStream_t st) {
    using C = typename RowCfgSel<T, LOGN, VAR>::type;
    auto kern = fft_row_spectral_kernel<C, VAR, S, true>;
    constexpr size_t LDSB = C::LDS_BYTES;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    // units: rows (a thread owns E consecutive ones) or, folded, row pairs
    const int per_wg = C::BO * (std::is_same<S, RowStoreFold<T>>::value ? 1 : C::E);
    const int grid = (units + per_wg - 1) / per_wg;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), LDSB, st, lp, sp, tw, log_g, w);
    return int(hipGetLastError());
}

template <typename T, int LOGM>
int launch_col_spectral_one(const ColLoadTiled<T>& lp, const ColStoreNat<T>& sp, const cx<T>* tw, int ntiles, int log_g,
                            const Spectral& w, hipStream_t st, int nplanes) {
    using C = typename ColCfgSel<T, LOGM, 0>::type;
    auto kern = fft_col_spectral_kernel<C, true>;
    constexpr size_t LDSB = C::LDS_BYTES;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int grid = (ntiles + C::BO - 1) / C::BO;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid, nplanes), dim3(C::NT), LDSB, st, lp, sp, tw, log_g, w);
    return int(hipGetLastError());
}

}  // namespace pm
