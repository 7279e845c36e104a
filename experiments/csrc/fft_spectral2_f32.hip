// Grouped-wavelength transform pair at four waves per SIMD, complex64 (kernels + explicit launchers; see fft_spectral2.h).
// EXPERIMENT BUILD ONLY (-DPM_EXPERIMENTS, `make -C tools exp`): measured against the per-wavelength loop and round 3's groups of 8
// on MI355X (profiles/r04/exp_spectral2.log) it lost at every size -- 4096^2: 109.0 us per wavelength in groups of 4 against 108.3 (loop)
// and 106.1; 2048^2: 23.7 against 21.5 (groups of 8); 1024^2: 13.7 against 10.4 -- see DESIGN.md 3.3d for why fewer bytes did not buy time.
#ifdef PM_EXPERIMENTS
#include "fft_kernels.h"
#include "fft_spectral2.h"

namespace pm {

// rows: E = the two wavelengths of a pair; FOLD: BO = 2 halves of the workgroup = memory rows (i, i + M/2)
template <int LOGN, bool FOLD>
struct Sp2RowCfg {
    static constexpr int N = 1 << LOGN, TPS = N / 16;
    static constexpr int BO = FOLD ? 2 : (TPS >= 256 ? 1 : 256 / TPS);
    using type = FftCfg<float, LOGN, 1, 2, BO, 1>;
};

// NPAIR pairs of wavelengths per workgroup, fully unrolled (as a run-time loop the register allocator keeps the raw values AND the loop-
// invariant twiddle products live through the transform and spills hundreds of registers under the 128-register cap).  NPAIR = 2:
//   KEEP   the raw (amplitude, OPD) values stay in registers between the two pairs: 161 VGPRs, three waves per SIMD;
//   else   they are read again -- from L2 / Infinity Cache, where the first read left them -- between the stores of the pair's two
//          wavelengths: the first wavelength's registers are free by then, and the vector memory counter is in order, so the reload
//          waits for 16 stores instead of 32 while the other 16 drain behind it.  128 VGPRs, four waves per SIMD.
template <typename C, bool FOLD, int NPAIR, bool KEEP>
__global__ void __launch_bounds__(C::NT, (KEEP && NPAIR > 1) ? 3 : 4)
    fft_row_spectral2_kernel(const Sp2Row<float> g, const cx<float>* __restrict__ tw, const Spectral w) {
    using T = float;
    static_assert(C::TPS >= 64, "the half of the workgroup a wave belongs to must be uniform over the wave");
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const int t = threadIdx.x % C::TPS;
    const int bo = PM_UNIFORM(int(threadIdx.x) / C::TPS);
    const ThreadPos pos{0, t, bo};
    int unit, memrow;
    bool ok = true;
    if constexpr (FOLD) {
        unit = blockIdx.x;                  // the pair of rows (i, i + M/2): grid = M/2 exactly
        memrow = unit + bo * g.drows;
    } else {
        unit = memrow = blockIdx.x * C::BO + bo;
        ok = memrow < g.nrows;
    }
    cx<T> wf = {T(1), T(0)};
    if constexpr (FOLD) wf = g.twm[unit];
    cx<T> raw[C::P];
    sp2_row_load_sel<C>(g, memrow, ok, t, raw);
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
        const int b = 2 * p;
        const bool two = b + 1 < w.nb;      // an odd group: the second sequence of the last pair is computed and dropped
        cx<T> v[C::E][C::P];
        sp2_synth<C>(raw, w.k2[b], w.k2[two ? b + 1 : b], v);
        if (p) __syncthreads();             // the exchange buffer of the previous transform is still being read
        if constexpr (FOLD) {
            cx<T>* lds = reinterpret_cast<cx<T>*>(pm_smem);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                sp2_fold_write<C>(v[e], t, bo, lds);
                __syncthreads();
                sp2_fold_combine<C>(v[e], t, bo, lds, wf, g.swap);
                __syncthreads();
            }
        }
        // an opaque copy of the slot per pair: the stage twiddles and their products depend on the slot only and would be shared by the
        // two transforms, i.e. held across the first
        ThreadPos pb = pos;
        asm volatile("" : "+v"(pb.t));
        fft_run_pipe2<C>(v, pb, pm_smem, tw);
        if (ok) sp2_row_store<C>(g, unit, FOLD ? bo : 0, b, t, v[0]);
        if constexpr (!KEEP && NPAIR > 1) {
            if (p + 1 < NPAIR) sp2_row_load_sel<C>(g, memrow, ok, t, raw);
        }
        if (ok && two) sp2_row_store<C>(g, unit, FOLD ? bo : 0, b + 1, t, v[1]);
    }
}

template <int LOGN, bool FOLD>
static int launch_row2_one(bool keep, const Sp2Row<float>& g, const cx<float>* tw, const Spectral& w, hipStream_t st) {
    using C = typename Sp2RowCfg<LOGN, FOLD>::type;
    // the fold's exchange area: both halves' 16 x TPS values of one wavelength
    static_assert(!FOLD || size_t(2) * C::N * sizeof(cx<float>) <= C::LDS_BYTES, "fold exchange fits the transform's LDS");
    if (w.nb < 1 || w.nb > 4) return -2;
    auto kern = w.nb <= 2 ? fft_row_spectral2_kernel<C, FOLD, 1, false>
                          : (keep ? fft_row_spectral2_kernel<C, FOLD, 2, true> : fft_row_spectral2_kernel<C, FOLD, 2, false>);
    constexpr size_t LDSB = C::LDS_BYTES;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int grid = FOLD ? g.drows : (g.nrows + C::BO - 1) / C::BO;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), LDSB, st, g, tw, w);
    return int(hipGetLastError());
}

int launch_row_spectral2(int logn, bool fold, bool keep, const Sp2Row<float>& g, const cx<float>* tw, const Spectral& w, hipStream_t st) {
    switch (logn) {
        case 10: return fold ? launch_row2_one<10, true>(keep, g, tw, w, st) : launch_row2_one<10, false>(keep, g, tw, w, st);
        case 11: return fold ? launch_row2_one<11, true>(keep, g, tw, w, st) : launch_row2_one<11, false>(keep, g, tw, w, st);
        case 12: return fold ? launch_row2_one<12, true>(keep, g, tw, w, st) : launch_row2_one<12, false>(keep, g, tw, w, st);
        default: return -2;
    }
}

// columns: one tile of 8 columns per workgroup for all wavelengths of the group, w_b |.|^2 summed in registers
template <typename C>
__global__ void __launch_bounds__(C::NT, 4) fft_col_spectral2_kernel(const Sp2Col<float> g, const cx<float>* __restrict__ tw, const int log_g,
                                                                     const Spectral w) {
    using T = float;
    static_assert(C::BO == 1, "one tile per workgroup");
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = group_remap(blockIdx.x, gridDim.x, log_g);
    const Sp2Addr<C> A(pos, g.log_k);
    const cx<T>* src = g.src + int64_t(blockIdx.y) * g.plane_stride;
    T acc[C::E][C::P];
#pragma unroll
    for (int e = 0; e < C::E; ++e)
#pragma unroll
        for (int m = 0; m < C::P; ++m) acc[e][m] = T(0);
#pragma unroll 1
    for (int b = 0; b < w.nb; ++b) {
        cx<T> v[C::E][C::P];
        if (g.rot) sp2_col_load<C, C::P / 2>(g, src, unit, A, v);
        else sp2_col_load<C, 0>(g, src, unit, A, v);
        src += g.fstride;
        if (b) __syncthreads();
        ThreadPos pb = pos;     // opaque per wavelength: see the row kernel
        asm volatile("" : "+v"(pb.t), "+v"(pb.cl));
        fft_run_pipe2<C>(v, pb, pm_smem, tw);
        sp2_accumulate<C>(acc, v, T(w.w[b]), g.s2);
    }
    T* dst = g.dst + int64_t(blockIdx.y) * g.out_plane;
    if (g.orot) sp2_col_store<C, C::P / 2>(g, dst, unit, pos, acc);
    else sp2_col_store<C, 0>(g, dst, unit, pos, acc);
}

template <int LOGM>
static int launch_col2_one(const Sp2Col<float>& g, const cx<float>* tw, int ntiles, int log_g, const Spectral& w, hipStream_t st, int nplanes) {
    using C = typename ColCfgSel<float, LOGM, 0>::type;
    auto kern = fft_col_spectral2_kernel<C>;
    constexpr size_t LDSB = C::LDS_BYTES;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    if (ntiles <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(ntiles, nplanes), dim3(C::NT), LDSB, st, g, tw, log_g, w);
    return int(hipGetLastError());
}

int launch_col_spectral2(int logm, const Sp2Col<float>& g, const cx<float>* tw, int ntiles, int log_g, const Spectral& w, hipStream_t st,
                         int nplanes) {
    switch (logm) {
        case 10: return launch_col2_one<10>(g, tw, ntiles, log_g, w, st, nplanes);
        case 11: return launch_col2_one<11>(g, tw, ntiles, log_g, w, st, nplanes);
        default: return -2;
    }
}

}  // namespace pm
#endif   // PM_EXPERIMENTS
