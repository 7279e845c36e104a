// Real-input 2-D transform on half the spectrum (Hermitian symmetry): fft2 of a real PSF / object
// (prysm/otf.py:28-33 transform_psf, :62-135 the centre-normalised MTF / PTF / OTF).
//
//   row pass   : a real row of N samples IS the complex sequence z[j] = x[2j] + i x[2j+1] of N/2 points; one N/2-point engine
//                transform Z, then (partner exchange through LDS)
//                    Xe[k] = (Z[k] + conj Z[N/2 - k]) / 2,  Xo[k] = (Z[k] - conj Z[N/2 - k]) / (2i),  X[k] = Xe[k] + W_N^k Xo[k]
//                for k = 0 .. N/2 - 1 and X[N/2] = Re Z[0] - Im Z[0].  X[0] and X[N/2] are real, so column 0 of the tiled
//                intermediate carries both, X[0] + i X[N/2]: exactly N/2 columns (half the bytes and half the butterflies of the
//                complex row pass, and no 257th column tile that would run alone in a second round of workgroups).
//   column pass: ordinary M-point column transforms of those N/2 columns (half the tiles); the packed column is separated after
//                its transform (C[u] and C[M - u] through LDS, in the workgroup of tile 0); every result F[u][k] is stored
//                twice -- at (u, k) and, conjugated, at ((M - u) mod M, N - k) -- through the caller's output view, optionally
//                divided by the DC bin F[0][0] (real for real input; each workgroup re-sums column 0 of the intermediate in a
//                fixed order) and reduced to |.|, |.|^2 or the phase angle: mtf_from_psf is one launch pair with no elementwise
//                sweeps afterwards.
// Forward transforms only (the adjoints of the OTF routines transform complex gradients).
#pragma once
#include <hip/hip_runtime.h>

#include "fft_kernels.h"
#include "fft_r2c_types.h"

namespace pm {

// ---------------------------------------------------------------- row pass
// X[k] of one real row from its packed transform Z and the partner Z[(N2 - k) mod N2]
template <typename T>
PM_HD cx<T> r2c_combine(cx<T> z, cx<T> zp, cx<T> w, bool k0) {
    if (k0) return {z.x + z.y, z.x - z.y};      // X[0] = Re Z[0] + Im Z[0] and X[N/2] = Re Z[0] - Im Z[0], both real, share column 0
    // Xe = (z + conj zp) / 2, Xo = -i (z - conj zp) / 2
    const cx<T> xe = {T(0.5) * (z.x + zp.x), T(0.5) * (z.y - zp.y)};
    const cx<T> d = {T(0.5) * (z.x - zp.x), T(0.5) * (z.y + zp.y)};
    return xe + cmul(w, mul_mi(d));
}

// W_N^k for the thread's bins k = t + m N/32: W_N^t (ONE table read, requested before the transform so its latency is hidden)
// times the constants W_32^m -- sixteen dependent table reads per thread after the transform cost the row pass ~10 %
template <typename T>
PM_HD cx<T> w32(int m) {
    constexpr double tab[16][2] = {{1.00000000000000000000e+00, -0.00000000000000000000e+00}, {9.80785280403230430579e-01, -1.95090322016128248084e-01}, {9.23879532511286738483e-01, -3.82683432365089781779e-01}, {8.31469612302545235671e-01, -5.55570233019602177649e-01}, {7.07106781186547572737e-01, -7.07106781186547461715e-01}, {5.55570233019602288671e-01, -8.31469612302545235671e-01}, {3.82683432365089837290e-01, -9.23879532511286738483e-01}, {1.95090322016128331351e-01, -9.80785280403230430579e-01}, {6.12323399573676603587e-17, -1.00000000000000000000e+00}, {-1.95090322016128192573e-01, -9.80785280403230430579e-01}, {-3.82683432365089726268e-01, -9.23879532511286738483e-01}, {-5.55570233019601955604e-01, -8.31469612302545457716e-01}, {-7.07106781186547461715e-01, -7.07106781186547572737e-01}, {-8.31469612302545346694e-01, -5.55570233019602177649e-01}, {-9.23879532511286738483e-01, -3.82683432365089892802e-01}, {-9.80785280403230430579e-01, -1.95090322016128608906e-01}};
    return {T(tab[m][0]), T(tab[m][1])};
}

template <typename C, typename L, bool FOLD>
__global__ void __launch_bounds__(C::NT) fft_row_r2c_kernel(const L lp, const R2CRowStore<typename C::T> sp,
                                                            const cx<typename C::T>* __restrict__ tw, const int log_g_packed) {
    using T = typename C::T;
    const int log_g = engine_stagger(log_g_packed);
    static_assert(C::COMP == 1 && C::CI == 1, "row mode, complex exchange");
    static_assert(!FOLD || C::E == 2, "the fold pairs the two rows of a thread");
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(pm_smem);
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = group_remap(blockIdx.x, gridDim.x, log_g);
    static_assert(C::P == 16, "rows of at least 32 samples");
    cx<T> v[C::E][C::P];
    const cx<T> wt = sp.twn[pos.t];
    load<C>(lp, unit, pos, v);
    if constexpr (C::E == 2 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else fft_run<C>(v, pos, pm_smem, tw);
    constexpr int N2 = C::N;          // complex points per row = N / 2
    // partner exchange, ONE slot e at a time through the same LDS region: the sequences of the workgroup in natural order, then each
    // thread combines its bins with Z[(N2 - k) mod N2] in place.  (Both slots at once doubled the kernel's LDS -- 69.6 KiB at 2048
    // complex points with two rows per thread: two 256-thread workgroups per CU instead of four.)
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < C::P; ++m) lds[lds_addr<C>(pos.bo, 0, pos.t + m * C::TPS)] = v[e][m];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const int k = pos.t + m * C::TPS;
            const cx<T> zp = lds[lds_addr<C>(pos.bo, 0, (N2 - k) & (N2 - 1))];
            v[e][m] = r2c_combine(v[e][m], zp, cmul(wt, w32<T>(m)), k == 0);
        }
    }
    // stores: the lean addressing of the tiled intermediate (fft_io.h TiledRowAddr)
    if constexpr (FOLD) {
        const int i = unit * C::BO + pos.bo;       // pair index = logical row of the lower half
        if (i >= sp.nseq) return;
        cx<T> wi = sp.twm[i];
        if (sp.swap) wi = {-wi.x, -wi.y};          // rows rotated by M/2: the difference changes sign (store(RowStoreFold))
        const TiledRowAddr<C> A(pos.t, i, sp.log_tc, sp.nseq);
        cx<T>* const d1 = sp.dst + sp.plane_stride;
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const cx<T> x0 = v[0][m], x1 = v[1][m];
            *A.at(sp.dst, m) = x0 + x1;
            *A.at(d1, m) = cmul(x0 - x1, wi);
            if ((m & 3) == 3) PM_SCHED_FENCE();
        }
    } else {
#pragma unroll
        for (int e = 0; e < C::E; ++e) {
            const int seq = (unit * C::BO + pos.bo) * C::E + e;
            if (seq >= sp.nseq) continue;
            const TiledRowAddr<C> A(pos.t, seq, sp.log_tc, sp.nseq);
#pragma unroll
            for (int m = 0; m < C::P; ++m) {
                *A.at(sp.dst, m) = v[e][m];
                if ((m & 3) == 3) PM_SCHED_FENCE();
            }
        }
    }
}

// ---------------------------------------------------------------- column pass
// real value of the epilogue EPI for x (mirror: for conj(x))
template <typename T, int EPI>
PM_HD T herm_real(cx<T> x, bool mirror) {
    if constexpr (EPI == EPI_ARG) {
        const T a = atan2(x.y, x.x);
        return mirror ? -a : a;
    } else {
        const T i2 = x.x * x.x + x.y * x.y;
        if constexpr (EPI == EPI_ABS) return sqrt(i2);
        else return i2;
    }
}

// one result through the epilogue; `mirror`: the conjugate image (per-element path: the thread that owns column 0, odd views)
template <typename T, int EPI>
PM_HD void herm_put(const HermStore<T>& p, int qy, int qx, cx<T> x, bool mirror) {
    if (qy < 0 || qx < 0) return;
    const int64_t at = int64_t(qy) * p.ld + qx;
    if constexpr (EPI == EPI_NONE) {
        if (mirror) x.y = -x.y;
        reinterpret_cast<cx<T>*>(p.dst)[at] = x;
    } else {
        reinterpret_cast<T*>(p.dst)[at] = herm_real<T, EPI>(x, mirror);
    }
}

// 8 / 16-byte stores at addresses that are only 4 / 8-byte aligned (the mirrored pair of columns starts one element off the
// natural boundary): the hardware takes them, the compiler would split them into single-dword stores.  The trailing s_nop covers
// the "VMEM store of more than 8 bytes followed by a write of its data registers" hazard, which hipcc cannot see through the asm.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void store_pair_unaligned(float* a, float x, float y) {
    typedef float V2 __attribute__((ext_vector_type(2)));
    const V2 w = {x, y};
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" : : "v"(a), "v"(w) : "memory");
}
__device__ __forceinline__ void store_pair_unaligned(double* a, double x, double y) {
    typedef double V2 __attribute__((ext_vector_type(2)));
    const V2 w = {x, y};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(a), "v"(w) : "memory");
}
__device__ __forceinline__ void store_quad_unaligned(float* a, float x, float y, float z, float u) {
    typedef float V4 __attribute__((ext_vector_type(4)));
    const V4 w = {x, y, z, u};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(a), "v"(w) : "memory");
}
#else
template <typename T> inline void store_pair_unaligned(T* a, T x, T y) { a[0] = x; a[1] = y; }
inline void store_quad_unaligned(float* a, float x, float y, float z, float u) { a[0] = x; a[1] = y; a[2] = z; a[3] = u; }
#endif

// Fast store: unwindowed output, rotations by 0 or half a length on both axes, the thread's columns strictly inside (0, N/2).
// Memory positions follow from the rotation alone -- a bin at position p has its conjugate image at (n - p) mod n on either axis,
// because twice the rotation is a whole turn -- so there is no index map per element.  The direct pair of columns goes out as one
// 16-byte (complex64) store, the mirrored pair (in reversed order) as one 16-byte store that starts on an 8-byte boundary.
template <typename C, int ROT, int EPI>
PM_HD void herm_store_fast(const HermStore<typename C::T>& p, int col0, ThreadPos pos, const cx<typename C::T> (&v)[C::E][C::P], typename C::T s) {
    using T = typename C::T;
    int qx0 = col0 + p.ax.shift;
    if (qx0 >= p.N) qx0 -= p.N;
    const int qm = p.N - qx0 - (C::E - 1);          // first (lowest) mirrored position: the image of the thread's LAST column
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int pp = slot_pos<C, ROT>(pos.t, m, p.ay.shift);
        // image row: (M - pp) mod M; in the plane of the odd bins of a folded transform (rows 2 pp + 1 of the output) M - 1 - pp
        const int pm_ = p.plane == 1 ? p.M - 1 - pp : (pp == 0 ? 0 : p.M - pp);
        cx<T> x[C::E];
#pragma unroll
        for (int e = 0; e < C::E; ++e) x[e] = cscale(v[e][m], s);
        if constexpr (EPI == EPI_NONE) {
            cx<T>* a = reinterpret_cast<cx<T>*>(p.dst) + int64_t(pp) * p.ld + qx0;
            cx<T>* b = reinterpret_cast<cx<T>*>(p.dst) + int64_t(pm_) * p.ld + qm;
            if constexpr (C::E == 2 && sizeof(T) == 4) {
                *reinterpret_cast<Vec4<T>*>(a) = Vec4<T>{x[0].x, x[0].y, x[1].x, x[1].y};
                store_quad_unaligned(reinterpret_cast<T*>(b), x[1].x, -x[1].y, x[0].x, -x[0].y);
            } else {
#pragma unroll
                for (int e = 0; e < C::E; ++e) {
                    a[e] = x[e];
                    b[C::E - 1 - e] = cx<T>{x[e].x, -x[e].y};
                }
            }
        } else {
            T* a = reinterpret_cast<T*>(p.dst) + int64_t(pp) * p.ld + qx0;
            T* b = reinterpret_cast<T*>(p.dst) + int64_t(pm_) * p.ld + qm;
            if constexpr (C::E == 2) {
                const T r0 = herm_real<T, EPI>(x[0], false), r1 = herm_real<T, EPI>(x[1], false);
                *reinterpret_cast<cx<T>*>(a) = cx<T>{r0, r1};                         // a pair of reals, 8-byte aligned
                store_pair_unaligned(b, EPI == EPI_ARG ? -r1 : r1, EPI == EPI_ARG ? -r0 : r0);
            } else {
                const T r0 = herm_real<T, EPI>(x[0], false);
                a[0] = r0;
                b[0] = EPI == EPI_ARG ? -r0 : r0;
            }
        }
    }
}

// LDS of the Hermitian column kernel: [exchange fabric of the transform | partner exchange of the packed column (same region)] then, 16-byte
// aligned behind both, one double per wave for the DC sum
template <typename C>
constexpr size_t herm_dc_offset() {
    constexpr size_t part = size_t(C::BO) * C::N * sizeof(cx<typename C::T>);
    constexpr size_t need = C::LDS_BYTES > part ? C::LDS_BYTES : part;
    return (need + 15) & ~size_t(15);
}

template <typename C, int EPI>
__global__ void __launch_bounds__(C::NT) fft_col_herm_kernel(const ColLoadTiled<typename C::T> lp0, const HermStore<typename C::T> sp0,
                                                            const cx<typename C::T>* __restrict__ tw, const int log_g_packed) {
    using T = typename C::T;
    const int log_g = engine_stagger(log_g_packed);
    constexpr int TC = C::CI * C::E;
    constexpr size_t DC_OFF = herm_dc_offset<C>();
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = group_remap(blockIdx.x, gridDim.x, log_g) * C::BO + pos.bo;
    cx<T> v[C::E][C::P];
    const auto lp = at_batch(lp0, blockIdx.y);       // blockIdx.y: plane of a folded transform
    HermStore<T> sp = sp0;
    if (sp.plane >= 0) {
        sp.plane = blockIdx.y;
        if constexpr (EPI == EPI_NONE) sp.dst = reinterpret_cast<cx<T>*>(sp.dst) + int64_t(blockIdx.y) * sp.plane_dst;
        else sp.dst = reinterpret_cast<T*>(sp.dst) + int64_t(blockIdx.y) * sp.plane_dst;
    }
    load<C>(lp, unit, pos, v);
    T s = sp.scale;
    // F[0][0] = sum over rows of X_row[0] (real), the same value, bit for bit, in every workgroup (strided partial sums per thread,
    // each wave reduced in registers in a fixed order, the wave sums through LDS).  Round 5: the wave sums go to a slot BEHIND the
    // exchange fabric before the transform and are read after it -- the transform's own barriers publish them, where the
    // reduction used to sit between the loads and the transform with two barriers of its own (2.1 of the pass's 39 us,
    // experiments/wave_spec/ws_bench.hip).
    constexpr int NW = (C::NT + 63) / 64;
    double* const red = reinterpret_cast<double*>(pm_smem + DC_OFF);
    if (sp.norm_dc) {
        double acc = 0.0;
        for (int q = threadIdx.x; q < sp.nrows_w; q += C::NT) acc += double(sp.w0[int64_t(q) * sp.w0_stride].x);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        if constexpr (C::NSTAGE < 2) __syncthreads();     // a single-stage transform has no barrier of its own
    }
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else fft_run<C>(v, pos, pm_smem, tw);
    if (sp.norm_dc) {
        double dc = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) dc += red[w];
        s = T(double(sp.scale) / dc);
    }
    const int n2 = sp.N / 2;
    const int col0 = unit * TC + pos.cl * C::E;
    // Column 0 carries X[0] + i X[N/2] of two real columns: its transform C separates as F0[u] = (C[u] + conj C[M - u]) / 2,
    // FN2[u] = (C[u] - conj C[M - u]) / (2i).  The partner C[M - u] comes through LDS, in the workgroup that owns tile 0 (with
    // several tiles per workgroup -- short columns -- every workgroup passes the barriers).
    // (the partners are read from LDS where they are used, slot by slot, not gathered into a 32-register array first: see
    // fft_col_mul_herm_kernel)
    cx<T>* const ex = reinterpret_cast<cx<T>*>(pm_smem);
    if (C::BO > 1 || unit == 0) {
        __syncthreads();
        if (col0 == 0) {
#pragma unroll
            for (int m = 0; m < C::P; ++m) ex[pos.bo * C::N + pos.t + m * C::TPS] = v[0][m];
        }
        __syncthreads();
    }
    const int rot = rot_of<C>(sp.ay.shift);
    // the thread's columns strictly inside (0, N/2): every bin has its image; the thread that owns column 0 takes the
    // per-element path below
    if (sp.fast && rot >= 0 && col0 > 0 && col0 + C::E - 1 < n2) {
        if (rot == 0) herm_store_fast<C, 0, EPI>(sp, col0, pos, v, s);
        else herm_store_fast<C, (C::P >= 2 ? C::P / 2 : 0), EPI>(sp, col0, pos, v, s);
        return;
    }
    int qx[C::E], qxm[C::E];
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        const int k = col0 + e;
        qx[e] = k < n2 ? sp.ax.map(k) : -1;
        qxm[e] = (k > 0 && k < n2) ? sp.ax.map(sp.N - k) : -1;
    }
    const int qxn = sp.ax.map(n2);
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int u = pos.t + m * C::TPS;
        const int qy = sp.ay.map(u);
        const int qym = sp.ay.map(sp.plane == 1 ? sp.M - 1 - u : (u == 0 ? 0 : sp.M - u));
#pragma unroll
        for (int e = 0; e < C::E; ++e) {
            const cx<T> x = cscale(v[e][m], s);
            if (col0 + e == 0) {
                // bin u within the plane; its partner: (M - u) mod M, or M - 1 - u among the odd bins
                const cx<T> xp = cscale(ex[pos.bo * C::N + (sp.plane == 1 ? C::N - 1 - u : ((C::N - u) & (C::N - 1)))], s);
                const cx<T> f0 = {T(0.5) * (x.x + xp.x), T(0.5) * (x.y - xp.y)};
                const cx<T> d = {T(0.5) * (x.x - xp.x), T(0.5) * (x.y + xp.y)};
                herm_put<T, EPI>(sp, qy, qx[e], f0, false);
                herm_put<T, EPI>(sp, qy, qxn, mul_mi(d), false);
            } else {
                herm_put<T, EPI>(sp, qy, qx[e], x, false);
                herm_put<T, EPI>(sp, qym, qxm[e], x, true);
            }
        }
    }
}

// ---------------------------------------------------------------- launchers (instantiated in fft_r2c_f32.hip / fft_r2c_f64.hip)
template <typename T, int LOGN2, int VAR, bool FOLD>
int launch_row_r2c_one(const RowLoadNat<T>& lp, const R2CRowStore<T>& sp, const cx<T>* tw, int nunits, int log_g, hipStream_t st) {
    using C = typename RowCfgSel<T, LOGN2, VAR>::type;
    auto kern = fft_row_r2c_kernel<C, RowLoadNat<T>, FOLD>;
    // the partner exchange keeps the sequences of the workgroup in LDS (one slot e at a time), also when the transform itself has a
    // single stage
    constexpr size_t part = size_t(C::LDS_ELEMS) * sizeof(cx<T>);
    constexpr size_t LDSB = C::LDS_BYTES > part ? C::LDS_BYTES : part;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int per_wg = C::BO * (FOLD ? 1 : C::E);    // units: rows, or row PAIRS of a folded transform
    const int grid = (nunits + per_wg - 1) / per_wg;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), LDSB, st, lp, sp, tw, engine_log_g(log_g, grid, LDSB, C::NT, 3));
    return int(hipGetLastError());
}

template <typename T>
int launch_row_r2c_impl(int logn2, const RowLoadNat<T>& lp, const R2CRowStore<T>& sp, const cx<T>* tw, int nunits, int log_g, hipStream_t st) {
    if (sp.fold) {      // two rows (i, i + M/2) per thread
        switch (logn2) {
#define PM_CASE(k) \
    case k:        \
        return launch_row_r2c_one<T, k, 4, true>(lp, sp, tw, nunits, log_g, st);
            PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASE(11)
#undef PM_CASE
            case 12:
                if constexpr (sizeof(T) == 4) return launch_row_r2c_one<T, 12, 4, true>(lp, sp, tw, nunits, log_g, st);
                else return -2;
            default:
                return -2;
        }
    }
    switch (logn2) {
#define PM_CASE(k) \
    case k:        \
        return launch_row_r2c_one<T, k, 0, false>(lp, sp, tw, nunits, log_g, st);
        PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10)
#undef PM_CASE
        case 11:
            if constexpr (sizeof(T) == 4) return launch_row_r2c_one<T, 11, 5, false>(lp, sp, tw, nunits, log_g, st);
            else return launch_row_r2c_one<T, 11, 0, false>(lp, sp, tw, nunits, log_g, st);
        case 12:
            if constexpr (sizeof(T) == 4) return launch_row_r2c_one<T, 12, 4, false>(lp, sp, tw, nunits, log_g, st);
            else return -2;     // complex128 rows of 4096 complex points exchange re / im separately (COMP = 2): not on this path
        default:
            return -2;
    }
}

template <typename T, int LOGM, int EPI, int VAR = 0>
int launch_col_herm_epi(const ColLoadTiled<T>& lp, const HermStore<T>& sp, const cx<T>* tw, int ntiles, int log_g, hipStream_t st) {
    using C = typename ColCfgSel<T, LOGM, VAR>::type;
    auto kern = fft_col_herm_kernel<C, EPI>;
    constexpr size_t LDSB = herm_dc_offset<C>() + size_t((C::NT + 63) / 64) * sizeof(double);     // fabric / partner exchange, then the wave sums
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int grid = (ntiles + C::BO - 1) / C::BO;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid, sp.plane >= 0 ? 2 : 1), dim3(C::NT), LDSB, st, lp, sp, tw, engine_log_g(log_g, grid * (sp.plane >= 0 ? 2 : 1), LDSB, C::NT, 4));
    return int(hipGetLastError());
}

template <typename T, int LOGM>
int launch_col_herm_one(const ColLoadTiled<T>& lp, const HermStore<T>& sp, const cx<T>* tw, int ntiles, int log_g, hipStream_t st) {
    if constexpr (LOGM == 11) {
        if (sp.wide) {
            switch (sp.epilogue) {
                case EPI_NONE: return launch_col_herm_epi<T, LOGM, EPI_NONE, 2>(lp, sp, tw, ntiles, log_g, st);
                case EPI_ABS2: return launch_col_herm_epi<T, LOGM, EPI_ABS2, 2>(lp, sp, tw, ntiles, log_g, st);
                case EPI_ABS: return launch_col_herm_epi<T, LOGM, EPI_ABS, 2>(lp, sp, tw, ntiles, log_g, st);
                case EPI_ARG: return launch_col_herm_epi<T, LOGM, EPI_ARG, 2>(lp, sp, tw, ntiles, log_g, st);
                default: return -2;
            }
        }
    }
    switch (sp.epilogue) {      // one kernel per epilogue: the phase angle alone (atan2, 32 times per thread) is most of a kernel's code
        case EPI_NONE: return launch_col_herm_epi<T, LOGM, EPI_NONE>(lp, sp, tw, ntiles, log_g, st);
        case EPI_ABS2: return launch_col_herm_epi<T, LOGM, EPI_ABS2>(lp, sp, tw, ntiles, log_g, st);
        case EPI_ABS: return launch_col_herm_epi<T, LOGM, EPI_ABS>(lp, sp, tw, ntiles, log_g, st);
        case EPI_ARG: return launch_col_herm_epi<T, LOGM, EPI_ARG>(lp, sp, tw, ntiles, log_g, st);
        default: return -2;
    }
}

template <typename T>
int launch_col_herm_impl(int logm, const ColLoadTiled<T>& lp, const HermStore<T>& sp, const cx<T>* tw, int ntiles, int log_g, hipStream_t st) {
    switch (logm) {
#define PM_CASE(k) \
    case k:        \
        return launch_col_herm_one<T, k>(lp, sp, tw, ntiles, log_g, st);
        PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASE(11) PM_CASE(12) PM_CASE(13)
#undef PM_CASE
        default:
            return -2;
    }
}

}  // namespace pm
