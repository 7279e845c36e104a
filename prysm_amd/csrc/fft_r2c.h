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
template <typename C, typename L>
__global__ void __launch_bounds__(C::NT) fft_row_r2c_kernel(const L lp, const R2CRowStore<typename C::T> sp,
                                                            const cx<typename C::T>* __restrict__ tw, const int log_g) {
    using T = typename C::T;
    static_assert(C::COMP == 1 && C::CI == 1, "row mode, complex exchange");
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(pm_smem);
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = group_remap(blockIdx.x, gridDim.x, log_g);
    cx<T> v[C::E][C::P];
    load<C>(lp, unit, pos, v);
    if constexpr (C::E == 2 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else fft_run<C>(v, pos, pm_smem, tw);
    constexpr int N2 = C::N;          // complex points per row = N / 2
    const int tcm = (1 << sp.log_tc) - 1;
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        // partner exchange: everybody's Z in natural order, then read Z[(N2 - k) mod N2]
        __syncthreads();
#pragma unroll
        for (int m = 0; m < C::P; ++m) lds[lds_addr<C>(pos.bo, 0, pos.t + m * C::TPS)] = v[e][m];
        __syncthreads();
        const int seq = (unit * C::BO + pos.bo) * C::E + e;
        const bool ok = seq < sp.nseq;
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const int k = pos.t + m * C::TPS;
            const cx<T> z = v[e][m];
            const cx<T> zp = lds[lds_addr<C>(pos.bo, 0, (N2 - k) & (N2 - 1))];
            // Xe = (z + conj zp) / 2, Xo = -i (z - conj zp) / 2
            const cx<T> xe = {T(0.5) * (z.x + zp.x), T(0.5) * (z.y - zp.y)};
            const cx<T> d = {T(0.5) * (z.x - zp.x), T(0.5) * (z.y + zp.y)};
            const cx<T> xo = mul_mi(d);
            const cx<T> w = sp.twn[k];
            const cx<T> x = xe + cmul(w, xo);
            if (ok) {
                const int64_t a = ((int64_t(k >> sp.log_tc) * sp.nseq + seq) << sp.log_tc) + (k & tcm);
                // k = 0: X[0] = Re Z[0] + Im Z[0] and X[N/2] = Re Z[0] - Im Z[0], both real, share column 0
                sp.dst[a] = k == 0 ? cx<T>{z.x + z.y, z.x - z.y} : x;
            }
        }
    }
}

// ---------------------------------------------------------------- column pass
// one result through the epilogue; `mirror`: the conjugate image
template <typename T>
PM_HD void herm_put(const HermStore<T>& p, int qy, int qx, cx<T> x, bool mirror) {
    if (qy < 0 || qx < 0) return;
    if (mirror) x.y = -x.y;
    const int64_t at = int64_t(qy) * p.ld + qx;
    if (p.epilogue == EPI_NONE) {
        reinterpret_cast<cx<T>*>(p.dst)[at] = x;
        return;
    }
    T* o = reinterpret_cast<T*>(p.dst) + at;
    if (p.epilogue == EPI_ARG) {
        *o = atan2(x.y, x.x);
        return;
    }
    const T i2 = x.x * x.x + x.y * x.y;
    if (p.epilogue == EPI_ABS2) *o = i2;
    else if (p.epilogue == EPI_ABS) *o = sqrt(i2);
    else *o += p.weight * i2;
}

// real value of the epilogue EPI for x, and for conj(x)
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
        const int pm_ = pp == 0 ? 0 : p.M - pp;
        cx<T> x[C::E];
#pragma unroll
        for (int e = 0; e < C::E; ++e) x[e] = cscale(v[e][m], s);
        if constexpr (EPI == EPI_NONE) {
            cx<T>* a = reinterpret_cast<cx<T>*>(p.dst) + int64_t(pp) * p.ld + qx0;
            cx<T>* b = reinterpret_cast<cx<T>*>(p.dst) + int64_t(pm_) * p.ld + qm;
            if constexpr (C::E == 2 && sizeof(T) == 4) {
                *reinterpret_cast<Vec4<T>*>(a) = Vec4<T>{x[0].x, x[0].y, x[1].x, x[1].y};
                typedef T V4 __attribute__((ext_vector_type(4), aligned(8)));
                *reinterpret_cast<V4*>(b) = V4{x[1].x, -x[1].y, x[0].x, -x[0].y};
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
            if constexpr (EPI == EPI_ABS2_ACCUM) {
#pragma unroll
                for (int e = 0; e < C::E; ++e) {
                    const T i2 = herm_real<T, EPI_ABS2>(x[e], false);
                    a[e] += p.weight * i2;
                    b[C::E - 1 - e] += p.weight * i2;
                }
            } else if constexpr (C::E == 2) {
                const T r0 = herm_real<T, EPI>(x[0], false), r1 = herm_real<T, EPI>(x[1], false);
                *reinterpret_cast<cx<T>*>(a) = cx<T>{r0, r1};                         // a pair of reals, 8-byte aligned
                typedef T V2 __attribute__((ext_vector_type(2), aligned(4)));
                *reinterpret_cast<V2*>(b) = V2{EPI == EPI_ARG ? -r1 : r1, EPI == EPI_ARG ? -r0 : r0};
            } else {
                const T r0 = herm_real<T, EPI>(x[0], false);
                a[0] = r0;
                b[0] = EPI == EPI_ARG ? -r0 : r0;
            }
        }
    }
}

template <typename C, int ROT>
PM_HD void herm_store_fast_epi(const HermStore<typename C::T>& p, int col0, ThreadPos pos, const cx<typename C::T> (&v)[C::E][C::P], typename C::T s) {
    switch (p.epilogue) {
        case EPI_NONE: herm_store_fast<C, ROT, EPI_NONE>(p, col0, pos, v, s); break;
        case EPI_ABS2: herm_store_fast<C, ROT, EPI_ABS2>(p, col0, pos, v, s); break;
        case EPI_ABS2_ACCUM: herm_store_fast<C, ROT, EPI_ABS2_ACCUM>(p, col0, pos, v, s); break;
        case EPI_ABS: herm_store_fast<C, ROT, EPI_ABS>(p, col0, pos, v, s); break;
        default: herm_store_fast<C, ROT, EPI_ARG>(p, col0, pos, v, s); break;
    }
}

template <typename C>
__global__ void __launch_bounds__(C::NT) fft_col_herm_kernel(const ColLoadTiled<typename C::T> lp, const HermStore<typename C::T> sp,
                                                            const cx<typename C::T>* __restrict__ tw, const int log_g) {
    using T = typename C::T;
    constexpr int TC = C::CI * C::E;
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = group_remap(blockIdx.x, gridDim.x, log_g) * C::BO + pos.bo;
    cx<T> v[C::E][C::P];
    load<C>(lp, unit, pos, v);
    T s = sp.scale;
    if (sp.norm_dc) {
        // F[0][0] = sum over rows of X_row[0] (real): partial sums per thread, then a fixed-order tree through LDS -- the same
        // value, bit for bit, in every workgroup
        double* red = reinterpret_cast<double*>(pm_smem);
        double acc = 0.0;
        for (int q = threadIdx.x; q < sp.nrows_w; q += C::NT) acc += double(sp.w0[int64_t(q) * sp.w0_stride].x);
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int half = C::NT / 2; half > 0; half >>= 1) {
            if (int(threadIdx.x) < half) red[threadIdx.x] += red[threadIdx.x + half];
            __syncthreads();
        }
        const double dc = red[0];
        __syncthreads();     // the exchange of the transform reuses this LDS
        s = T(double(sp.scale) / dc);
    }
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else fft_run<C>(v, pos, pm_smem, tw);
    const int n2 = sp.N / 2;
    const int col0 = unit * TC + pos.cl * C::E;
    // Column 0 carries X[0] + i X[N/2] of two real columns: its transform C separates as F0[u] = (C[u] + conj C[M - u]) / 2,
    // FN2[u] = (C[u] - conj C[M - u]) / (2i).  The partner C[M - u] comes through LDS, in the workgroup that owns tile 0 (with
    // several tiles per workgroup -- short columns -- every workgroup passes the barriers).
    cx<T> part[C::P];
    if (C::BO > 1 || unit == 0) {
        cx<T>* ex = reinterpret_cast<cx<T>*>(pm_smem);
        __syncthreads();
        if (col0 == 0) {
#pragma unroll
            for (int m = 0; m < C::P; ++m) ex[pos.bo * C::N + pos.t + m * C::TPS] = v[0][m];
        }
        __syncthreads();
        if (col0 == 0) {
#pragma unroll
            for (int m = 0; m < C::P; ++m) part[m] = ex[pos.bo * C::N + ((C::N - pos.t - m * C::TPS) & (C::N - 1))];
        }
    }
    const int rot = rot_of<C>(sp.ay.shift);
    // the thread's columns strictly inside (0, N/2): every bin has its image; the thread that owns column 0 takes the
    // per-element path below
    if (sp.fast && rot >= 0 && col0 > 0 && col0 + C::E - 1 < n2) {
        if (rot == 0) herm_store_fast_epi<C, 0>(sp, col0, pos, v, s);
        else herm_store_fast_epi<C, (C::P >= 2 ? C::P / 2 : 0)>(sp, col0, pos, v, s);
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
        const int qym = sp.ay.map(u == 0 ? 0 : sp.M - u);
#pragma unroll
        for (int e = 0; e < C::E; ++e) {
            const cx<T> x = cscale(v[e][m], s);
            if (col0 + e == 0) {
                const cx<T> xp = cscale(part[m], s);
                const cx<T> f0 = {T(0.5) * (x.x + xp.x), T(0.5) * (x.y - xp.y)};
                const cx<T> d = {T(0.5) * (x.x - xp.x), T(0.5) * (x.y + xp.y)};
                herm_put(sp, qy, qx[e], f0, false);
                herm_put(sp, qy, qxn, mul_mi(d), false);
            } else {
                herm_put(sp, qy, qx[e], x, false);
                herm_put(sp, qym, qxm[e], x, true);
            }
        }
    }
}

// ---------------------------------------------------------------- launchers (instantiated in fft_r2c_f32.hip / fft_r2c_f64.hip)
template <typename T, int LOGN2, int VAR>
int launch_row_r2c_one(const RowLoadNat<T>& lp, const R2CRowStore<T>& sp, const cx<T>* tw, int nseq, int log_g, hipStream_t st) {
    using C = typename RowCfgSel<T, LOGN2, VAR>::type;
    auto kern = fft_row_r2c_kernel<C, RowLoadNat<T>>;
    // the partner exchange needs one whole sequence per row group in LDS even when the transform itself has a single stage
    constexpr size_t LDSB = size_t(C::LDS_ELEMS) * sizeof(cx<T>);
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int per_wg = C::BO * C::E;
    const int grid = (nseq + per_wg - 1) / per_wg;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), LDSB, st, lp, sp, tw, log_g);
    return int(hipGetLastError());
}

template <typename T>
int launch_row_r2c_impl(int logn2, const RowLoadNat<T>& lp, const R2CRowStore<T>& sp, const cx<T>* tw, int nseq, int log_g, hipStream_t st) {
    switch (logn2) {
#define PM_CASE(k) \
    case k:        \
        return launch_row_r2c_one<T, k, 0>(lp, sp, tw, nseq, log_g, st);
        PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10)
#undef PM_CASE
        case 11:
            if constexpr (sizeof(T) == 4) return launch_row_r2c_one<T, 11, 5>(lp, sp, tw, nseq, log_g, st);
            else return launch_row_r2c_one<T, 11, 0>(lp, sp, tw, nseq, log_g, st);
        case 12:
            if constexpr (sizeof(T) == 4) return launch_row_r2c_one<T, 12, 4>(lp, sp, tw, nseq, log_g, st);
            else return -2;     // complex128 rows of 4096 complex points exchange re / im separately (COMP = 2): not on this path
        default:
            return -2;
    }
}

template <typename T, int LOGM>
int launch_col_herm_one(const ColLoadTiled<T>& lp, const HermStore<T>& sp, const cx<T>* tw, int ntiles, int log_g, hipStream_t st) {
    using C = typename ColCfgSel<T, LOGM, 0>::type;
    auto kern = fft_col_herm_kernel<C>;
    constexpr size_t red = size_t(C::NT) * sizeof(double);                  // the DC reduction
    constexpr size_t part = size_t(C::BO) * C::N * sizeof(cx<T>);           // the partner exchange of the packed column
    constexpr size_t need = red > part ? red : part;
    constexpr size_t LDSB = C::LDS_BYTES > need ? C::LDS_BYTES : need;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int grid = (ntiles + C::BO - 1) / C::BO;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), LDSB, st, lp, sp, tw, log_g);
    return int(hipGetLastError());
}

template <typename T>
int launch_col_herm_impl(int logm, const ColLoadTiled<T>& lp, const HermStore<T>& sp, const cx<T>* tw, int ntiles, int log_g, hipStream_t st) {
    switch (logm) {
#define PM_CASE(k) \
    case k:        \
        return launch_col_herm_one<T, k>(lp, sp, tw, ntiles, log_g, st);
        PM_CASE(1) PM_CASE(2) PM_CASE(3) PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7)
        PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASE(11) PM_CASE(12) PM_CASE(13)
#undef PM_CASE
        default:
            return -2;
    }
}

}  // namespace pm
