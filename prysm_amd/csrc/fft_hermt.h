// Real-input 2-D transform on half the spectrum, TRANSPOSED form (round 6): the Hermitian symmetry is carried by the COLUMN axis.
// fft2 of a real PSF / object (prysm/otf.py:28-33 transform_psf, :62-135 the centre-normalised MTF / PTF / OTF).
//
// fft_r2c.h keeps the half spectrum along x: its column pass stores every result twice, and the mirror image of a tile's columns
// [16 j, 16 j + 16) is [N - 16 j - 15, N - 16 j] -- one element off a 64 B line whichever way the tiles are cut, so half of the pass's
// stores are partial sectors written by two different workgroups (WRITE_SIZE 1.34x the algorithmic bytes, the pass at 0.375 of the HBM
// peak; profiles/r05/pmc_mtf_summary.txt).  Here the mirror image of a ROW is a row:
//
//   pass A (columns): the real M x N array IS an M x N/2 complex array (pairs of adjacent samples, no copy); an M-point column
//                transform Z_c of packed column c, then (partner exchange through LDS, u <-> M - u inside the column)
//                    X_2c[u] = (Z_c[u] + conj Z_c[M - u]) / 2,     X_2c+1[u] = (Z_c[u] - conj Z_c[M - u]) / (2i)
//                are the column spectra of the two real columns.  Only u <= M/2 is kept: rows u = 0 .. M/2 - 1 of a natural
//                (row-major) M/2 x N complex intermediate -- a thread's two packed columns are four adjacent elements of a row, a
//                workgroup tile writes whole 128 B lines -- with X[0] and X[M/2], both real, sharing row 0 as X[0] + i X[M/2].
//   pass B (rows): ordinary N-point row transforms of those M/2 rows (two rows per thread where the complex path pairs rows too);
//                row u goes out twice, F[u][k] at (u, k) and conj F[u][k] at (M - u, N - k): BOTH are complete rows written by one
//                workgroup in whole lines (consecutive lanes hold consecutive k, the image runs backwards through the same lines).
//                Row 0 separates into F[0][.] and F[M/2][.] (partner exchange k <-> N - k, in the workgroup that owns it).  The
//                centre normalisation (F[0][0] = the sum of row 0's real parts, re-summed by every workgroup in a fixed order: the
//                same bits everywhere) and |.|, |.|^2 or the phase angle ride in the store.
// Input rotations by half a length (ifftshift) are signs: (-1)^u applied in pass A, (-1)^k in pass B.  Forward transforms only.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_kernels.h"
#include "fft_r2c.h"
#include "fft_hermt_types.h"

namespace pm {

// ---------------------------------------------------------------- pass A: columns of the packed real array
// partners Z[(N - u) mod N] of the thread's bins u = t + m TPS, m < P/2, of column slot e -- through the exchange fabric's LDS (whole
// complex values where the fabric exchanges complex values, else one real component at a time: the region is BO * N * CI values)
// `plane` (folded form): -1 none; 0 / 1: the even / odd bins of a length-2N column, whose partner of bin u' is (N - u') mod N / N - 1 - u'
template <typename C>
__device__ __forceinline__ void hermt_partners(const cx<typename C::T> (&v)[C::E][C::P], int e, ThreadPos pos, char* smem,
                                               cx<typename C::T> (&zp)[C::P / 2], int plane) {
    using T = typename C::T;
    const int base = pos.bo * C::N;
    // partner index of slot m: both forms are a per-thread term minus m TPS
    const int p0 = plane == 1 ? C::N - 1 - pos.t : C::N - pos.t;
    if constexpr (C::COMP == 1) {
        cx<T>* const ex = reinterpret_cast<cx<T>*>(smem);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < C::P; ++m) ex[(base + pos.t + m * C::TPS) * C::CI + pos.cl] = v[e][m];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < C::P / 2; ++m) zp[m] = ex[(base + ((p0 - m * C::TPS) & (C::N - 1))) * C::CI + pos.cl];
    } else {
        T* const ex = reinterpret_cast<T*>(smem);
#pragma unroll
        for (int comp = 0; comp < 2; ++comp) {
            __syncthreads();
#pragma unroll
            for (int m = 0; m < C::P; ++m) ex[(base + pos.t + m * C::TPS) * C::CI + pos.cl] = comp ? v[e][m].y : v[e][m].x;
            __syncthreads();
#pragma unroll
            for (int m = 0; m < C::P / 2; ++m) {
                const T r = ex[(base + ((p0 - m * C::TPS) & (C::N - 1))) * C::CI + pos.cl];
                if (comp) zp[m].y = r;
                else zp[m].x = r;
            }
        }
    }
}

// FOLD: C is the configuration of HALF the column length; the workgroup (tile, plane) loads rows r and r + C::N of its tile, takes the
// radix-2 decimation-in-frequency step of the 2 C::N-point transform for its plane -- even bins: y[r] + y[r + C::N]; odd bins:
// (y[r] - y[r + C::N]) W_2N^r -- and transforms C::N points.  The partners of a plane's bins are in the plane, so everything after the
// transform is as in the unfolded form with rows u = 2 u' + plane.  Both planes of a tile read the same 64 B pieces: they are dispatched
// back to back on ONE XCD (hermt_fold_unit), beside the planes of the adjacent tiles, and the second read is an L2 hit.  What it buys: tiles of half the height -- two workgroups
// per CU whose load / transform / store phases overlap, where the 4096-point tile (256 KiB, the registers of 1024 threads) runs alone:
// pass A of a 4096^2 fp32 MTF 44.5 us unfolded (profiles/r06/exp_herm_t.log).
PM_HD void hermt_fold_unit(int lin, int total, int log_g, int& tile, int& plane) {
    // units 2 tile + plane, 2^(log_g + 1) adjacent units back to back on one XCD: both planes of 2^log_g adjacent tiles
    const int u2 = group_remap(lin, total, log_g + 1);
    tile = u2 >> 1;
    plane = u2 & 1;
}

template <typename C, bool FOLD>
__global__ void __launch_bounds__(C::NT, (FOLD && sizeof(typename C::T) == 4 && C::NT == 512) ? 4 : 1)
fft_col_hermt_kernel(const ColLoadNat<typename C::T> lp, const HermTColStore<typename C::T> sp, const cx<typename C::T>* __restrict__ tw,
                     const int log_g_packed) {
    using T = typename C::T;
    static_assert(C::P == 16 && C::TPS >= 2, "columns of at least 32 samples");
    static_assert(!FOLD || C::BO == 1, "the fold is for tall columns: one tile per workgroup");
    const int log_g = engine_stagger(log_g_packed);
    constexpr int TC = C::CI * C::E, H = C::P / 2;
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    int unit, plane = -1;
    cx<T> v[C::E][C::P];
    if constexpr (FOLD) {
        hermt_fold_unit(blockIdx.x, gridDim.x, log_g, unit, plane);
        const int col0 = unit * TC + pos.cl * C::E;
        const cx<T>* const col = lp.src + col0;
        const int64_t half = int64_t(C::N) * lp.ld;
        cx<T> wt = {T(1), T(0)};
        if (plane) wt = sp.twm[pos.t];      // W_2N^(t + m TPS) = W_2N^t W_32^m  (TPS = 2N / 32)
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const cx<T>* a = col + int64_t(pos.t + m * C::TPS) * lp.ld;
            cx<T> lo[C::E], hi[C::E];
            bool done = false;
            if constexpr (C::E == 2 && sizeof(T) == 4) {
                if (lp.vec_ok) {
                    const Vec4<T> w0 = *reinterpret_cast<const Vec4<T>*>(a), w1 = *reinterpret_cast<const Vec4<T>*>(a + half);
                    lo[0] = {w0.a, w0.b}; lo[1] = {w0.c, w0.d};
                    hi[0] = {w1.a, w1.b}; hi[1] = {w1.c, w1.d};
                    done = true;
                }
            }
            if (!done) {
#pragma unroll
                for (int e = 0; e < C::E; ++e) {
                    lo[e] = a[e];
                    hi[e] = a[half + e];
                }
            }
            if (plane) {
                const cx<T> w = cmul(wt, w32<T>(m));
#pragma unroll
                for (int e = 0; e < C::E; ++e) v[e][m] = cmul(lo[e] - hi[e], w);
            } else {
#pragma unroll
                for (int e = 0; e < C::E; ++e) v[e][m] = lo[e] + hi[e];
            }
        }
    } else {
        unit = group_remap(blockIdx.x, gridDim.x, log_g) * C::BO + pos.bo;
        load<C>(lp, unit, pos, v);
    }
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else fft_run<C>(v, pos, pm_smem, tw);
    const int col0 = unit * TC + pos.cl * C::E;
    // (-1)^u of rows that came in rotated by M/2: u = t + m TPS has the parity of t (TPS is even), u = 2 u' + plane that of the plane; the
    // halves of the separation ride along
    const T hs = (sp.neg_odd && (FOLD ? plane == 1 : (pos.t & 1))) ? T(-0.5) : T(0.5);
    const int rmul = FOLD ? 2 : 1, radd = FOLD ? plane : 0;      // output row of bin u': rmul u' + radd
    // the separated spectra replace the column's values in place (slot m: the even column's bin, slot m + P/2: the odd column's): every
    // column slot of the thread is separated BEFORE anything is stored, so that the thread's 2 E adjacent results of a row go out together
    // (one column slot at a time they were 16 B pieces with 16 B gaps, the gaps filled an exchange later: WRITE_SIZE 133.7 MB for 67.1,
    // profiles/r06/pmc_mtf_summary_s13.txt)
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        cx<T> zp[H];
        hermt_partners<C>(v, e, pos, pm_smem, zp, plane);
        const cx<T> zh = v[e][H];
#pragma unroll
        for (int m = 0; m < H; ++m) {
            const cx<T> z = v[e][m];
            // A = X_even[u] = (z + conj zp) / 2,  B = X_odd[u] = -i (z - conj zp) / 2
            cx<T> a = {hs * (z.x + zp[m].x), hs * (z.y - zp[m].y)};
            cx<T> b = {hs * (z.y + zp[m].y), -hs * (z.x - zp[m].x)};
            if (m == 0 && pos.t == 0 && plane <= 0) {
                // u = 0 (its own partner): both real; the bins u = M/2 (slot P/2 of this thread, real too) share the row
                a = {z.x, zh.x};
                b = {z.y, zh.y};
            }
            v[e][m] = a;
            v[e][m + H] = b;
        }
    }
    if (col0 >= sp.ncols2) return;
    cx<T>* const out = sp.dst + int64_t(rmul * pos.t + radd) * sp.ld + 2 * col0;
#pragma unroll
    for (int m = 0; m < H; ++m) {
        cx<T>* const o = out + int64_t(m) * (rmul * C::TPS) * sp.ld;
#pragma unroll
        for (int e = 0; e < C::E; ++e) {
            if (col0 + e >= sp.ncols2) continue;
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<Vec4<T>*>(o + 2 * e) = Vec4<T>{v[e][m].x, v[e][m].y, v[e][m + H].x, v[e][m + H].y};
            } else {
                o[2 * e] = v[e][m];
                o[2 * e + 1] = v[e][m + H];
            }
        }
    }
}

// ---------------------------------------------------------------- pass B: rows of the half spectrum
template <typename T, int EPI>
__device__ __forceinline__ void hermt_put(void* dst, int64_t at, cx<T> x, bool mirror) {
    if constexpr (EPI == EPI_NONE) {
        if (mirror) x.y = -x.y;
        reinterpret_cast<cx<T>*>(dst)[at] = x;
    } else {
        reinterpret_cast<T*>(dst)[at] = herm_real<T, EPI>(x, mirror);
    }
}

// LDS of the row kernel: [exchange fabric | the packed row's partner exchange (same region)], then one double per wave for the DC sum
template <typename C>
constexpr size_t hermt_dc_offset() {
    constexpr size_t part = size_t(C::N) * sizeof(cx<typename C::T>);
    constexpr size_t need = C::LDS_BYTES > part ? C::LDS_BYTES : part;
    return (need + 15) & ~size_t(15);
}

template <typename C, int EPI, int MINW>
__global__ void __launch_bounds__(C::NT, MINW) fft_row_hermt_kernel(const RowLoadNat<typename C::T> lp, const HermTRowStore<typename C::T> sp,
                                                                   const cx<typename C::T>* __restrict__ tw, const int log_g_packed) {
    using T = typename C::T;
    static_assert(C::CI == 1 && C::P == 16, "row mode, rows of at least 32 points");
    const int log_g = engine_stagger(log_g_packed);
    constexpr size_t DC_OFF = hermt_dc_offset<C>();
    constexpr int NW = (C::NT + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = group_remap(blockIdx.x, gridDim.x, log_g);
    double* const red = reinterpret_cast<double*>(pm_smem + DC_OFF);
    if (sp.norm_dc) {
        // F[0][0] = sum_j X_j(0): strided partial sums per thread, each wave reduced in registers in a fixed order, the wave sums through
        // a slot behind the fabric -- published by the transform's own barriers (as in fft_r2c.h)
        double acc = 0.0;
        for (int q = threadIdx.x; q < C::N; q += C::NT) acc += double(sp.i0[q].x);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        if constexpr (C::NSTAGE < 2) __syncthreads();
    }
    cx<T> v[C::E][C::P];
    load<C>(lp, unit, pos, v);
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else fft_run<C>(v, pos, pm_smem, tw);
    T s = sp.scale;
    if (sp.norm_dc) {
        double dc = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) dc += red[w];
        s = T(double(sp.scale) / dc);
    }
    if (sp.neg_odd && (pos.t & 1)) s = -s;      // (-1)^k of columns that came in rotated by N/2: k = t + m TPS has the parity of t
    const int M = sp.M;
    constexpr int N = C::N;
    // positions along the row: direct (k + sx) mod N, image (N - k + sx) mod N; k = t + m TPS and sx is 0 or N/2 = 8 TPS, so both are
    // a uniform term plus / minus t
    if (unit == 0) {
        // row 0 = F[0][.] + i F[M/2][.] (two real rows transformed as one): F0[k] = (z + conj z')/2, Fh[k] = -i (z - conj z')/2 with
        // z' = Z[(N - k) mod N].  Fh goes out at once (row M/2, every k: no image needed), F0 stays in the registers as a complete row.
        cx<T>* const ex = reinterpret_cast<cx<T>*>(pm_smem);
        __syncthreads();
        if (pos.bo == 0) {
#pragma unroll
            for (int m = 0; m < C::P; ++m) ex[pos.t + m * C::TPS] = v[0][m];
        }
        __syncthreads();
        if (pos.bo == 0) {
            const int64_t rowh = int64_t((M / 2 + sp.sy) & (M - 1)) * sp.ld;
#pragma unroll
            for (int m = 0; m < C::P; ++m) {
                const int k = pos.t + m * C::TPS;
                const cx<T> z = v[0][m], zq = ex[(N - k) & (N - 1)];
                const cx<T> f0 = {T(0.5) * (z.x + zq.x), T(0.5) * (z.y - zq.y)};
                const cx<T> fh = {T(0.5) * (z.y + zq.y), T(-0.5) * (z.x - zq.x)};
                hermt_put<T, EPI>(sp.dst, rowh + ((k + sp.sx) & (N - 1)), cscale(fh, s), false);
                v[0][m] = f0;
            }
        }
    }
    // the centre sample of a normalised transform is data[c] / data[c] in the reference (prysm/otf.py:11-13, 36-74): exactly 1 -- the thread
    // that owns F[0][0] writes the scale itself, not F[0][0] times the reciprocal of the (re-summed, differently rounded) DC value
    const bool dc_here = sp.norm_dc && unit == 0 && threadIdx.x == 0;
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        const int u = (unit * C::BO + pos.bo) * C::E + e;
        if (u >= sp.nseq) continue;
        const int64_t rowd = int64_t((u + sp.sy) & (M - 1)) * sp.ld, rowm = int64_t((M - u + sp.sy) & (M - 1)) * sp.ld;
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            cx<T> x = cscale(v[e][m], s);
            if (e == 0 && m == 0 && dc_here) x = {sp.scale, T(0)};
            const int pd = pos.t + ((m * C::TPS + sp.sx) & (N - 1));
            hermt_put<T, EPI>(sp.dst, rowd + pd, x, false);
            if (u != 0) {
                const int pm_ = (N + sp.sx - m * C::TPS - pos.t) & (N - 1);
                hermt_put<T, EPI>(sp.dst, rowm + pm_, x, true);
            }
            if ((m & 3) == 3) PM_SCHED_FENCE();
        }
    }
}

// ---------------------------------------------------------------- launchers (instantiated in fft_hermt_f32.hip / fft_hermt_f64.hip)
template <typename T, int LOGM, bool FOLD>
int launch_col_hermt_one(const ColLoadNat<T>& lp, const HermTColStore<T>& sp, const cx<T>* tw, int ntiles, int log_g, hipStream_t st) {
    using C = typename ColCfgSel<T, FOLD ? LOGM - 1 : LOGM, 0>::type;
    if constexpr (FOLD && C::BO != 1) {
        return -2;
    } else {
        auto kern = fft_col_hermt_kernel<C, FOLD>;
        constexpr size_t LDSB = C::LDS_BYTES;
        if (LDSB > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
            if (e != hipSuccess) return int(e);
        }
        const int grid = FOLD ? 2 * ntiles : (ntiles + C::BO - 1) / C::BO;
        if (grid <= 0) return 0;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), LDSB, st, lp, sp, tw, engine_log_g(log_g, grid, LDSB, C::NT, 4));
        return int(hipGetLastError());
    }
}

// tw: the table of the transform the kernel runs -- M points, or M / 2 with sp.fold (sp.twm is then the M-point table)
template <typename T>
int launch_col_hermt_impl(int logm, const ColLoadNat<T>& lp, const HermTColStore<T>& sp, const cx<T>* tw, int ntiles, int log_g, hipStream_t st) {
    if (sp.fold) {
        switch (logm) {
#define PM_CASE(k) \
    case k:        \
        return launch_col_hermt_one<T, k, true>(lp, sp, tw, ntiles, log_g, st);
            PM_CASE(10) PM_CASE(11) PM_CASE(12) PM_CASE(13)
#undef PM_CASE
            default:
                return -2;
        }
    }
    switch (logm) {
#define PM_CASE(k) \
    case k:        \
        return launch_col_hermt_one<T, k, false>(lp, sp, tw, ntiles, log_g, st);
        PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASE(11) PM_CASE(12)
#undef PM_CASE
        default:
            return -2;
    }
}

template <typename T, int LOGN, int VAR, int EPI>
int launch_row_hermt_epi(const RowLoadNat<T>& lp, const HermTRowStore<T>& sp, const cx<T>* tw, int log_g, hipStream_t st) {
    using C = typename RowCfgSel<T, LOGN, VAR>::type;
    // the paired complex64 rows sit at the 128-register boundary of four workgroups per CU, like the complex path's (fft_kernel_min_waves)
    constexpr int MINW = (VAR == 4 && C::E == 2 && sizeof(T) == 4) ? 4 : 1;
    auto kern = fft_row_hermt_kernel<C, EPI, MINW>;
    constexpr size_t LDSB = hermt_dc_offset<C>() + size_t((C::NT + 63) / 64) * sizeof(double);
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int per_wg = C::BO * C::E;
    const int grid = (sp.nseq + per_wg - 1) / per_wg;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), LDSB, st, lp, sp, tw, engine_log_g(log_g, grid, LDSB, C::NT, 3));
    return int(hipGetLastError());
}

template <typename T, int LOGN, int VAR>
int launch_row_hermt_one(const RowLoadNat<T>& lp, const HermTRowStore<T>& sp, const cx<T>* tw, int log_g, hipStream_t st) {
    switch (sp.epilogue) {
        case EPI_NONE: return launch_row_hermt_epi<T, LOGN, VAR, EPI_NONE>(lp, sp, tw, log_g, st);
        case EPI_ABS2: return launch_row_hermt_epi<T, LOGN, VAR, EPI_ABS2>(lp, sp, tw, log_g, st);
        case EPI_ABS: return launch_row_hermt_epi<T, LOGN, VAR, EPI_ABS>(lp, sp, tw, log_g, st);
        case EPI_ARG: return launch_row_hermt_epi<T, LOGN, VAR, EPI_ARG>(lp, sp, tw, log_g, st);
        default: return -2;
    }
}

// one tiling per length and precision, the complex path's (pm_internal.h row_variant): `var` only tells 2048-point rows apart
template <typename T>
int launch_row_hermt_impl(int logn, int var, const RowLoadNat<T>& lp, const HermTRowStore<T>& sp, const cx<T>* tw, int log_g, hipStream_t st) {
    switch (logn) {
#define PM_CASE(k) \
    case k:        \
        return launch_row_hermt_one<T, k, 0>(lp, sp, tw, log_g, st);
        PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10)
#undef PM_CASE
        case 11:
            if constexpr (sizeof(T) == 4) return launch_row_hermt_one<T, 11, 5>(lp, sp, tw, log_g, st);
            else return launch_row_hermt_one<T, 11, 1>(lp, sp, tw, log_g, st);
        case 12:
            if constexpr (sizeof(T) == 4) return var == 0 ? launch_row_hermt_one<T, 12, 0>(lp, sp, tw, log_g, st) : launch_row_hermt_one<T, 12, 4>(lp, sp, tw, log_g, st);
            else return launch_row_hermt_one<T, 12, 0>(lp, sp, tw, log_g, st);
        case 13:
            if constexpr (sizeof(T) == 4) return launch_row_hermt_one<T, 13, 4>(lp, sp, tw, log_g, st);
            else return -2;
        default:
            return -2;
    }
}

}  // namespace pm
