// One axis of a chirp-Z transform (prysm/fttools.py:297-323, CZT.__call__ / .adjoint) as ONE kernel:
//
//     out[m] = scale * post[m] * IFFT_K( FFT_K( pad_K( pre . in ) ) . H )[out_off + m],   m < out_len
//
// i.e. chirp multiply, zero-padded forward transform of length K, multiply by the transformed chirp H, inverse transform, slice,
// chirp multiply.  The sequence of K points never leaves the registers of its workgroup (the engine returns natural order, so the
// inverse starts where the forward ended -- the same idea as fft_col_mul_kernel, the middle pass of the fused 2-D convolution):
// per axis the transform reads its input once and writes out_len results, where the composition of pm_fft1 and pm_scale_sep
// calls moved the K-point intermediate through memory four times.
//   row mode (axis 1): sequences are the rows; column mode (axis 0): the columns, a tile of 64 B rows per workgroup.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_kernels.h"
#include "fft_conv1_types.h"

namespace pm {

// v *= f (or conj f)
template <typename T>
PM_HD cx<T> mulc(cx<T> v, cx<T> f, int conj) {
    return conj ? cmulc(v, f) : cmul(v, f);
}

template <typename C, bool COL>
__global__ void __launch_bounds__(C::NT) fft_conv1_kernel(const Conv1<typename C::T> p, const cx<typename C::T>* __restrict__ tw) {
    using T = typename C::T;
    constexpr int TC = C::CI * C::E;
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = COL ? int(blockIdx.x) * C::BO + pos.bo : int(blockIdx.x);
    cx<T> v[C::E][C::P];
    // ---- load: element i of the length-K sequence is in[i - in_off] * pre[i - in_off] inside the window, zero outside
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int i = pos.t + m * C::TPS;
        const int q = i - p.in_off;
        const bool in = q >= 0 && q < p.in_len;
        cx<T> f = {T(1), T(0)};
        if (in && p.pre) f = p.pre[q];
#pragma unroll
        for (int e = 0; e < C::E; ++e) {
            // row mode: sequence = row (unit * BO + bo) * E + e, element q along the row; column mode: column tile * TC + cl * E + e
            const int seq = COL ? unit * TC + pos.cl * C::E + e : (unit * C::BO + pos.bo) * C::E + e;
            cx<T> x = {T(0), T(0)};
            if (in && seq < p.nseq) {
                x = COL ? p.in[int64_t(q) * p.in_ld + seq] : p.in[int64_t(seq) * p.in_ld + q];
                if (p.pre) x = mulc(x, f, p.pre_conj);
                if (p.single == 2) x.y = -x.y;      // inverse transform = conj-in / conj-out of the forward one
            }
            v[e][m] = x;
        }
    }
    // ---- forward transform, x H, inverse transform (conj-in / conj-out of the same engine transform)
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else fft_run<C>(v, pos, pm_smem, tw);
    if (!p.single) {       // uniform over the launch
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const cx<T> h = p.H[pos.t + m * C::TPS];
#pragma unroll
            for (int e = 0; e < C::E; ++e) {
                const cx<T> x = mulc(v[e][m], h, p.h_conj);
                v[e][m] = {x.x, -x.y};
            }
        }
        __syncthreads();   // LDS of the forward exchange is reused by the inverse
        // opaque copy of the slot: otherwise the twiddles (and their products) of the first transform are CSE'd with the second
        // and kept live across it (see fft_col_mul_kernel)
        ThreadPos pos2 = pos;
        asm volatile("" : "+v"(pos2.t), "+v"(pos2.cl), "+v"(pos2.bo));
        if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos2, pm_smem, tw);
        else fft_run<C>(v, pos2, pm_smem, tw);
    }
    const T osign = p.single == 1 ? T(1) : T(-1);      // conj-out of an inverse transform (the convolution's second, or single == 2)
    // ---- store the window [out_off, out_off + out_len) of the K results: conj (inverse), scale, x post
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int i = pos.t + m * C::TPS;
        const int q = i - p.out_off;
        if (q < 0 || q >= p.out_len) continue;
        cx<T> f = {p.scale, T(0)};
        if (p.post) {
            cx<T> w = p.post[q];
            if (p.post_conj) w.y = -w.y;
            f = cscale(w, p.scale);
        }
#pragma unroll
        for (int e = 0; e < C::E; ++e) {
            const int seq = COL ? unit * TC + pos.cl * C::E + e : (unit * C::BO + pos.bo) * C::E + e;
            if (seq >= p.nseq) continue;
            const cx<T> x = cmul(cx<T>{v[e][m].x, osign * v[e][m].y}, f);
            if (COL) p.out[int64_t(q) * p.out_ld + seq] = x;
            else p.out[int64_t(seq) * p.out_ld + q] = x;
        }
    }
}

template <typename T, bool COL, int LOGK, int VAR>
int launch_conv1_one(const Conv1<T>& p, const cx<T>* tw, hipStream_t st) {
    using Sel = typename std::conditional<COL, ColCfgSel<T, LOGK, 0>, RowCfgSel<T, LOGK, VAR>>::type;
    using C = typename Sel::type;
    auto kern = fft_conv1_kernel<C, COL>;
    constexpr size_t LDSB = C::LDS_BYTES;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int per_wg = COL ? C::BO * C::CI * C::E : C::BO * C::E;
    const int grid = (p.nseq + per_wg - 1) / per_wg;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), LDSB, st, p, tw);
    return int(hipGetLastError());
}

template <typename T, bool COL>
int launch_conv1_impl(int logk, const Conv1<T>& p, const cx<T>* tw, hipStream_t st) {
    switch (logk) {
#define PM_CASE(k) \
    case k:        \
        return launch_conv1_one<T, COL, k, 0>(p, tw, st);
        PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7) PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASE(11)
#undef PM_CASE
        case 12:   // rows of 4096 points: two rows per thread share pre / H / post (complex64)
            if constexpr (!COL && sizeof(T) == 4) return launch_conv1_one<T, COL, 12, 4>(p, tw, st);
            else return launch_conv1_one<T, COL, 12, 0>(p, tw, st);
        case 13:
            if constexpr (!COL && sizeof(T) == 4) return launch_conv1_one<T, COL, 13, 4>(p, tw, st);
            else return launch_conv1_one<T, COL, 13, 0>(p, tw, st);
        default:
            return -2;
    }
}

}  // namespace pm
