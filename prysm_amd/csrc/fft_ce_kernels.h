// Kernels and launchers of the composite register engine (fft_ce.h); the plans are instantiated per precision in fft_ce_f32.hip /
// fft_ce_f64.hip.  Entry points ce_rows / ce_cols (pm_internal.h) answer -1 when the length has no plan here or the view needs what these
// kernels do not do (real input, synthesis, multipliers, epilogues, mapped 1-D outputs): the caller then takes the general kernel.
#pragma once
#include "fft_ce.h"
#include "fft_mixed_kernels.h"

namespace pm {

// twiddles of the NEXT stage requested before the exchange that feeds it (their latency then hides behind two barriers)
template <typename C, int s>
__device__ __forceinline__ void ce_run(cx<typename C::T> (&v)[C::P], CePos pos, void* lds_raw, const cx<typename C::T>* __restrict__ tw) {
    using LT = typename CeLds<C>::type;
    LT* lds = reinterpret_cast<LT*>(lds_raw);
    if constexpr (s > 0 && !(C::ABL & 8)) {
#pragma unroll
        for (int comp = 0; comp < C::COMP; ++comp) {
            if (s > 1 || comp > 0) __syncthreads();     // the previous gather has finished
            ce_exch_write<C, s, LT>(v, comp, pos, lds);
            __syncthreads();
            ce_exch_read<C, s, LT>(v, comp, pos, lds);
        }
    }
    if constexpr (!(C::ABL & 4)) ce_stage<C, s>(v, pos.t, tw);
    if constexpr (s + 1 < C::PL::S) ce_run<C, s + 1>(v, pos, lds_raw, tw);
}

// waves per SIMD the kernel is compiled for: two workgroups per CU wherever two fit the LDS
template <typename C>
constexpr int ce_waves_per_eu() {
    if (C::WPE > 0) return C::WPE;
    constexpr int waves = (C::NT + 63) / 64;
    constexpr int by_lds = int(size_t(160) * 1024 / (C::LDS_BYTES ? C::LDS_BYTES : 1));
    constexpr int wgs = by_lds < 1 ? 1 : (by_lds > 4 ? 4 : by_lds);
    constexpr int need = (waves * wgs + 3) / 4;            // waves per SIMD with `wgs` workgroups resident
    constexpr int fit = 512 / (2 * C::P * int(sizeof(typename C::T)) / 4 + 44);     // ... that the data registers + ~40 allow
    return need < fit ? need : (fit < 1 ? 1 : fit);
}

// SYN 0: the rows are read; 2 / 3: synthesised from the OPD map + amplitude / from packed pairs (complex64 only: CeSynth)
template <typename C, bool WIN, int SYN = 0>
__global__ __launch_bounds__(C::NT, ce_waves_per_eu<C>()) void ce_rows_kernel(CeIn<typename C::T> in, CeRowOut<typename C::T> out,
                                                                               const cx<typename C::T>* __restrict__ tw, CeSynth sy) {
    using T = typename C::T;
    extern __shared__ __align__(16) char ce_smem[];
    const CePos pos = ce_pos<C>(threadIdx.x);
    const int row0 = int(blockIdx.x) * C::SEQS, row = row0 + pos.sl, slc = row < in.nseq ? pos.sl : in.nseq - 1 - row0;
    in.src += int64_t(blockIdx.y) * in.bstride;        // field of a stack
    out.dst += int64_t(blockIdx.y) * out.bstride;
    cx<T> v[C::P];
    if constexpr (C::ABL & 1) {
#pragma unroll
        for (int m = 0; m < C::P; ++m) v[m] = cx<T>{T(pos.tid + m), T(row)};
    } else if constexpr (SYN != 0) {
        ce_load_synth<C, SYN>(v, in, sy, row0, slc, pos.t);
    } else {
        ce_load<C, WIN>(v, in, row0, slc, pos.t);
    }
    ce_run<C, 0>(v, pos, ce_smem, tw);
    if constexpr (C::ABL & 2) {
        T acc = T(0);
#pragma unroll
        for (int m = 0; m < C::P; ++m) acc += v[m].x * v[m].y;
        if (acc == T(-12345.678)) ce_store_row<C>(v, out, row0, pos.sl, pos.t);
    } else {
        if (row < in.nseq) ce_store_row<C>(v, out, row0, pos.sl, pos.t);
    }
}

template <typename C, bool WIN>
__global__ __launch_bounds__(C::NT, ce_waves_per_eu<C>()) void ce_cols_kernel(CeIn<typename C::T> in, CeColOut<typename C::T> out,
                                                                               const cx<typename C::T>* __restrict__ tw, int log_g) {
    using T = typename C::T;
    extern __shared__ __align__(16) char ce_smem[];
    // tiles that share 128 B lines run on one XCD (fft_mixed_kernels.h mix_cols_kernel)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = ((slot >> log_g) << (log_g + 3)) + (xcd << log_g) + (slot & ((1 << log_g) - 1));
    const int c0 = tile * C::SEQS;
    if (c0 >= in.nseq) return;
    const CePos pos = ce_pos<C>(threadIdx.x);
    const int col = c0 + pos.sl, slc = col < in.nseq ? pos.sl : in.nseq - 1 - c0;
    in.src += int64_t(blockIdx.y) * in.bstride;        // field of a stack
    out.dst = static_cast<char*>(out.dst) + int64_t(blockIdx.y) * out.bstride * int64_t(out.epilogue == 0 ? sizeof(cx<T>) : sizeof(T));
    cx<T> v[C::P];
    if constexpr (C::ABL & 1) {
#pragma unroll
        for (int m = 0; m < C::P; ++m) v[m] = cx<T>{T(pos.tid + m), T(col)};
    } else {
        ce_load<C, WIN>(v, in, c0, slc, pos.t);
    }
    ce_run<C, 0>(v, pos, ce_smem, tw);
    if constexpr (C::ABL & 2) {
        T acc = T(0);
#pragma unroll
        for (int m = 0; m < C::P; ++m) acc += v[m].x * v[m].y;
        if (acc == T(-12345.678)) ce_store_col<C>(v, out, c0, pos.sl, pos.t);
    } else {
        if (col < in.nseq) ce_store_col<C>(v, out, c0, pos.sl, pos.t);
    }
}

// Middle pass of fft2 -> x H -> ifft2 on a composite column length: forward transform, multiplier, inverse transform on the registers of
// the tile -- the spectrum comes out of the forward stages in the layout the loads had, so the second run starts where the first ended
// (ifft = conj fft conj: the multiplier step leaves conj(x h), the store conjugates).  What mix_cols_mul_kernel does in LDS.
template <typename C, bool WIN, int KIND>
__global__ __launch_bounds__(C::NT, ce_waves_per_eu<C>()) void ce_cols_mul_kernel(CeIn<typename C::T> in, CeMul<typename C::T> mm, cx<typename C::T>* dst,
                                                                                   int64_t dst_pitch, const cx<typename C::T>* __restrict__ tw, int log_g) {
    using T = typename C::T;
    extern __shared__ __align__(16) char ce_smem[];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = ((slot >> log_g) << (log_g + 3)) + (xcd << log_g) + (slot & ((1 << log_g) - 1));
    const int c0 = tile * C::SEQS;
    if (c0 >= in.nseq) return;
    const CePos pos = ce_pos<C>(threadIdx.x);
    const int col = c0 + pos.sl, slc = col < in.nseq ? pos.sl : in.nseq - 1 - c0;
    cx<T> v[C::P];
    ce_load<C, WIN>(v, in, c0, slc, pos.t);
    ce_run<C, 0>(v, pos, ce_smem, tw);
    ce_mul_col<C, KIND>(v, mm, c0, slc, pos.t);
    __syncthreads();        // the last gather of the forward run has finished in every wave
    // the second run loads its twiddles again: through a pointer the compiler cannot identify with the first, or it keeps every stage's
    // twiddles of the forward run alive for the inverse (60 .. 130 registers spilled)
    const cx<T>* tw2 = tw;
    asm volatile("" : "+s"(tw2));
    ce_run<C, 0>(v, pos, ce_smem, tw2);
    if (col < in.nseq) ce_store_mid<C>(v, dst, dst_pitch, c0, pos.sl, pos.t);
}

template <typename C>
int ce_rows_go(const CeIn<typename C::T>& in, const CeRowOut<typename C::T>& out, const cx<typename C::T>* tw, hipStream_t st, const CeSynth& sy, int nb) {
    const bool win = !(in.ax.off == 0 && in.ax.len == in.ax.n);
    const int groups = (in.nseq + C::SEQS - 1) / C::SEQS;
    auto go = [&](auto kernel) {
        const int rc = mix_set_lds(kernel, C::LDS_BYTES);
        if (rc) return rc;
        hipLaunchKernelGGL(kernel, dim3(groups, nb), dim3(C::NT), C::LDS_BYTES, st, in, out, tw, sy);
        return int(hipGetLastError());
    };
    if constexpr (sizeof(typename C::T) == 4) {
        if (sy.kind == 3) return go(ce_rows_kernel<C, true, 3>);
        if (sy.kind == 2) return go(ce_rows_kernel<C, true, 2>);
    }
    return win ? go(ce_rows_kernel<C, true>) : go(ce_rows_kernel<C, false>);
}
template <typename C>
int ce_cols_go(const CeIn<typename C::T>& in, const CeColOut<typename C::T>& out, const cx<typename C::T>* tw, hipStream_t st, int log_g, int nb) {
    const bool win = !(in.ax.off == 0 && in.ax.len == in.ax.n);
    // adjacent tiles that run on one XCD: the measured best of the shape (tools/ce_gen.py), else as many as share a 128 B line
    if (log_g < 0)
        for (log_g = 0; (size_t(C::SEQS) << log_g) * sizeof(cx<typename C::T>) < 128 && log_g < 3;) ++log_g;
    if (tuning().ce_log_g >= 0) log_g = tuning().ce_log_g > 8 ? 8 : tuning().ce_log_g;
    const int tiles = (in.nseq + C::SEQS - 1) / C::SEQS, round = 8 << log_g, groups = (tiles + round - 1) / round * round;
    if (win) {
        const int rc = mix_set_lds(ce_cols_kernel<C, true>, C::LDS_BYTES);
        if (rc) return rc;
        hipLaunchKernelGGL((ce_cols_kernel<C, true>), dim3(groups, nb), dim3(C::NT), C::LDS_BYTES, st, in, out, tw, log_g);
    } else {
        const int rc = mix_set_lds(ce_cols_kernel<C, false>, C::LDS_BYTES);
        if (rc) return rc;
        hipLaunchKernelGGL((ce_cols_kernel<C, false>), dim3(groups, nb), dim3(C::NT), C::LDS_BYTES, st, in, out, tw, log_g);
    }
    return int(hipGetLastError());
}

template <typename C>
int ce_cols_mul_go(const CeIn<typename C::T>& in, const CeMul<typename C::T>& mm, cx<typename C::T>* dst, int64_t dst_pitch, const cx<typename C::T>* tw,
                   hipStream_t st, int log_g) {
    const bool win = !(in.ax.off == 0 && in.ax.len == in.ax.n);
    if (log_g < 0)
        for (log_g = 0; (size_t(C::SEQS) << log_g) * sizeof(cx<typename C::T>) < 128 && log_g < 3;) ++log_g;
    if (tuning().ce_log_g >= 0) log_g = tuning().ce_log_g > 8 ? 8 : tuning().ce_log_g;
    const int tiles = (in.nseq + C::SEQS - 1) / C::SEQS, round = 8 << log_g, groups = (tiles + round - 1) / round * round;
    auto go = [&](auto kernel) {
        const int rc = mix_set_lds(kernel, C::LDS_BYTES);
        if (rc) return rc;
        hipLaunchKernelGGL(kernel, dim3(groups), dim3(C::NT), C::LDS_BYTES, st, in, mm, dst, dst_pitch, tw, log_g);
        return int(hipGetLastError());
    };
    if (mm.kind == MUL_FULL) return win ? go(ce_cols_mul_kernel<C, true, MUL_FULL>) : go(ce_cols_mul_kernel<C, false, MUL_FULL>);
    return win ? go(ce_cols_mul_kernel<C, true, MUL_SEPARABLE>) : go(ce_cols_mul_kernel<C, false, MUL_SEPARABLE>);
}

// what the general entry points hand over (fft_mixed_kernels.h mix_rows_impl / mix_cols_impl / mix_cols_mul_impl).  The kernels address
// with one unsigned 32-bit BYTE offset per lane (fft_ce.h ce_at): the largest one a view can produce must fit, else the general kernel runs.
// (ce_fits32, kCeMaxSeqs: pm_internal.h)

template <typename T>
bool ce_rows_view(const DirectIn<T>& in, cx<T>* out, int64_t out_ld, int64_t out_bstride, const RowStoreNat<T>* o, CeIn<T>& ci, CeRowOut<T>& ro, CeSynth& sy) {
    if (in.real || in.s_i != 1 || in.nseq <= 0) return false;
    if (in.synth && (sizeof(T) != 4 || (in.synth != 2 && in.synth != 3))) return false;
    if (in.nb > 1 && (in.synth || o)) return false;      // stacks: plain complex fields to natural rows      // synthesis: complex64 only (the fp64 sincospi does not fit the tile's registers)
    sy = CeSynth{in.synth, in.k2, in.amp, in.amp ? in.amp_kind : 0, in.amp_ld};
    if (o && (o->use_ay || o->bstride || o->ax.n != in.ax.n)) return false;
    const int64_t n = in.ax.n, old_ = o ? o->ld : out_ld;
    // loads: (sl pitch + q0 + N) elements; stores: (sl ld + q0 + N)
    if (in.s_seq < 0 || old_ < 0 || !ce_fits32(kCeMaxSeqs * in.s_seq + 2 * n, sizeof(cx<T>)) || !ce_fits32(kCeMaxSeqs * old_ + 2 * n, sizeof(cx<T>))) return false;
    ci = CeIn<T>{in.src, in.s_seq, in.ax, in.nseq, in.conj ? T(-1) : T(1), in.bstride};
    ro = o ? CeRowOut<T>{o->dst, o->ld, 1, o->ax, o->scale, o->conj ? -o->scale : o->scale, 0}
           : CeRowOut<T>{out, out_ld, 0, AxisMap{in.ax.n, in.ax.n, 0, 0}, T(1), T(1), out_bstride};
    return true;
}
template <typename T>
bool ce_cols_view(const DirectIn<T>& in, const ColStoreNat<T>& out, CeIn<T>& ci, CeColOut<T>& co) {
    if (in.real || in.synth || in.s_seq != 1 || in.nseq <= 0) return false;
    const bool whole = out.ay.off == 0 && out.ay.len == out.ay.n && out.ax.off == 0 && out.ax.len == out.ax.n;
    if (!whole || out.mul_kind != MUL_NONE || out.ay.n != in.ax.n || out.ax.n != in.nseq) return false;
    if (out.epilogue != EPI_NONE && out.epilogue != EPI_ABS2 && out.epilogue != EPI_ABS2_ACCUM) return false;
    const int64_t n = in.ax.n;
    // loads: (q0 + N) pitch + sl elements; stores: (k0 + N) ld + qx
    if (in.s_i < 0 || out.ld < 0 || !ce_fits32(2 * n * in.s_i + kCeMaxSeqs, sizeof(cx<T>)) || !ce_fits32(2 * n * out.ld + out.ax.n, sizeof(cx<T>))) return false;
    ci = CeIn<T>{in.src, in.s_i, in.ax, in.nseq, in.conj ? T(-1) : T(1), in.bstride};
    co = CeColOut<T>{out.dst, out.ld, out.ay.n, out.ay.shift, out.ax.n, out.ax.shift, out.scale, out.conj ? -out.scale : out.scale,
                     out.epilogue, out.weight, out.bstride};
    return true;
}
template <typename T>
bool ce_mid_view(const DirectIn<T>& in, const MidMul<T>& m, int64_t dst_pitch, CeIn<T>& ci, CeMul<T>& mm) {
    if (in.real || in.synth || in.conj || in.s_seq != 1 || in.nseq <= 0) return false;
    if ((m.kind != MUL_FULL && m.kind != MUL_SEPARABLE) || m.bstride || m.bstride_x || m.ystep > 1) return false;
    const int64_t n = in.ax.n;
    if (in.s_i < 0 || dst_pitch < 0 || !ce_fits32(2 * n * in.s_i + kCeMaxSeqs, sizeof(cx<T>)) || !ce_fits32(n * dst_pitch + kCeMaxSeqs, sizeof(cx<T>)) ||
        (m.kind == MUL_FULL && (m.ld < 0 || !ce_fits32(n * m.ld + kCeMaxSeqs, sizeof(cx<T>)))))
        return false;
    if (in.nb > 1) return false;
    ci = CeIn<T>{in.src, in.s_i, in.ax, in.nseq, T(1), 0};
    mm = CeMul<T>{m.kind, m.conj, m.mul, m.mul_x, m.ld};
    return true;
}

}  // namespace pm
