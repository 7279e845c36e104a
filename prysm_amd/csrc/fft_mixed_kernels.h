// Kernels and launch templates of the mixed-radix path (fft_mixed.h); instantiated per precision, direction and kernel class in
// fft_mixed_*.hip so that the translation units compile in parallel.  Same contracts as direct_rows / direct_cols (dft_direct.hip) and
// blue_rows / blue_cols (bluestein.hip), no scratch memory.
#pragma once
#include <mutex>
#include <utility>
#include <vector>

#include "fft_mixed.h"
#include "pm_internal.h"

namespace pm {

bool mix_plan_for(int n, size_t es, MixPlan& p);             // fft_mixed.hip: the cached factorisation of n for elements of es bytes
void mix_pick_pads(const MixPlan& p, size_t es, bool col, MixShape& sh);     // fft_mixed.hip: the LDS padding of this launch shape (bank model)
const MixPlan* mix_plan_dev(int n, size_t es, int* err);     // capi.hip: its device-resident copy (plan cache, beside the twiddles)

// ColStoreNat element store (fft_io.h store_one) with a 32-bit offset from the array base, and without the window test when the caller
// knows the view keeps every bin
template <bool CHECK, typename T>
__device__ __forceinline__ void mix_store_col(const ColStoreNat<T>& p, int k, int c, cx<T> x) {
    const int qy = p.ay.map(k), qx = p.ax.map(c);
    if (CHECK && (qy < 0 || qx < 0)) return;
    x = cscale(x, p.scale);
    if (p.conj) x.y = -x.y;
    if (p.mul_kind == MUL_FULL) {
        const cx<T> h = p.mul[int64_t(k) * p.mul_ld + c];
        x = p.mul_conj ? cmulc(x, h) : cmul(x, h);
    } else if (p.mul_kind == MUL_SEPARABLE) {
        const cx<T> h = cmul(p.mul[k], p.mul_x[c]);
        x = p.mul_conj ? cmulc(x, h) : cmul(x, h);
    }
    const uint32_t off = mix_mul24(uint32_t(qy), uint32_t(p.ld)) + uint32_t(qx);
    if (p.epilogue == EPI_NONE) {
        mix_st(reinterpret_cast<cx<T>*>(p.dst) + off, x);
    } else {
        T* o = reinterpret_cast<T*>(p.dst) + off;
        const T i2 = x.x * x.x + x.y * x.y;
        if (p.epilogue == EPI_ABS2)
            *o = i2;
        else
            *o += p.weight * i2;
    }
}

// ... and the store of the plain 2-D transform -- every bin kept (rotations only), no multiplier, complex output, a full tile: the
// element's row is one add and one unsigned min (k + shift wrapped), its column depends on the butterfly only, the offset is one 24-bit
// multiply-add.  The general form above costs ~25 vector instructions and six uniform branches PER ELEMENT inside the unrolled butterfly
// (the last stage of a column kernel of the class of 20 came to 80 instructions per sample against 17 in the row kernel).
template <typename T>
struct MixColStoreWhole {
    cx<T>* dst;
    uint32_t ld;
    int ny, sy, nx, sx;      // lengths and rotations of the output rows / columns (AxisMap n, shift)
    T sr, si;                // scale, and -scale for a conjugated output
    int c0;
    __device__ __forceinline__ void operator()(int sl, int k, cx<T> v) const {
        const uint32_t qy0 = uint32_t(k + sy), qy1 = qy0 - uint32_t(ny), qy = qy0 < qy1 ? qy0 : qy1;
        const uint32_t qx0 = uint32_t(c0 + sl + sx), qx1 = qx0 - uint32_t(nx), qx = qx0 < qx1 ? qx0 : qx1;
        mix_st(dst + (mix_mul24(qy, ld) + qx), cx<T>{v.x * sr, v.y * si});
    }
};

// start-up stagger (MixShape::stagger)
__device__ __forceinline__ void mix_stagger_delay(MixShape sh) {
    if (sh.stagger > 0 && int(blockIdx.x) < sh.first_round) {
        const unsigned h = (blockIdx.x * 2654435761u) >> 29;
        for (unsigned i = 0; i < h * unsigned(sh.stagger); ++i) __builtin_amdgcn_s_sleep(8);
    }
}

// LDS layout of the row mode (round 4): the sequences of the workgroup INTERLEAVED, [point][sequence] with lanes across the sequences first
// -- the column mode's layout and lane order (template argument COL = true of the stage functions) on top of the row mode's global
// addressing (the Fetch / Store functors).  With [sequence][point] the lanes of a wave ran along ONE sequence and met the digit-reversed
// strides of the last stage and the short blocks of the middle stages in every 32-lane group (bank conflicts on 56 % of the LDS cycles at
// 3000 points, 34 % with the best padding); interleaved, a group covers 16 (8, 4) consecutive points of 2 (4, 8) rows.  A wave's global
// accesses become 2 (4, 8) contiguous runs instead of one -- still whole 256 B / 128 B / 64 B pieces.  Rows per workgroup: a power of two.
// SYNTH: the instantiation whose first stage synthesises the pupil (in.synth); kept apart because the fp64 sincospi of the complex128 form
// costs 40 registers that the plain kernel of the class of 10 does not have to pay (94 -> 134 VGPRs, five waves per SIMD -> three)
template <typename T, int MAXR, bool SYNTH = false>
__global__ __launch_bounds__(512) void mix_rows_kernel(const MixPlan* __restrict__ pp, MixShape sh, DirectIn<T> in, MixRowOut<T> out,
                                                       const cx<T>* __restrict__ tw) {
    const MixPlan& p = *pp;
    extern __shared__ __align__(16) char mix_smem[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(mix_smem);
    // the rows of the workgroup: seqs consecutive ones, or (FOLD, experiment builds) the pair (g, g + H)
    constexpr bool fold = false;
    const int seq0 = fold ? int(blockIdx.x) : int(blockIdx.x) * sh.seqs, tid = threadIdx.x, nt = blockDim.x;
    const int nvalid = fold ? 2 : (in.nseq - seq0 < sh.seqs ? in.nseq - seq0 : sh.seqs);
    const uint32_t rpitch = fold ? uint32_t(out.fold_h) * uint32_t(in.s_seq) : uint32_t(in.s_seq);
    const T ysign = in.conj ? T(-1) : T(1);
    const bool whole = in.ax.off == 0 && in.ax.len == in.ax.n && nvalid == sh.seqs;
    if (SYNTH && in.synth == 3) {
        const MixFetchSynth<T, true> fetch{in.src + int64_t(seq0) * in.s_seq, rpitch, in.ax, nvalid, in.k2, nullptr, 0, 0u};
        mix_run_first<T, true, MAXR>(p, sh, tid, nt, lds, tw, fetch);
    } else if (SYNTH && in.synth == 2) {
        const char* a0 = reinterpret_cast<const char*>(in.amp);
        if (a0) a0 += int64_t(seq0) * in.amp_ld * (in.amp_kind == 1 ? 4 : (in.amp_kind == 2 ? 8 : 1));
        const MixFetchSynth<T, false> fetch{reinterpret_cast<const T*>(in.src) + int64_t(seq0) * in.s_seq, rpitch, in.ax, nvalid, in.k2,
                                            a0, a0 ? in.amp_kind : 0, fold ? uint32_t(out.fold_h) * uint32_t(in.amp_ld) : uint32_t(in.amp_ld)};
        mix_run_first<T, true, MAXR>(p, sh, tid, nt, lds, tw, fetch);
    } else if (in.real) {
        const MixFetch<T, false, true> fetch{reinterpret_cast<const T*>(in.src) + int64_t(seq0) * in.s_seq, rpitch, in.ax, ysign, nvalid};
        mix_run_first<T, true, MAXR>(p, sh, tid, nt, lds, tw, fetch);
    } else if (whole) {
        const MixFetchWhole<T, false> fetch{in.src + int64_t(seq0) * in.s_seq, rpitch, in.ax.n, in.ax.shift, ysign};
        mix_run_first<T, true, MAXR, 3>(p, sh, tid, nt, lds, tw, fetch);
    } else {
        const MixFetch<T, false, false> fetch{in.src + int64_t(seq0) * in.s_seq, rpitch, in.ax, ysign, nvalid};
        mix_run_first<T, true, MAXR>(p, sh, tid, nt, lds, tw, fetch);
    }
    __syncthreads();
    const int nstage = p.nstage;
    for (int s = 1; s + 1 < nstage; ++s) {
        mix_run_mid<T, true, MAXR>(p, sh, s, tid, nt, lds, tw);
        __syncthreads();
    }
    if (nvalid == sh.seqs && !out.mapped) {
        cx<T>* dst0 = out.dst + int64_t(seq0) * out.ld;
        const uint32_t ld = uint32_t(out.ld);
        auto store = [&](int sl, int k, cx<T> v) { mix_st(dst0 + (mix_mul24(uint32_t(sl), ld) + uint32_t(k)), v); };
        mix_run_last<T, true, MAXR>(p, sh, tid, nt, lds, store);
    } else {
        auto store = [&](int sl, int k, cx<T> v) {
            if (sl < nvalid) mix_store_row(out, seq0 + sl, k, v);
        };
        mix_run_last<T, true, MAXR>(p, sh, tid, nt, lds, store);
    }
}

// NTMAX: 512, or 1024 (128-register cap) for the classes whose register count allows it -- the column pass holds ONE workgroup per CU
// from ~2000-point columns (four columns take most of the LDS), so the waves of that workgroup are all the latency hiding there is
template <typename T, int MAXR, int NTMAX = 512>
__global__ __launch_bounds__(NTMAX) void mix_cols_kernel(const MixPlan* __restrict__ pp, MixShape sh, DirectIn<T> in, ColStoreNat<T> out,
                                                         const cx<T>* __restrict__ tw, int log_g) {
    const MixPlan& p = *pp;
    extern __shared__ __align__(16) char mix_smem[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(mix_smem);
    // workgroups go round the 8 XCDs: slots s, s + 1 .. of ONE XCD take 2^log_g adjacent tiles, which share the 128 B lines of the rows
    // they read and write -- the lines then stay in that XCD's L2 instead of crossing the fabric once per tile
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = ((slot >> log_g) << (log_g + 3)) + (xcd << log_g) + (slot & ((1 << log_g) - 1));
    const int c0 = tile * sh.seqs, tid = threadIdx.x, nt = blockDim.x;
    if (c0 >= in.nseq) return;
    const int nvalid = in.nseq - c0 < sh.seqs ? in.nseq - c0 : sh.seqs;
    const T ysign = in.conj ? T(-1) : T(1);
    mix_stagger_delay(sh);
    const bool whole_in = in.ax.off == 0 && in.ax.len == in.ax.n && nvalid == sh.seqs;
    if (in.real) {
        const MixFetch<T, true, true> fetch{reinterpret_cast<const T*>(in.src) + c0, uint32_t(in.s_i), in.ax, ysign, nvalid};
        mix_run_first<T, true, MAXR>(p, sh, tid, nt, lds, tw, fetch);
    } else if (whole_in) {
        const MixFetchWhole<T, true> fetch{in.src + c0, uint32_t(in.s_i), in.ax.n, in.ax.shift, ysign};
        mix_run_first<T, true, MAXR>(p, sh, tid, nt, lds, tw, fetch);
    } else {
        const MixFetch<T, true, false> fetch{in.src + c0, uint32_t(in.s_i), in.ax, ysign, nvalid};
        mix_run_first<T, true, MAXR>(p, sh, tid, nt, lds, tw, fetch);
    }
    __syncthreads();
    const int nstage = p.nstage;
    for (int s = 1; s + 1 < nstage; ++s) {
        mix_run_mid<T, true, MAXR>(p, sh, s, tid, nt, lds, tw);
        __syncthreads();
    }
    const bool whole = out.ay.off == 0 && out.ay.len == out.ay.n && out.ax.off == 0 && out.ax.len == out.ax.n;
    if (nvalid == sh.seqs && whole && out.mul_kind == MUL_NONE && out.epilogue == EPI_NONE) {
        const MixColStoreWhole<T> store{reinterpret_cast<cx<T>*>(out.dst), uint32_t(out.ld), out.ay.n, out.ay.shift, out.ax.n, out.ax.shift,
                                        out.scale, out.conj ? -out.scale : out.scale, c0};
        mix_run_last<T, true, MAXR>(p, sh, tid, nt, lds, store);
    } else if (nvalid == sh.seqs && whole) {
        auto store = [&](int sl, int k, cx<T> v) { mix_store_col<false>(out, k, c0 + sl, v); };
        mix_run_last<T, true, MAXR>(p, sh, tid, nt, lds, store);
    } else {
        auto store = [&](int sl, int k, cx<T> v) {
            if (sl < nvalid) mix_store_col<true>(out, k, c0 + sl, v);
        };
        mix_run_last<T, true, MAXR>(p, sh, tid, nt, lds, store);
    }
}



// Middle pass of fft2 -> x H -> ifft2 on a composite column length (fft_mixed.h, the transposed stages): a tile of adjacent columns of the
// natural intermediate goes through forward stages, the multiplier and the transposed stages without leaving the LDS, and comes back as
// the UNNORMALISED inverse column transform of (column spectrum x H) -- what the engine's middle pass (fft_kernels.h) leaves for the last
// row pass.  2 nstage - 1 phases, one barrier between phases; may run in place (a workgroup reads its columns whole before it writes).
template <typename T>
struct MixMul {
    int kind, conj;
    const cx<T>* mul;       // MUL_FULL: mul[k * ld + c]; MUL_SEPARABLE: hy[k]
    const cx<T>* mul_x;     // MUL_SEPARABLE: hx[c]
    int64_t ld;
    int ncols;
};

template <typename T, int MAXR, int NTMAX = 512>
__global__ __launch_bounds__(NTMAX) void mix_cols_mul_kernel(const MixPlan* __restrict__ pp, MixShape sh, DirectIn<T> in, MixMul<T> mm, cx<T>* dst,
                                                             uint32_t dst_pitch, const cx<T>* __restrict__ tw, int log_g) {
    const MixPlan& p = *pp;
    extern __shared__ __align__(16) char mix_smem[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(mix_smem);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = ((slot >> log_g) << (log_g + 3)) + (xcd << log_g) + (slot & ((1 << log_g) - 1));
    const int c0 = tile * sh.seqs, tid = threadIdx.x, nt = blockDim.x;
    if (c0 >= in.nseq) return;
    const int nvalid = in.nseq - c0 < sh.seqs ? in.nseq - c0 : sh.seqs;
    mix_stagger_delay(sh);
    const bool whole_in = in.ax.off == 0 && in.ax.len == in.ax.n && nvalid == sh.seqs;
    if (whole_in) {
        const MixFetchWhole<T, true> fetch{in.src + c0, uint32_t(in.s_i), in.ax.n, in.ax.shift, T(1)};
        mix_run_first<T, true, MAXR>(p, sh, tid, nt, lds, tw, fetch);
    } else {
        const MixFetch<T, true, false> fetch{in.src + c0, uint32_t(in.s_i), in.ax, T(1), nvalid};
        mix_run_first<T, true, MAXR>(p, sh, tid, nt, lds, tw, fetch);
    }
    __syncthreads();
    const int nstage = p.nstage;
    for (int s = 1; s + 1 < nstage; ++s) {
        mix_run_mid<T, true, MAXR>(p, sh, s, tid, nt, lds, tw);
        __syncthreads();
    }
    if (mm.kind == MUL_FULL) {
        auto mul = [&](int sl, int k, cx<T> v) {
            const int c = sl < nvalid ? c0 + sl : c0;
            const cx<T> h = mix_ld(mm.mul + (int64_t(k) * mm.ld + c));
            const cx<T> r = mm.conj ? cmulc(v, h) : cmul(v, h);
            return cx<T>{r.x, -r.y};
        };
        mix_run_last_mul<T, true, MAXR>(p, sh, tid, nt, lds, mul);
    } else {
        auto mul = [&](int sl, int k, cx<T> v) {
            const int c = sl < nvalid ? c0 + sl : c0;
            const cx<T> h = cmul(mix_ld(mm.mul + k), mix_ld(mm.mul_x + c));
            const cx<T> r = mm.conj ? cmulc(v, h) : cmul(v, h);
            return cx<T>{r.x, -r.y};
        };
        mix_run_last_mul<T, true, MAXR>(p, sh, tid, nt, lds, mul);
    }
    __syncthreads();
    for (int s = nstage - 2; s >= 1; --s) {
        mix_run_mid_t<T, true, MAXR>(p, sh, s, tid, nt, lds, tw);
        __syncthreads();
    }
    cx<T>* d0 = dst + c0;
    auto store = [&](int sl, int k, cx<T> v) {
        if (sl < nvalid) mix_st(d0 + (mix_mul24(uint32_t(k), dst_pitch) + uint32_t(sl)), cx<T>{v.x, -v.y});
    };
    mix_run_first_t<T, true, MAXR>(p, sh, tid, nt, lds, tw, store);
}

static constexpr size_t kMixLdsHard = 156 * 1024;

static inline int round_up64(int v) { return (v + 63) & ~63; }
// workgroups that start together when the launch begins (256 CUs; LDS and the 32 wave slots of a CU bound the workgroups per CU)
static inline int mix_first_round(size_t lds, int nt) {
    const int by_lds = lds ? int(size_t(160) * 1024 / lds) : 8, by_waves = 2048 / (nt < 64 ? 64 : nt);
    const int per_cu = by_lds < by_waves ? by_lds : by_waves;
    return 256 * (per_cu < 1 ? 1 : per_cu);
}

// threads of a workgroup: one butterfly per thread in the stage with the most butterflies, within [64, cap]
static inline int mix_threads(const MixPlan& p, int seqs, int cap, int forced, bool even = true) {
    int rmin = kMixMaxRadix;
    for (int s = 0; s < p.nstage; ++s) rmin = p.radix[s] < rmin ? p.radix[s] : rmin;
    const int most = seqs * (p.n / rmin);
    int nt = round_up64(most);
    // several rounds per stage: even them out (not in the row kernel, whose first stage takes its trips at once: 3000-point rows 75.4 us at
    // 256 threads against 78.3 at 192, profiles/r04/exp_mix_shape.log)
    if (nt > cap && !even) nt = cap;
    if (nt > cap) {
        const int rounds = (most + cap - 1) / cap;
        nt = round_up64((most + rounds - 1) / rounds);
    }
    if (forced > 0) nt = round_up64(forced);
    const int hard = cap > 512 ? cap : 512;
    return nt < 64 ? 64 : (nt > hard ? hard : nt);
}
// classes whose column kernel also exists for 1024-thread workgroups (128 registers per thread: complex64 94 .. 106, complex128 class of 10: 94)
template <typename T> constexpr bool mix_cols_wide(int maxr_class) { return sizeof(T) == 4 || maxr_class <= 10; }

// more than 64 KiB of dynamic LDS needs the kernel's limit raised: once per (kernel, device), to the most any launch asks for
template <typename K>
static int mix_set_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> done;
    const void* f = reinterpret_cast<const void*>(kernel);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lk(mu);
    for (const auto& d : done)
        if (d.first == f && d.second == dev) return 0;
    const int rc = int(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, int(kMixLdsHard)));
    if (rc == 0) done.emplace_back(f, dev);
    return rc;
}

// one kernel class (largest factor <= MAXR: 10, 16, 20); defined in fft_mixed_{rows,cols}_{f32,f64}[_mid].hip.  Registers per class, rows
// complex64 / complex128: 66 / 94, 94 / 176, 106 / 186.  (A class of factors up to 32 -- 171 / 256 + spills -- was built and lost to plans
// with one more stage in a leaner class: fft_mixed.h mix_radix_ok.)
template <typename T, int MAXR>
int mix_rows_launch(const MixPlan* p, MixShape sh, const DirectIn<T>& in, const MixRowOut<T>& ro, const cx<T>* tw, int groups, int nt, size_t lds, hipStream_t st);
template <typename T, int MAXR>
int mix_cols_launch(const MixPlan* p, MixShape sh, const DirectIn<T>& in, const ColStoreNat<T>& out, const cx<T>* tw, int log_g, int groups, int nt, size_t lds,
                    hipStream_t st);

template <typename T, int MAXR>
int mix_rows_launch_impl(const MixPlan* p, MixShape sh, const DirectIn<T>& in, const MixRowOut<T>& ro, const cx<T>* tw, int groups, int nt, size_t lds,
                         hipStream_t st) {
    if (in.synth) {
        const int rc = mix_set_lds(mix_rows_kernel<T, MAXR, true>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL((mix_rows_kernel<T, MAXR, true>), dim3(groups), dim3(nt), lds, st, p, sh, in, ro, tw);
        return int(hipGetLastError());
    }
    const int rc = mix_set_lds(mix_rows_kernel<T, MAXR>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((mix_rows_kernel<T, MAXR>), dim3(groups), dim3(nt), lds, st, p, sh, in, ro, tw);
    return int(hipGetLastError());
}
template <typename T, int MAXR>
int mix_cols_launch_impl(const MixPlan* p, MixShape sh, const DirectIn<T>& in, const ColStoreNat<T>& out, const cx<T>* tw, int log_g, int groups, int nt,
                         size_t lds, hipStream_t st) {
    if constexpr (mix_cols_wide<T>(MAXR)) {
        if (nt > 512) {
            const int rc = mix_set_lds(mix_cols_kernel<T, MAXR, 1024>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL((mix_cols_kernel<T, MAXR, 1024>), dim3(groups), dim3(nt), lds, st, p, sh, in, out, tw, log_g);
            return int(hipGetLastError());
        }
    }
    if (nt > 512) nt = 512;
    const int rc = mix_set_lds(mix_cols_kernel<T, MAXR>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((mix_cols_kernel<T, MAXR>), dim3(groups), dim3(nt), lds, st, p, sh, in, out, tw, log_g);
    return int(hipGetLastError());
}

template <typename T>
int mix_rows_impl(const DirectIn<T>& in, cx<T>* out, int64_t out_ld, hipStream_t st, const RowStoreNat<T>* o, const MixFold<T>* fold, int64_t out_bstride) {
    const int n = in.ax.n, nseq = in.nseq;
    if (nseq <= 0 || n <= 0) return 0;
    if (!fold && tuning().mix_engine) {     // a length with a compile-time plan, a plain view: the register engine (fft_ce.h)
        int rc = 0;
        if (ce_rows<T>(in, out, out_ld, o, st, &rc, out_bstride)) return rc;
    }
    if (in.nb > 1) {      // a stack on the general kernel: field by field (complex fields, natural outputs: capi.hip fft2_run)
        if (o || fold || in.real || in.synth) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: a stack with a view it does not take");
        for (int b = 0; b < in.nb; ++b) {
            DirectIn<T> one = in;
            one.src = in.src + int64_t(b) * in.bstride;
            one.nb = 1;
            const int rc = mix_rows_impl<T>(one, out + int64_t(b) * out_bstride, out_ld, st, nullptr, nullptr, 0);
            if (rc) return rc;
        }
        return 0;
    }
    if (fold && (o || nseq != 2 * fold->H || !mix_fits(n, int64_t(fold->H) * in.s_seq, sizeof(cx<T>), false) ||
                 (in.amp && !mix_fits(n, int64_t(fold->H) * in.amp_ld, sizeof(cx<T>), false))))
        return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: folded row pass on a shape it does not take");
    MixPlan p;
    if (!mix_plan_for(n, sizeof(cx<T>), p)) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d has no plan", n);
    if (in.s_i != 1 || !mix_fits(n, in.s_seq, sizeof(cx<T>), false) || !mix_fits(n, o ? o->ld : out_ld, sizeof(cx<T>), false))
        return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: row pitch beyond 2^24 elements");
    int err = 0;
    const cx<T>* tw = twiddles<T>(n, &err);
    if (!tw) return err;
    const size_t per = size_t(p.n) * sizeof(cx<T>);
    // rows per workgroup: two (256 threads then have enough butterflies per stage), more for short rows (about 2048 points per
    // workgroup), within 48 KiB of LDS so that three workgroups share a CU and their load / transform / store phases overlap.  Measured
    // (profiles/r03/exp_mix_sweep.log, 2-D transform, us): 1000^2 complex64 19.8 at 2 rows against 22.2 at 3; 4000^2 167 at 1 row
    // (32 KiB) against 182 at 2; 3000^2 97 at 1 or 2 and 115 at 4; complex128 3000^2 206 at 1 against 219 at 2
    // ... and again after round 4's changes (profiles/r04/exp_mix_shape.log, 1 / 2 rows at 256 threads): 3000^2 75.4 / 80.0, 4000^2 126.0 /
    // 144.9, complex128 3000^2 165.7 / 193.0, 2000^2 38.4 / 37.5, 1000^2 17.9 / 18.2 -- one row from 1024 points (more workgroups per CU)
    int seqs = n >= 1024 ? 1 : 2048 / n;
    while (seqs > 1 && size_t(seqs) * per > size_t(48) * 1024) --seqs;
    if (tuning().mix_seqs > 0) seqs = tuning().mix_seqs;
    while (seqs > 1 && seqs / 2 >= nseq) seqs /= 2;
    if (fold) seqs = 2;
    {   // a power of two: the rows of a workgroup are interleaved in LDS (mix_rows_kernel)
        int pw = 1;
        while (pw * 2 <= seqs) pw *= 2;
        seqs = pw;
    }
    if (size_t(seqs) * per > kMixLdsHard) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d does not fit the LDS", n);
    MixShape sh{seqs, 0};
    while ((1 << sh.log_seqs) < seqs) ++sh.log_seqs;
    mix_pick_pads(p, sizeof(cx<T>), true, sh);
    sh.stagger = tuning().mix_stagger;
    const MixPlan* pd = mix_plan_dev(n, sizeof(cx<T>), &err);
    if (!pd) return err;
    const size_t lds = size_t(seqs) * size_t(sh.npad) * sizeof(cx<T>);
    MixRowOut<T> ro{out, out_ld, AxisMap{n, n, 0, 0}, T(1), 0, 0, 0, 0, nullptr};
    if (o) ro = MixRowOut<T>{o->dst, o->ld, o->ax, o->scale, o->conj, 1, 0, 0, nullptr};
    if (fold) {
        ro.fold_h = fold->H;
        ro.fold_swap = fold->swap;
        ro.fold_tw = fold->tw;
    }
    const int groups = fold ? fold->H : (nseq + seqs - 1) / seqs, nt = mix_threads(p, seqs, 256, tuning().mix_nt, false);
    sh.stagger = 0;
    if (p.maxr <= 10) return mix_rows_launch<T, 10>(pd, sh, in, ro, tw, groups, nt, lds, st);
    if (p.maxr <= 16) return mix_rows_launch<T, 16>(pd, sh, in, ro, tw, groups, nt, lds, st);
    return mix_rows_launch<T, 20>(pd, sh, in, ro, tw, groups, nt, lds, st);
}

template <typename T>
int mix_cols_impl(const DirectIn<T>& in, const ColStoreNat<T>& out, hipStream_t st) {
    const int n = in.ax.n, ncols = in.nseq;
    if (ncols <= 0 || n <= 0) return 0;
    if (tuning().mix_engine) {
        int rc = 0;
        if (ce_cols<T>(in, out, st, &rc)) return rc;
    }
    if (in.nb > 1) {      // a stack on the general kernel: field by field
        if (in.real || out.mul_kind != MUL_NONE) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: a stack with a view it does not take");
        const size_t oes = out.epilogue == EPI_NONE ? sizeof(cx<T>) : sizeof(T);
        for (int b = 0; b < in.nb; ++b) {
            DirectIn<T> one = in;
            one.src = in.src + int64_t(b) * in.bstride;
            one.nb = 1;
            ColStoreNat<T> oo = out;
            oo.dst = static_cast<char*>(out.dst) + size_t(b) * size_t(out.bstride) * oes;
            const int rc = mix_cols_impl<T>(one, oo, st);
            if (rc) return rc;
        }
        return 0;
    }
    MixPlan p;
    if (!mix_plan_for(n, sizeof(cx<T>), p)) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d has no plan", n);
    if (in.s_seq != 1 || !mix_fits(n, in.s_i, sizeof(cx<T>), true) || !mix_fits(out.ay.n, out.ld, sizeof(cx<T>), true))
        return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: the array does not fit 32-bit offsets");
    int err = 0;
    const cx<T>* tw = twiddles<T>(n, &err);
    if (!tw) return err;
    const size_t per = size_t(p.n) * sizeof(cx<T>);
    // adjacent columns per workgroup (a power of two): four (32 B pieces of complex64, 64 B of complex128; the tiles of one 128 B line run
    // on one XCD, see log_g below), fewer when the LDS cannot hold four columns.  Measured (same log): 1000-point columns complex64 10.4 us
    // at 4 against 12.9 at 8 (two workgroups per CU instead of four); 3000-point columns 78 at 4 (one workgroup per CU) against 102 at 2
    int tc = 4;
    // ... eight columns of complex64 (whole 64 B pieces) where four would leave the LDS half empty and eight fit: 2000^2 35.2 us against 37.6
    // (complex128 1000^2, the same bytes: 25.0 against 23.6 -- stays at four)
    if (sizeof(T) == 4 && per > size_t(10) * 1024 && 8 * per <= kMixLdsHard) tc = 8;
    while (tc > 1 && size_t(tc) * per > kMixLdsHard) tc /= 2;
    if (tuning().mix_tc > 0) {
        tc = 1;
        while (tc * 2 <= tuning().mix_tc) tc *= 2;
    }
    while (tc > 1 && tc / 2 >= ncols) tc /= 2;
    if (size_t(tc) * per > kMixLdsHard) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d does not fit the LDS", n);
    MixShape sh{tc, 0};
    while ((1 << sh.log_seqs) < tc) ++sh.log_seqs;
    mix_pick_pads(p, sizeof(cx<T>), true, sh);
    sh.stagger = tuning().mix_stagger;
    const MixPlan* pd = mix_plan_dev(n, sizeof(cx<T>), &err);
    if (!pd) return err;
    const size_t lds = size_t(tc) * size_t(sh.npad) * sizeof(cx<T>);
    // tiles that share a 128 B line run on one XCD (mix_cols_kernel): 2^log_g adjacent tiles, the grid padded to whole rounds of them
    int log_g = 0;
    while ((size_t(tc) << log_g) * sizeof(cx<T>) < 128 && log_g < 3) ++log_g;
    // ... sixteen above 4096 points (round 5, one workgroup per CU, rotated outputs: 6006^2 558 -> 515 us, 4200^2 194 -> 182, 4800^2 / 7200^2
    // 1 - 2 %; nothing either way below -- profiles/r05/exp_knob_mix_log_g.log)
    if (n > 4096 && log_g < 4) log_g = 4;
    if (tuning().mix_log_g >= 0) log_g = tuning().mix_log_g > 4 ? 4 : tuning().mix_log_g;
    const int tiles = (ncols + tc - 1) / tc, round = 8 << log_g;
    // threads: one butterfly per thread of the busiest stage up to 512 -- and 1024 where the tile takes more than half the LDS (ONE
    // workgroup per CU: its waves are all the latency hiding there is) and the kernel class has the 1024-thread form.  Measured
    // (profiles/r04/exp_mix_ntc.log, 2-D transform us at 512 / 1024 threads): complex64 3000^2 95.1 / 84.8, 4000^2 162.4 / 158.2,
    // complex128 3000^2 189.4 / 180.4, 2000^2 66.1 / 62.6; with two workgroups per CU it loses (complex64 2000^2 40.0 / 44.5, 1536^2 34.5 / 37.7)
    const int cls = mix_class_of(p.maxr);
    const bool one_wg = lds > size_t(80) * 1024 && mix_cols_wide<T>(cls);
    const int forced = tuning().mix_ntc > 0 ? tuning().mix_ntc : (one_wg ? 1024 : 0);
    const int cap = (forced > 512 && mix_cols_wide<T>(cls)) ? 1024 : 512;
    const int groups = (tiles + round - 1) / round * round, nt = mix_threads(p, tc, cap, forced);
    sh.first_round = mix_first_round(lds, nt);
    // the stagger is for one workgroup per CU and several rounds of them (MixShape::stagger).  Measured, column kernel only, 2-D transform us
    // at 0 / 4 / 8 units (profiles/r04/exp_mix_stagger.log): complex64 3000^2 75.4 / 74.3 / 76.6, 4000^2 131.2 / 129.7 / 127.6, complex128
    // 3000^2 167.1 / 166.4 / 160.8, 2000^2 62.7 / 59.6 / 65.3; a single round only pays the delay (complex64 2000^2 35.6 / 38.3 / 43.6)
    if (lds <= size_t(80) * 1024 || tiles <= sh.first_round) sh.stagger = 0;
    // persistent workgroups (mix_cols_pers_kernel) where a CU holds one tile: more tiles than CUs, whole tiles, a plain complex input
    const bool whole_in = !in.real && in.ax.off == 0 && in.ax.len == in.ax.n, whole_out = out.ay.off == 0 && out.ay.len == out.ay.n && out.ax.off == 0 && out.ax.len == out.ax.n;
    const bool aligned = (reinterpret_cast<uintptr_t>(in.src) & 15) == 0 && ((size_t(in.s_i) * sizeof(cx<T>)) & 15) == 0 && size_t(tc) * sizeof(cx<T>) >= 16;
    if (tuning().mix_pers && lds > size_t(80) * 1024 && lds <= size_t(128) * 1024 && tiles > 256 && ncols % tc == 0 && whole_in && whole_out && aligned && out.bstride == 0) sh.pers_tiles = tiles;
    if (p.maxr <= 10) return mix_cols_launch<T, 10>(pd, sh, in, out, tw, log_g, groups, nt, lds, st);
    if (p.maxr <= 16) return mix_cols_launch<T, 16>(pd, sh, in, out, tw, log_g, groups, nt, lds, st);
    return mix_cols_launch<T, 20>(pd, sh, in, out, tw, log_g, groups, nt, lds, st);
}

// the middle pass: one launcher per precision (fft_mixed_mid_f32.hip / _f64.hip hold its kernel classes)
template <typename T, int MAXR>
int mix_cols_mul_launch_impl(const MixPlan* p, MixShape sh, const DirectIn<T>& in, const MixMul<T>& mm, cx<T>* dst, uint32_t pitch, const cx<T>* tw, int log_g,
                             int groups, int nt, size_t lds, hipStream_t st) {
    if constexpr (mix_cols_wide<T>(MAXR)) {
        if (nt > 512) {
            const int rc = mix_set_lds(mix_cols_mul_kernel<T, MAXR, 1024>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL((mix_cols_mul_kernel<T, MAXR, 1024>), dim3(groups), dim3(nt), lds, st, p, sh, in, mm, dst, pitch, tw, log_g);
            return int(hipGetLastError());
        }
    }
    if (nt > 512) nt = 512;
    const int rc = mix_set_lds(mix_cols_mul_kernel<T, MAXR>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((mix_cols_mul_kernel<T, MAXR>), dim3(groups), dim3(nt), lds, st, p, sh, in, mm, dst, pitch, tw, log_g);
    return int(hipGetLastError());
}

template <typename T>
int mix_cols_mul_impl(const DirectIn<T>& in, const MidMul<T>& m, cx<T>* dst, int64_t dst_pitch, hipStream_t st) {
    const int n = in.ax.n, ncols = in.nseq;
    if (ncols <= 0 || n <= 0) return 0;
    if (tuning().mix_engine) {
        int rc = 0;
        if (ce_cols_mul<T>(in, m, dst, dst_pitch, st, &rc)) return rc;
    }
    MixPlan p;
    if (!mix_plan_for(n, sizeof(cx<T>), p)) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d has no plan", n);
    if (in.s_seq != 1 || in.conj || in.real || !mix_fits(n, in.s_i, sizeof(cx<T>), true) || !mix_fits(n, dst_pitch, sizeof(cx<T>), true))
        return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: the array does not fit 32-bit offsets");
    int err = 0;
    const cx<T>* tw = twiddles<T>(n, &err);
    if (!tw) return err;
    const size_t per = size_t(p.n) * sizeof(cx<T>);
    int tc = 4;     // as mix_cols_impl
    while (tc > 1 && size_t(tc) * per > kMixLdsHard) tc /= 2;
    if (tuning().mix_tc > 0) {
        tc = 1;
        while (tc * 2 <= tuning().mix_tc) tc *= 2;
    }
    while (tc > 1 && tc / 2 >= ncols) tc /= 2;
    if (size_t(tc) * per > kMixLdsHard) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d does not fit the LDS", n);
    MixShape sh{tc, 0};
    while ((1 << sh.log_seqs) < tc) ++sh.log_seqs;
    mix_pick_pads(p, sizeof(cx<T>), true, sh);
    sh.stagger = tuning().mix_stagger;
    const MixPlan* pd = mix_plan_dev(n, sizeof(cx<T>), &err);
    if (!pd) return err;
    const size_t lds = size_t(tc) * size_t(sh.npad) * sizeof(cx<T>);
    int log_g = 0;
    while ((size_t(tc) << log_g) * sizeof(cx<T>) < 128 && log_g < 3) ++log_g;
    if (tuning().mix_log_g >= 0) log_g = tuning().mix_log_g > 4 ? 4 : tuning().mix_log_g;
    const int tiles = (ncols + tc - 1) / tc, round = 8 << log_g;
    const int cls = mix_class_of(p.maxr);
    const bool one_wg = lds > size_t(80) * 1024 && mix_cols_wide<T>(cls);      // as mix_cols_impl
    const int forced = tuning().mix_ntc > 0 ? tuning().mix_ntc : (one_wg ? 1024 : 0);
    const int cap = (forced > 512 && mix_cols_wide<T>(cls)) ? 1024 : 512;
    const int groups = (tiles + round - 1) / round * round, nt = mix_threads(p, tc, cap, forced);
    sh.first_round = mix_first_round(lds, nt);
    if (lds <= size_t(80) * 1024 || tiles <= sh.first_round) sh.stagger = 0;     // as mix_cols_impl
    const MixMul<T> mm{m.kind, m.conj, m.mul, m.mul_x, m.ld, ncols};
    if (p.maxr <= 10) return mix_cols_mul_launch_impl<T, 10>(pd, sh, in, mm, dst, uint32_t(dst_pitch), tw, log_g, groups, nt, lds, st);
    if (p.maxr <= 16) return mix_cols_mul_launch_impl<T, 16>(pd, sh, in, mm, dst, uint32_t(dst_pitch), tw, log_g, groups, nt, lds, st);
    return mix_cols_mul_launch_impl<T, 20>(pd, sh, in, mm, dst, uint32_t(dst_pitch), tw, log_g, groups, nt, lds, st);
}

}  // namespace pm
