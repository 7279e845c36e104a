// Kernels and launch templates of the mixed-radix path (fft_mixed.h); instantiated per precision and direction in fft_mixed_*.hip so
// that the four translation units compile in parallel.  Same contracts as direct_rows / direct_cols (dft_direct.hip) and blue_rows /
// blue_cols (bluestein.hip), no scratch memory.
#pragma once
#include "fft_mixed.h"
#include "pm_internal.h"

namespace pm {

bool mix_plan_for(int n, MixPlan& p);   // fft_mixed.hip: the cached factorisation of n

template <typename T>
static BlueIn<T> mix_in(const DirectIn<T>& in) {
    return BlueIn<T>{in.src, in.s_seq, in.s_i, in.ax, in.conj, in.real};
}

// ColStoreNat element store (fft_io.h store_one) without the window test when the caller knows the view keeps every bin
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
    if (p.epilogue == EPI_NONE) {
        reinterpret_cast<cx<T>*>(p.dst)[int64_t(qy) * p.ld + qx] = x;
    } else {
        T* o = reinterpret_cast<T*>(p.dst) + int64_t(qy) * p.ld + qx;
        const T i2 = x.x * x.x + x.y * x.y;
        if (p.epilogue == EPI_ABS2)
            *o = i2;
        else
            *o += p.weight * i2;
    }
}

template <typename T, int MAXR>
__global__ __launch_bounds__(512) void mix_rows_kernel(MixPlan p, BlueIn<T> in, int nseq, MixRowOut<T> out, const cx<T>* __restrict__ tw) {
    extern __shared__ __align__(16) char mix_smem[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(mix_smem);
    const int seq0 = blockIdx.x * p.seqs, tid = threadIdx.x, nt = blockDim.x;
    const bool full = seq0 + p.seqs <= nseq;
    auto fetch = [&](int sl, int i) {
        const bool ok = full || seq0 + sl < nseq;
        return mix_fetch(in, ok ? seq0 + sl : seq0, ok, i);
    };
    mix_run_first<T, false, MAXR>(p, tid, nt, lds, tw, fetch);
    __syncthreads();
    for (int s = 1; s + 1 < p.nstage; ++s) {
        mix_run_mid<T, false, MAXR>(p, s, tid, nt, lds, tw);
        __syncthreads();
    }
    if (full && !out.mapped) {
        auto store = [&](int sl, int k, cx<T> v) { out.dst[int64_t(seq0 + sl) * out.ld + k] = v; };
        mix_run_last<T, false, MAXR>(p, tid, nt, lds, store);
    } else {
        auto store = [&](int sl, int k, cx<T> v) {
            if (seq0 + sl < nseq) mix_store_row(out, seq0 + sl, k, v);
        };
        mix_run_last<T, false, MAXR>(p, tid, nt, lds, store);
    }
}

template <typename T, int MAXR>
__global__ __launch_bounds__(512) void mix_cols_kernel(MixPlan p, BlueIn<T> in, int ncols, ColStoreNat<T> out, const cx<T>* __restrict__ tw,
                                                       int log_g) {
    extern __shared__ __align__(16) char mix_smem[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(mix_smem);
    // workgroups go round the 8 XCDs: slots s, s + 1 .. of ONE XCD take 2^log_g adjacent tiles, which share the 128 B lines of the rows
    // they read and write -- the lines then stay in that XCD's L2 instead of crossing the fabric once per tile
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = ((slot >> log_g) << (log_g + 3)) + (xcd << log_g) + (slot & ((1 << log_g) - 1));
    const int c0 = tile * p.seqs, tid = threadIdx.x, nt = blockDim.x;
    if (c0 >= ncols) return;
    const bool full = c0 + p.seqs <= ncols;
    auto fetch = [&](int sl, int i) {
        const bool ok = full || c0 + sl < ncols;
        return mix_fetch(in, ok ? c0 + sl : c0, ok, i);
    };
    mix_run_first<T, true, MAXR>(p, tid, nt, lds, tw, fetch);
    __syncthreads();
    for (int s = 1; s + 1 < p.nstage; ++s) {
        mix_run_mid<T, true, MAXR>(p, s, tid, nt, lds, tw);
        __syncthreads();
    }
    const bool whole = out.ay.off == 0 && out.ay.len == out.ay.n && out.ax.off == 0 && out.ax.len == out.ax.n;
    if (full && whole) {
        auto store = [&](int sl, int k, cx<T> v) { mix_store_col<false>(out, k, c0 + sl, v); };
        mix_run_last<T, true, MAXR>(p, tid, nt, lds, store);
    } else {
        auto store = [&](int sl, int k, cx<T> v) {
            if (c0 + sl < ncols) mix_store_col<true>(out, k, c0 + sl, v);
        };
        mix_run_last<T, true, MAXR>(p, tid, nt, lds, store);
    }
}

static constexpr size_t kMixLdsHard = 156 * 1024;

static inline int round_up64(int v) { return (v + 63) & ~63; }

// threads of a workgroup: one butterfly per thread in the stage with the most butterflies, within [64, 512]
static inline int mix_threads(const MixPlan& p, int cap = 512) {
    int rmin = kMixMaxRadix;
    for (int s = 0; s < p.nstage; ++s) rmin = p.radix[s] < rmin ? p.radix[s] : rmin;
    const int most = p.seqs * (p.n / rmin);
    int nt = round_up64(most);
    // several rounds per stage: even them out
    if (nt > cap) {
        const int rounds = (most + cap - 1) / cap;
        nt = round_up64((most + rounds - 1) / rounds);
    }
    if (tuning().mix_nt > 0) nt = round_up64(tuning().mix_nt);
    return nt < 64 ? 64 : (nt > 512 ? 512 : nt);
}

template <typename K>
static int mix_set_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    return int(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
}

template <typename T>
int mix_rows_impl(const DirectIn<T>& in, cx<T>* out, int64_t out_ld, hipStream_t st, const RowStoreNat<T>* o) {
    const int n = in.ax.n, nseq = in.nseq;
    if (nseq <= 0 || n <= 0) return 0;
    MixPlan p;
    if (!mix_plan_for(n, p)) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d has no plan", n);
    int err = 0;
    const cx<T>* tw = twiddles<T>(n, &err);
    if (!tw) return err;
    const size_t per = size_t(p.n) * sizeof(cx<T>);
    // rows per workgroup: two (the twiddle and index work of a butterfly column is shared by nothing, but two rows give the 256 threads
    // enough butterflies per stage), more for short rows (>= 2048 points per workgroup), within 64 KiB of LDS so that two or three
    // workgroups share a CU and their load / transform / store phases overlap.  Measured (profiles/r03/exp_mix_sweep.log, us per pass):
    // 3000-point rows complex64 57.8 at 2 rows x 256 threads against 86.5 (2 x 512) and 69.4 (4 x 512)
    int seqs = (2048 + n - 1) / n;
    if (seqs < 2) seqs = 2;
    while (seqs > 1 && size_t(seqs) * per > size_t(64) * 1024) --seqs;
    if (tuning().mix_seqs > 0) seqs = tuning().mix_seqs;
    if (seqs > nseq) seqs = nseq;
    if (size_t(seqs) * per > kMixLdsHard) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d does not fit the LDS", n);
    p.seqs = seqs;
    const size_t lds = size_t(seqs) * per;
    MixRowOut<T> ro{out, out_ld, AxisMap{n, n, 0, 0}, T(1), 0, 0};
    if (o) ro = MixRowOut<T>{o->dst, o->ld, o->ax, o->scale, o->conj, 1};
    const int groups = (nseq + seqs - 1) / seqs;
    auto launch = [&](auto kernel) {
        const int rc = mix_set_lds(kernel, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kernel, dim3(groups), dim3(mix_threads(p, 256)), lds, st, p, mix_in(in), nseq, ro, tw);
        return int(hipGetLastError());
    };
    if (p.maxr <= 10) return launch(mix_rows_kernel<T, 10>);
    if (p.maxr <= 16) return launch(mix_rows_kernel<T, 16>);
    return launch(mix_rows_kernel<T, 32>);
}

template <typename T>
int mix_cols_impl(const DirectIn<T>& in, const ColStoreNat<T>& out, hipStream_t st) {
    const int n = in.ax.n, ncols = in.nseq;
    if (ncols <= 0 || n <= 0) return 0;
    MixPlan p;
    if (!mix_plan_for(n, p)) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d has no plan", n);
    int err = 0;
    const cx<T>* tw = twiddles<T>(n, &err);
    if (!tw) return err;
    const size_t per = size_t(p.n) * sizeof(cx<T>);
    // adjacent columns per workgroup (a power of two): four (32 B pieces of complex64, 64 B of complex128; the tiles of one 128 B line run
    // on one XCD, see log_g below), fewer when the LDS cannot hold four columns.  Measured (same log): 1000-point columns complex64 10.4 us
    // at 4 against 12.9 at 8 (two workgroups per CU instead of four); 3000-point columns 78 at 4 (one workgroup per CU) against 102 at 2
    int tc = 4;
    while (tc > 1 && size_t(tc) * per > kMixLdsHard) tc /= 2;
    if (tuning().mix_tc > 0) {
        tc = 1;
        while (tc * 2 <= tuning().mix_tc) tc *= 2;
    }
    while (tc > 1 && tc / 2 >= ncols) tc /= 2;
    if (size_t(tc) * per > kMixLdsHard) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d does not fit the LDS", n);
    p.seqs = tc;
    p.log_seqs = 0;
    while ((1 << p.log_seqs) < tc) ++p.log_seqs;
    const size_t lds = size_t(tc) * per;
    // tiles that share a 128 B line run on one XCD (mix_cols_kernel): 2^log_g adjacent tiles, the grid padded to whole rounds of them
    int log_g = 0;
    while ((size_t(tc) << log_g) * sizeof(cx<T>) < 128 && log_g < 3) ++log_g;
    if (tuning().mix_log_g >= 0) log_g = tuning().mix_log_g > 4 ? 4 : tuning().mix_log_g;
    const int tiles = (ncols + tc - 1) / tc, round = 8 << log_g;
    const int groups = (tiles + round - 1) / round * round;
    auto launch = [&](auto kernel) {
        const int rc = mix_set_lds(kernel, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kernel, dim3(groups), dim3(mix_threads(p)), lds, st, p, mix_in(in), ncols, out, tw, log_g);
        return int(hipGetLastError());
    };
    if (p.maxr <= 10) return launch(mix_cols_kernel<T, 10>);
    if (p.maxr <= 16) return launch(mix_cols_kernel<T, 16>);
    return launch(mix_cols_kernel<T, 32>);
}

}  // namespace pm
