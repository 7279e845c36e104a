// Kernels and launchers of the mixed-radix path (fft_mixed.h): same contracts as direct_rows / direct_cols (dft_direct.hip) and
// blue_rows / blue_cols (bluestein.hip), no scratch memory.
#include "fft_mixed.h"
#include "pm_internal.h"

namespace pm {

template <typename T>
static BlueIn<T> mix_in(const DirectIn<T>& in) {
    return BlueIn<T>{in.src, in.s_seq, in.s_i, in.ax, in.conj, in.real};
}

template <typename T>
__global__ __launch_bounds__(512) void mix_rows_kernel(MixPlan p, BlueIn<T> in, int nseq, MixRowOut<T> out, const cx<T>* __restrict__ tw) {
    extern __shared__ __align__(16) char mix_smem[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(mix_smem);
    const int seq0 = blockIdx.x * p.seqs, tid = threadIdx.x, nt = blockDim.x;
    auto fetch = [&](int sl, int i) { return seq0 + sl < nseq ? blue_fetch(in, seq0 + sl, i) : cx<T>{T(0), T(0)}; };
    auto store = [&](int sl, int k, cx<T> v) {
        if (seq0 + sl < nseq) mix_store_row(out, seq0 + sl, k, v);
    };
    for (int ph = 0; ph < p.nstage; ++ph) {
        mix_phase<T, false>(p, ph, tid, nt, lds, tw, fetch, store);
        if (ph + 1 < p.nstage) __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(512) void mix_cols_kernel(MixPlan p, BlueIn<T> in, int ncols, ColStoreNat<T> out, const cx<T>* __restrict__ tw) {
    extern __shared__ __align__(16) char mix_smem[];
    cx<T>* lds = reinterpret_cast<cx<T>*>(mix_smem);
    const int c0 = blockIdx.x * p.seqs, tid = threadIdx.x, nt = blockDim.x;
    auto fetch = [&](int sl, int i) { return c0 + sl < ncols ? blue_fetch(in, c0 + sl, i) : cx<T>{T(0), T(0)}; };
    auto store = [&](int sl, int k, cx<T> v) {
        if (c0 + sl < ncols) store_one(out, k, c0 + sl, v);
    };
    for (int ph = 0; ph < p.nstage; ++ph) {
        mix_phase<T, true>(p, ph, tid, nt, lds, tw, fetch, store);
        if (ph + 1 < p.nstage) __syncthreads();
    }
}

static constexpr size_t kMixLdsSoft = 80 * 1024, kMixLdsHard = 156 * 1024;

static int round_up64(int v) { return (v + 63) & ~63; }

// threads of a workgroup: one butterfly per thread in the stage with the most butterflies, within [64, 512]
static int mix_threads(const MixPlan& p) {
    int rmin = kMixMaxRadix;
    for (int s = 0; s < p.nstage; ++s) rmin = p.radix[s] < rmin ? p.radix[s] : rmin;
    const int most = p.seqs * (p.n / rmin);
    int nt = round_up64(most);
    // several rounds per stage: even them out
    if (nt > 512) {
        const int rounds = (most + 511) / 512;
        nt = round_up64((most + rounds - 1) / rounds);
    }
    return nt < 64 ? 64 : (nt > 512 ? 512 : nt);
}

template <typename K>
static int mix_set_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    return int(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
}

// the factorisations of every length, built on first use (a depth-first search per length: ~1 ms in all)
struct MixFactors {
    unsigned char nstage[kMixMaxN + 1];
    unsigned char radix[kMixMaxN + 1][kMixMaxStages];
    MixFactors() {
        for (int n = 0; n <= kMixMaxN; ++n) {
            int r[kMixMaxStages], ns = 0;
            nstage[n] = 0;
            if (n >= 2 && mix_factor(n, r, &ns) && ns >= 2) {
                nstage[n] = (unsigned char)ns;
                for (int s = 0; s < ns; ++s) radix[n][s] = (unsigned char)r[s];
            }
        }
    }
};
static const MixFactors& mix_factors() {
    static const MixFactors f;
    return f;
}

bool mix_length(int64_t n) { return n >= 2 && n <= kMixMaxN && mix_factors().nstage[n] != 0; }

static bool mix_plan_for(int n, MixPlan& p) {
    if (!mix_length(n)) return false;
    const MixFactors& f = mix_factors();
    int r[kMixMaxStages];
    for (int s = 0; s < f.nstage[n]; ++s) r[s] = f.radix[n][s];
    mix_fill_plan(n, r, f.nstage[n], p);
    return true;
}

template <typename T>
int mix_rows(const DirectIn<T>& in, cx<T>* out, int64_t out_ld, hipStream_t st, const RowStoreNat<T>* o) {
    const int n = in.ax.n, nseq = in.nseq;
    if (nseq <= 0 || n <= 0) return 0;
    MixPlan p;
    if (!mix_plan_for(n, p)) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d has no plan", n);
    int err = 0;
    const cx<T>* tw = twiddles<T>(n, &err);
    if (!tw) return err;
    const size_t per = size_t(p.npad) * sizeof(cx<T>);
    // rows per workgroup: as many as keep two workgroups on a CU, at least enough for 256 butterflies per stage of radix 16
    int seqs = int(kMixLdsSoft / per);
    if (seqs < 1) seqs = 1;
    if (seqs > 16) seqs = 16;
    while (seqs > 1 && (seqs - 1) * (n / 16) >= 256 && int64_t(nseq + seqs - 1) / seqs < 1024) --seqs;   // small arrays: more workgroups
    if (seqs > nseq) seqs = nseq;
    if (size_t(seqs) * per > kMixLdsHard) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d does not fit the LDS", n);
    p.seqs = seqs;
    const size_t lds = size_t(seqs) * per;
    int rc = mix_set_lds(mix_rows_kernel<T>, lds);
    if (rc) return rc;
    MixRowOut<T> ro{out, out_ld, AxisMap{n, n, 0, 0}, T(1), 0, 0};
    if (o) ro = MixRowOut<T>{o->dst, o->ld, o->ax, o->scale, o->conj, 1};
    const int groups = (nseq + seqs - 1) / seqs;
    hipLaunchKernelGGL(mix_rows_kernel<T>, dim3(groups), dim3(mix_threads(p)), lds, st, p, mix_in(in), nseq, ro, tw);
    return int(hipGetLastError());
}

template <typename T>
int mix_cols(const DirectIn<T>& in, const ColStoreNat<T>& out, hipStream_t st) {
    const int n = in.ax.n, ncols = in.nseq;
    if (ncols <= 0 || n <= 0) return 0;
    MixPlan p;
    if (!mix_plan_for(n, p)) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d has no plan", n);
    int err = 0;
    const cx<T>* tw = twiddles<T>(n, &err);
    if (!tw) return err;
    const size_t per = size_t(p.npad) * sizeof(cx<T>);
    // adjacent columns per workgroup (a power of two): two workgroups per CU if that leaves pieces of 64 B, else as wide as the LDS holds
    const int full = int(128 / sizeof(cx<T>));      // a whole 128 B line
    int tc = 1;
    while (tc < full && size_t(2 * tc) * per <= kMixLdsSoft) tc *= 2;
    while (size_t(tc) * sizeof(cx<T>) < 64 && size_t(2 * tc) * per <= kMixLdsHard) tc *= 2;
    while (tc > 1 && tc / 2 >= ncols) tc /= 2;
    if (size_t(tc) * per > kMixLdsHard) return fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d does not fit the LDS", n);
    p.seqs = tc;
    p.log_seqs = 0;
    while ((1 << p.log_seqs) < tc) ++p.log_seqs;
    const size_t lds = size_t(tc) * per;
    int rc = mix_set_lds(mix_cols_kernel<T>, lds);
    if (rc) return rc;
    const int groups = (ncols + tc - 1) / tc;
    hipLaunchKernelGGL(mix_cols_kernel<T>, dim3(groups), dim3(mix_threads(p)), lds, st, p, mix_in(in), ncols, out, tw);
    return int(hipGetLastError());
}

template int mix_rows<float>(const DirectIn<float>&, cx<float>*, int64_t, hipStream_t, const RowStoreNat<float>*);
template int mix_rows<double>(const DirectIn<double>&, cx<double>*, int64_t, hipStream_t, const RowStoreNat<double>*);
template int mix_cols<float>(const DirectIn<float>&, const ColStoreNat<float>&, hipStream_t);
template int mix_cols<double>(const DirectIn<double>&, const ColStoreNat<double>&, hipStream_t);

}  // namespace pm
