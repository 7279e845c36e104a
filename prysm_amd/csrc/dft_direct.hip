// Direct O(n^2) DFT for lengths the Stockham engine does not cover (non powers of two such as
// the 9 x 12, 7 x 9 and Q = 1.5 shapes of the reference's tests, and n = 1).  Twiddles come
// from an fp64 table and the accumulation is fp64 for both precisions, so these small
// transforms are as accurate as the oracle; throughput is irrelevant here (the engine owns
// every benchmarked size).
#include "pm_internal.h"

namespace pm {

template <typename T>
__device__ __forceinline__ cx<double> direct_sum(const DirectIn<T>& in, int seq, int k, const cx<double>* tw) {
    const int n = in.ax.n;
    const cx<T>* base = in.src + int64_t(seq) * in.s_seq;
    const T* rbase = reinterpret_cast<const T*>(in.src) + int64_t(seq) * in.s_seq;
    double ar = 0.0, ai = 0.0;
    // walk the stored window; logical index i of stored element q
    int i = in.ax.unmap(0);
    int64_t idx = (int64_t(i) * k) % n;
    for (int q = 0; q < in.ax.len; ++q) {
        cx<T> x;
        if (in.real)
            x = {rbase[int64_t(q) * in.s_i], T(0)};
        else
            x = base[int64_t(q) * in.s_i];
        double xr = double(x.x), xi = in.conj ? -double(x.y) : double(x.y);
        const cx<double> w = tw[idx];
        ar += xr * w.x - xi * w.y;
        ai += xr * w.y + xi * w.x;
        ++i;
        idx += k;
        if (idx >= n) idx -= n;
        if (i >= n) {  // wrapped around the rotation: restart the phase
            i = 0;
            idx = 0;
        }
    }
    return {ar, ai};
}

template <typename T>
__global__ void direct_rows_kernel(DirectIn<T> in, cx<T>* out, int64_t out_ld, const cx<double>* tw) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int n = in.ax.n;
    if (g >= int64_t(in.nseq) * n) return;
    const int seq = int(g / n), k = int(g % n);
    const cx<double> s = direct_sum(in, seq, k, tw);
    out[int64_t(seq) * out_ld + k] = {T(s.x), T(s.y)};
}

template <typename T>
__global__ void direct_rows_out_kernel(DirectIn<T> in, RowStoreNat<T> o, const cx<double>* tw) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int n = in.ax.n;
    if (g >= int64_t(in.nseq) * n) return;
    const int seq = int(g / n), k = int(g % n);
    const int q = o.ax.map(k);
    if (q < 0) return;
    const cx<double> s = direct_sum(in, seq, k, tw);
    cx<T> v = {T(s.x * double(o.scale)), T(s.y * double(o.scale))};
    if (o.conj) v.y = -v.y;
    o.dst[int64_t(seq) * o.ld + q] = v;
}

template <typename T>
__global__ void direct_cols_kernel(DirectIn<T> in, ColStoreNat<T> o, const cx<double>* tw) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int n = in.ax.n;
    if (g >= int64_t(in.nseq) * n) return;
    // adjacent threads -> adjacent columns (coalesced along the contiguous axis)
    const int c = int(g % in.nseq), k = int(g / in.nseq);
    const cx<double> s = direct_sum(in, c, k, tw);
    store_one(o, k, c, cx<T>{T(s.x), T(s.y)});
}

template <typename T>
int direct_rows(const DirectIn<T>& in, cx<T>* out, int64_t out_ld, const cx<double>* tw, hipStream_t st) {
    const int64_t total = int64_t(in.nseq) * in.ax.n;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(direct_rows_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, st, in, out, out_ld, tw);
    return int(hipGetLastError());
}
template <typename T>
int direct_rows_out(const DirectIn<T>& in, const RowStoreNat<T>& o, const cx<double>* tw, hipStream_t st) {
    const int64_t total = int64_t(in.nseq) * in.ax.n;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(direct_rows_out_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, st, in, o, tw);
    return int(hipGetLastError());
}
template <typename T>
int direct_cols(const DirectIn<T>& in, const ColStoreNat<T>& o, const cx<double>* tw, hipStream_t st) {
    const int64_t total = int64_t(in.nseq) * in.ax.n;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(direct_cols_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, st, in, o, tw);
    return int(hipGetLastError());
}

template int direct_rows<float>(const DirectIn<float>&, cx<float>*, int64_t, const cx<double>*, hipStream_t);
template int direct_rows<double>(const DirectIn<double>&, cx<double>*, int64_t, const cx<double>*, hipStream_t);
template int direct_rows_out<float>(const DirectIn<float>&, const RowStoreNat<float>&, const cx<double>*, hipStream_t);
template int direct_rows_out<double>(const DirectIn<double>&, const RowStoreNat<double>&, const cx<double>*, hipStream_t);
template int direct_cols<float>(const DirectIn<float>&, const ColStoreNat<float>&, const cx<double>*, hipStream_t);
template int direct_cols<double>(const DirectIn<double>&, const ColStoreNat<double>&, const cx<double>*, hipStream_t);

}  // namespace pm
