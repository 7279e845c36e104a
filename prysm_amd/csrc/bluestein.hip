// Lengths that are not powers of two (up to 4096) on the FFT engine through Bluestein's identity (bluestein.h):
// chirp multiply -> zero-padded engine FFT of length MB -> multiply by the chirp's spectrum -> engine IFFT -> chirp
// multiply.  Same contracts as direct_rows / direct_cols (dft_direct.hip), O(n log n) instead of O(n^2).  The three
// pointwise passes are separate HBM-bound kernels here; the transforms are the engine's natural-order row / column
// passes with a pad window on the load and a crop window on the store.
#include "bluestein.h"
#include "pm_internal.h"

namespace pm {

template <typename T>
static BlueIn<T> blue_in(const DirectIn<T>& in) {
    return BlueIn<T>{in.src, in.s_seq, in.s_i, in.ax, in.conj, in.real};
}

// ---- rows: sequence = memory row, element j at [seq * ld + j]
template <typename T>
__global__ void blue_pre_rows_kernel(BlueIn<T> in, int nseq, cx<T>* a, const cx<T>* w) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int n = in.ax.n;
    if (g >= int64_t(nseq) * n) return;
    const int seq = int(g / n), j = int(g - int64_t(seq) * n);
    a[g] = cmul(blue_fetch(in, seq, j), w[j]);
}

// buf[seq][k] *= v[k], k < len (rows of `ld` elements)
template <typename T>
__global__ void blue_mul_rows_kernel(cx<T>* buf, int64_t ld, int nseq, int len, const cx<T>* v) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= int64_t(nseq) * len) return;
    const int seq = int(g / len), k = int(g - int64_t(seq) * len);
    cx<T>* p = buf + int64_t(seq) * ld + k;
    *p = cmul(*p, v[k]);
}

// out row `seq`, bin k through the 1-D API's output treatment (window / rotation, scale, conj): t[seq][k] * w[k]
template <typename T>
__global__ void blue_post_rows_out_kernel(const cx<T>* t, int n, const cx<T>* w, RowStoreNat<T> o) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= int64_t(o.nseq) * n) return;
    const int seq = int(g / n), k = int(g - int64_t(seq) * n);
    const int q = o.ax.map(k);
    if (q < 0) return;
    cx<T> v = cscale(cmul(t[g], w[k]), o.scale);
    if (o.conj) v.y = -v.y;
    o.dst[int64_t(seq) * o.ld + q] = v;
}

// ---- columns: sequence = column c, element i at [i * ncols + c]; adjacent threads -> adjacent columns
template <typename T>
__global__ void blue_pre_cols_kernel(BlueIn<T> in, int ncols, cx<T>* a, const cx<T>* w) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int n = in.ax.n;
    if (g >= int64_t(n) * ncols) return;
    const int i = int(g / ncols), c = int(g - int64_t(i) * ncols);
    a[g] = cmul(blue_fetch(in, c, i), w[i]);
}

// buf[k][c] *= v[k], k < len
template <typename T>
__global__ void blue_mul_cols_kernel(cx<T>* buf, int ncols, int len, const cx<T>* v) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= int64_t(len) * ncols) return;
    const int k = int(g / ncols);
    buf[g] = cmul(buf[g], v[k]);
}

// out(k, c) = epilogue(t[k][c] * w[k])
template <typename T>
__global__ void blue_post_cols_kernel(const cx<T>* t, int n, int ncols, const cx<T>* w, ColStoreNat<T> o) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= int64_t(n) * ncols) return;
    const int k = int(g / ncols), c = int(g - int64_t(k) * ncols);
    store_one(o, k, c, cmul(t[g], w[k]));
}

// ---- both axes at once (blue2d, capi.hip): a[i][j] = x(i, j) w1[i] w2[j] over the LOGICAL n1 x n2 array (windowed / rotated /
// real / conjugated input), and the epilogue out(k, c) = epilogue(t[k][c] w1[k] w2[c])
template <typename T>
__global__ void blue_pre2d_kernel(Blue2dIn<T> in, cx<T>* a, const cx<T>* w1, const cx<T>* w2) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int n1 = in.ay.n, n2 = in.ax.n;
    if (g >= int64_t(n1) * n2) return;
    const int i = int(g / n2), j = int(g - int64_t(i) * n2);
    a[g] = cmul(fetch2d(in, i, j), cmul(w1[i], w2[j]));
}

template <typename T>
__global__ void blue_post2d_kernel(const cx<T>* t, int n1, int n2, const cx<T>* w1, const cx<T>* w2, ColStoreNat<T> o) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= int64_t(n1) * n2) return;
    const int k = int(g / n2), c = int(g - int64_t(k) * n2);
    store_one(o, k, c, cmul(t[g], cmul(w1[k], w2[c])));
}

static inline dim3 grid_for(int64_t total) { return dim3(unsigned((total + 255) / 256)); }

static inline size_t align256(size_t b) { return (b + 255) & ~size_t(255); }

size_t blue_rows_scratch(size_t es, int64_t nseq, int64_t n) {
    return align256(size_t(nseq) * size_t(n) * es) + align256(size_t(nseq) * size_t(blue_conv_len(n)) * es);
}
size_t blue_cols_scratch(size_t es, int64_t ncols, int64_t n) {
    return align256(size_t(ncols) * size_t(n) * es) + align256(size_t(ncols) * size_t(blue_conv_len(n)) * es);
}

template <typename T>
int blue_rows(const DirectIn<T>& in, cx<T>* out, int64_t out_ld, void* scratch, hipStream_t st, const RowStoreNat<T>* o) {
    const int n = in.ax.n, nseq = in.nseq;
    if (nseq <= 0 || n <= 0) return 0;
    const int mb = blue_conv_len(n), lg = engine_log2(mb);
    int err = 0;
    const cx<T>* tab = blue_tables<T>(n, &err);
    if (!tab) return err;
    const cx<T>* tw = twiddles<T>(mb, &err);
    if (!tw) return err;
    const cx<T>*w = tab, *bf = tab + n;
    const int dt = sizeof(T) == 4 ? PM_C64 : PM_C128;
    cx<T>* a = reinterpret_cast<cx<T>*>(scratch);
    cx<T>* b = reinterpret_cast<cx<T>*>(static_cast<char*>(scratch) + align256(size_t(nseq) * size_t(n) * sizeof(cx<T>)));

    hipLaunchKernelGGL(blue_pre_rows_kernel<T>, grid_for(int64_t(nseq) * n), dim3(256), 0, st, blue_in(in), nseq, a, w);
    {   // A = FFT_MB(pad(a))
        RowLoadNat<T> lp{a, n, AxisMap{mb, n, 0, 0}, nseq, 0, 0};
        RowStoreNat<T> sp{b, mb, AxisMap{mb, mb, 0, 0}, nseq, 0, T(1), 0, AxisMap{1, 1, 0, 0}};
        const int rc = launch_row_nat<T>(lg, row_variant(dt, lg), lp, sp, tw, nseq, 0, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(blue_mul_rows_kernel<T>, grid_for(int64_t(nseq) * mb), dim3(256), 0, st, b, int64_t(mb), nseq, mb, bf);
    if (o) {   // 1-D API: the first n bins back into `a`, then through the output view
        RowLoadNat<T> lp{b, mb, AxisMap{mb, mb, 0, 0}, nseq, 1, 0};
        RowStoreNat<T> sp{a, n, AxisMap{mb, n, 0, 0}, nseq, 1, T(1), 0, AxisMap{1, 1, 0, 0}};
        const int rc = launch_row_nat<T>(lg, row_variant(dt, lg), lp, sp, tw, nseq, 0, st);
        if (rc) return rc;
        hipLaunchKernelGGL(blue_post_rows_out_kernel<T>, grid_for(int64_t(nseq) * n), dim3(256), 0, st, a, n, w, *o);
        return int(hipGetLastError());
    }
    {   // c = IFFT_MB(A .* B), first n bins straight into the caller's rows
        RowLoadNat<T> lp{b, mb, AxisMap{mb, mb, 0, 0}, nseq, 1, 0};
        RowStoreNat<T> sp{out, out_ld, AxisMap{mb, n, 0, 0}, nseq, 1, T(1), 0, AxisMap{1, 1, 0, 0}};
        const int rc = launch_row_nat<T>(lg, row_variant(dt, lg), lp, sp, tw, nseq, 0, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(blue_mul_rows_kernel<T>, grid_for(int64_t(nseq) * n), dim3(256), 0, st, out, out_ld, nseq, n, w);
    return int(hipGetLastError());
}

template <typename T>
int blue_cols(const DirectIn<T>& in, const ColStoreNat<T>& out, void* scratch, hipStream_t st) {
    const int n = in.ax.n, ncols = in.nseq;
    if (ncols <= 0 || n <= 0) return 0;
    const int mb = blue_conv_len(n), lg = engine_log2(mb);
    int err = 0;
    const cx<T>* tab = blue_tables<T>(n, &err);
    if (!tab) return err;
    const cx<T>* tw = twiddles<T>(mb, &err);
    if (!tw) return err;
    const cx<T>*w = tab, *bf = tab + n;
    const int dt = sizeof(T) == 4 ? PM_C64 : PM_C128;
    cx<T>* a = reinterpret_cast<cx<T>*>(scratch);
    cx<T>* b = reinterpret_cast<cx<T>*>(static_cast<char*>(scratch) + align256(size_t(ncols) * size_t(n) * sizeof(cx<T>)));
    // both scratch arrays are 256 B aligned with `ncols` elements per row: pairs of complex64 columns may be moved as
    // one 16-byte access when ncols is even
    const int vec = (sizeof(T) != 4 || ncols % 2 == 0) ? 1 : 0;
    const int tc = col_tile_width_for(dt, lg, tuning().col_var);
    const int ntiles = (ncols + tc - 1) / tc;

    hipLaunchKernelGGL(blue_pre_cols_kernel<T>, grid_for(int64_t(n) * ncols), dim3(256), 0, st, blue_in(in), ncols, a, w);
    ColStoreNat<T> cs{};
    cs.ld = ncols;
    cs.ax = AxisMap{ncols, ncols, 0, 0};
    cs.epilogue = EPI_NONE;
    cs.scale = T(1);
    cs.weight = T(1);
    cs.mul_kind = MUL_NONE;
    cs.vec_ok = vec;
    {   // A = FFT_MB(pad(a)) down the columns
        ColLoadNat<T> cl{a, ncols, AxisMap{mb, n, 0, 0}, ncols, 0, vec};
        cs.dst = b;
        cs.ay = AxisMap{mb, mb, 0, 0};
        cs.conj = 0;
        const int rc = launch_col_nat<T>(lg, 0, cl, cs, tw, ntiles, 1, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(blue_mul_cols_kernel<T>, grid_for(int64_t(mb) * ncols), dim3(256), 0, st, b, ncols, mb, bf);
    {   // c = IFFT_MB(A .* B), first n bins back into `a`
        ColLoadNat<T> cl{b, ncols, AxisMap{mb, mb, 0, 0}, ncols, 1, vec};
        cs.dst = a;
        cs.ay = AxisMap{mb, n, 0, 0};
        cs.conj = 1;
        const int rc = launch_col_nat<T>(lg, 0, cl, cs, tw, ntiles, 1, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(blue_post_cols_kernel<T>, grid_for(int64_t(n) * ncols), dim3(256), 0, st, a, n, ncols, w, out);
    return int(hipGetLastError());
}

template <typename T>
int blue_pre2d(const Blue2dIn<T>& in, cx<T>* a, const cx<T>* w1, const cx<T>* w2, hipStream_t st) {
    const int64_t total = int64_t(in.ay.n) * in.ax.n;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(blue_pre2d_kernel<T>, grid_for(total), dim3(256), 0, st, in, a, w1, w2);
    return int(hipGetLastError());
}
template <typename T>
int blue_post2d(const cx<T>* t, int n1, int n2, const cx<T>* w1, const cx<T>* w2, const ColStoreNat<T>& out, hipStream_t st) {
    const int64_t total = int64_t(n1) * n2;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(blue_post2d_kernel<T>, grid_for(total), dim3(256), 0, st, t, n1, n2, w1, w2, out);
    return int(hipGetLastError());
}

template int blue_pre2d<float>(const Blue2dIn<float>&, cx<float>*, const cx<float>*, const cx<float>*, hipStream_t);
template int blue_pre2d<double>(const Blue2dIn<double>&, cx<double>*, const cx<double>*, const cx<double>*, hipStream_t);
template int blue_post2d<float>(const cx<float>*, int, int, const cx<float>*, const cx<float>*, const ColStoreNat<float>&, hipStream_t);
template int blue_post2d<double>(const cx<double>*, int, int, const cx<double>*, const cx<double>*, const ColStoreNat<double>&, hipStream_t);
template int blue_rows<float>(const DirectIn<float>&, cx<float>*, int64_t, void*, hipStream_t, const RowStoreNat<float>*);
template int blue_rows<double>(const DirectIn<double>&, cx<double>*, int64_t, void*, hipStream_t, const RowStoreNat<double>*);
template int blue_cols<float>(const DirectIn<float>&, const ColStoreNat<float>&, void*, hipStream_t);
template int blue_cols<double>(const DirectIn<double>&, const ColStoreNat<double>&, void*, hipStream_t);

}  // namespace pm
