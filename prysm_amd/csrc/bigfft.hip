// Lengths n = R n' with n' a power of two the engine transforms: the powers of two ABOVE its longest transform (16384 and 32768
// per axis, R = 2 or 4) and the mixed-radix lengths 3 / 5 / 7 x 2^k (1536, 2560, 3584, 6144 ...: what scipy's next_fast_len and
// Q = 1.5 pads produce), by one radix-R step around engine transforms of length n'.  The mixed lengths used to pay Bluestein's
// convolution at the next power of two above 2n (7 - 16x the area in 2-D).  MI355X has the memory for such fields (16384^2 complex64 = 2 GiB of 288);
// before this path they fell to the O(n^2) direct kernel (12 s for 16384^2).
//
//   rows    (decimation in frequency, the input is ours to pre-process): for j < n', m < R
//               y_m[j] = (sum_r x[j + r n'] W_R^{r m}) W_n^{j m},          X[R k + m] = FFT_{n'}(y_m)[k]
//           big_pre_rows builds the R planes y_m from the caller's windowed / rotated / real / conjugated input; ONE
//           natural-order engine row pass over all planes follows (capi.hip big2d_run).
//   columns (decimation in time: the sub-sequences x[R i + r] are rows r, r + R, ... = a leading dimension of R rows):
//               F_r = FFT_{n'}(x[R i + r]),          X[k' + q n'] = sum_r (W_n^{r k'} F_r[k']) W_R^{r q}
//           big_finish combines the R sub-transforms and sends every bin through the common epilogue (store_one: scale,
//           conj, rotation / crop, multiplier, |.|^2), un-interleaving the row split (logical column R k + m of plane m).
// HBM bound; extra passes over the engine's two (pre-process, combine) are the price of sizes that are rare in practice.
#include "pm_internal.h"

namespace pm {


// v W_R^e.  R = 2 or 4: exact, a multiplication by (-i)^(e * 4 / R); other R: W_R^e = W_n^{(e mod R) n'} from the table of the
// full length (step = n' = n / R)
template <typename T>
__device__ __forceinline__ cx<T> mul_wr(cx<T> v, int e, int R, const cx<T>* twn, int64_t step) {
    if (R == 2 || R == 4) {
        const int q = (e * (4 / R)) & 3;   // quarter turns of exp(-2 pi i / 4)
        switch (q) {
            case 0: return v;
            case 1: return {v.y, -v.x};    // * (-i)
            case 2: return {-v.x, -v.y};
            default: return {-v.y, v.x};   // * (+i)
        }
    }
    const int em = e % R;
    return em ? cmul(v, twn[int64_t(em) * step]) : v;
}

// Y[m][i][j], planes of M x n' (all M LOGICAL rows: rows outside the stored window come out zero)
// R is a template parameter: with a run-time radix the arrays x[] / t[] below are indexed dynamically and live in scratch memory
// (16384^2: the pre-processing kernel then ran at 1.5 TB/s)
template <typename T, int R>
__global__ void big_pre_rows_kernel(Blue2dIn<T> in, int M, int np, cx<T>* Y, const cx<T>* twN, int row0) {
    // one row per blockIdx.y: no 64-bit division per thread (it cost this kernel half its time at 16384^2)
    const int i = row0 + int(blockIdx.y), j = int(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= np) return;
    const int64_t g = int64_t(i) * np + j;
    cx<T> x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = fetch2d(in, i, j + r * np);
    const int64_t plane = int64_t(M) * np;
#pragma unroll
    for (int m = 0; m < R; ++m) {
        cx<T> s{T(0), T(0)};
#pragma unroll
        for (int r = 0; r < R; ++r) s = s + mul_wr(x[r], r * m, R, twN, int64_t(np));
        Y[int64_t(m) * plane + g] = m ? cmul(s, twN[int64_t(j) * m]) : s;
    }
}

// F[(m * Rm + r)][k'][k], planes of mp x np;  thread (k', logical column c = Rn k + m) writes the Rm bins k' + q mp
template <typename T, int Rm>
__global__ void big_finish_kernel(const cx<T>* F, int mp, int np, int Rn, const cx<T>* twM, ColStoreNat<T> o, int row0) {
    const int N = np * Rn;
    const int kp = row0 + int(blockIdx.y), c = int(blockIdx.x * blockDim.x + threadIdx.x);      // one row of bins per blockIdx.y
    if (c >= N) return;
    const int k = int(unsigned(c) / unsigned(Rn)), m = c - k * Rn;
    const int64_t plane = int64_t(mp) * np;
    cx<T> t[Rm];
#pragma unroll
    for (int r = 0; r < Rm; ++r) {
        const cx<T> f = F[(int64_t(m) * Rm + r) * plane + int64_t(kp) * np + k];
        t[r] = r ? cmul(f, twM[int64_t(r) * kp]) : f;
    }
#pragma unroll
    for (int q = 0; q < Rm; ++q) {
        cx<T> s{T(0), T(0)};
#pragma unroll
        for (int r = 0; r < Rm; ++r) s = s + mul_wr(t[r], r * q, Rm, twM, int64_t(mp));
        store_one(o, kp + q * mp, c, s);
    }
}

template <typename T>
int big_pre_rows(const Blue2dIn<T>& in, int M, int np, int R, cx<T>* Y, const cx<T>* twN, hipStream_t st) {
    const int64_t total = int64_t(M) * np;
    if (total <= 0) return 0;
    for (int r0 = 0; r0 < M; r0 += 32768) {     // grid.y is limited to 65535: a 1-D transform may have more rows than that
        const int nr = M - r0 < 32768 ? M - r0 : 32768;
        const dim3 grid(unsigned((np + 255) / 256), unsigned(nr));
        switch (R) {
#define PM_R(r) \
    case r:     \
        hipLaunchKernelGGL((big_pre_rows_kernel<T, r>), grid, dim3(256), 0, st, in, M, np, Y, twN, r0); \
        break;
            PM_R(1) PM_R(2) PM_R(3) PM_R(4) PM_R(5) PM_R(7)
#undef PM_R
            default: return -2;
        }
    }
    return int(hipGetLastError());
}
template <typename T>
int big_finish(const cx<T>* F, int mp, int np, int Rm, int Rn, const cx<T>* twM, const ColStoreNat<T>& o, hipStream_t st) {
    const int64_t total = int64_t(mp) * np * Rn;
    if (total <= 0) return 0;
    for (int r0 = 0; r0 < mp; r0 += 32768) {
        const int nr = mp - r0 < 32768 ? mp - r0 : 32768;
        const dim3 grid(unsigned((int64_t(np) * Rn + 255) / 256), unsigned(nr));
        switch (Rm) {
#define PM_R(r) \
    case r:     \
        hipLaunchKernelGGL((big_finish_kernel<T, r>), grid, dim3(256), 0, st, F, mp, np, Rn, twM, o, r0); \
        break;
            PM_R(1) PM_R(2) PM_R(3) PM_R(4) PM_R(5) PM_R(7)
#undef PM_R
            default: return -2;
        }
    }
    return int(hipGetLastError());
}

template int big_pre_rows<float>(const Blue2dIn<float>&, int, int, int, cx<float>*, const cx<float>*, hipStream_t);
template int big_pre_rows<double>(const Blue2dIn<double>&, int, int, int, cx<double>*, const cx<double>*, hipStream_t);
template int big_finish<float>(const cx<float>*, int, int, int, int, const cx<float>*, const ColStoreNat<float>&, hipStream_t);
template int big_finish<double>(const cx<double>*, int, int, int, int, const cx<double>*, const ColStoreNat<double>&, hipStream_t);

}  // namespace pm
