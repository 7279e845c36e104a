// Complex GEMM on the CDNA4 matrix cores -- the two dense contractions of the matrix DFT
// (fttools.MDFT.__call__ / .adjoint, prysm/fttools.py:201-228).
//
//   C (M x N) = alpha * opA(A) (M x K) @ opB(B) (K x N)
//
// complex64  : v_mfma_f32_32x32x2_f32  (exact f32, 157.3 TF peak = the whole f32 rate of the chip)
// complex128 : v_mfma_f64_16x16x4_f64
//
// A complex product is four real MFMA chains on split (planar) operands:
//     Cr += Ar Br + (-Ai) Bi        Ci += Ar Bi + Ai Br
// Global memory stays interleaved (re, im); the de-interleave, the transposes of opA / opB and the
// conjugations happen once per K-tile while staging into LDS, where both operands are held
// k-major ([k][row]) so each MFMA operand read is one conflict-free ds_read of consecutive words.
//
// Work decomposition for the MDFT shapes (e.g. 512 x 2048 x 2048 then 512 x 512 x 2048): 64 x 64
// output tiles alone give only 256 / 64 workgroups for 256 CUs, so K is split (blockIdx.z) into
// slabs reduced by a second tiny kernel in a FIXED order -- deterministic, unlike atomics.
#include "pm_internal.h"

namespace pm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct MfmaTraits;

template <>
struct MfmaTraits<float> {
    static constexpr int TM = 32;   // MFMA tile edge
    static constexpr int KS = 2;    // k per instruction
    static constexpr int NR = 16;   // accumulator registers per lane
    using acc_t = f32x16;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row_of(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
    static __device__ __forceinline__ int col_of(int lane) { return lane & 31; }
    static __device__ __forceinline__ int op_row(int lane) { return lane & 31; }
    static __device__ __forceinline__ int op_k(int lane) { return lane >> 5; }
};

template <>
struct MfmaTraits<double> {
    static constexpr int TM = 16;
    static constexpr int KS = 4;
    static constexpr int NR = 4;
    using acc_t = f64x4;
    static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // f64 C/D layout differs from the f32 family: row = (lane >> 4) + 4 * reg
    static __device__ __forceinline__ int row_of(int r, int lane) { return (lane >> 4) + 4 * r; }
    static __device__ __forceinline__ int col_of(int lane) { return lane & 15; }
    static __device__ __forceinline__ int op_row(int lane) { return lane & 15; }
    static __device__ __forceinline__ int op_k(int lane) { return lane >> 4; }
};

// Staging of one operand tile (ROWS x BK complex elements, 256 threads, E = ROWS*BK/256 per thread).
// KFAST: the k index is contiguous in memory (A stored M x K, or B stored N x K) -> lanes run along k.
// Loads are UNCONDITIONAL (out-of-range coordinates are clamped to element 0 and zeroed by a select after the
// load): bounds-checked branches made the compiler wait for every load separately (vmcnt(0) x 8 per K-tile,
// i.e. the kernel ran at L2 latency); now the loads of a K-tile issue back to back and are first waited for
// when they are written to LDS, after the MFMAs of the current tile.
template <typename T, int ROWS, int BK, bool KFAST>
struct Stager {
    static constexpr int E = ROWS * BK / 256;
    const cx<T>* p[E];      // address of this thread's element of the current K-tile
    bool rok[E];            // row (m or n index) in range
    int kk[E], rr[E];       // coordinates inside the tile
    int64_t kstep;          // pointer advance per K-tile
    __device__ __forceinline__ void init(const cx<T>* X, int64_t ld, int64_t r0, int64_t R, int64_t k0, int tid) {
#pragma unroll
        for (int s = 0; s < E; ++s) {
            const int e = tid + s * 256;
            kk[s] = KFAST ? e % BK : e / ROWS;
            rr[s] = KFAST ? e / BK : e % ROWS;
            const int64_t r = r0 + rr[s];
            rok[s] = r < R;
            const int64_t rc = rok[s] ? r : 0;
            p[s] = KFAST ? X + rc * ld + (k0 + kk[s]) : X + (k0 + kk[s]) * ld + rc;
        }
        kstep = KFAST ? BK : int64_t(BK) * ld;
    }
    // k0: first k of the tile being fetched; kend: one past the last valid k; X: base (for clamping)
    // issue the loads only; `ok` remembers which of them were real.  The zeroing / conjugation happens in
    // finish(), called when the registers are written to LDS AFTER the MFMAs of the current tile, so the first
    // use of the loaded data (and the vmcnt wait the compiler puts in front of it) sits behind the matrix work.
    __device__ __forceinline__ void fetch(cx<T> (&reg)[E], unsigned& okmask, const cx<T>* X, int64_t k0, int64_t kend) {
        okmask = 0;
#pragma unroll
        for (int s = 0; s < E; ++s) {
            const bool ok = rok[s] && (k0 + kk[s] < kend);
            okmask |= ok ? (1u << s) : 0u;
            reg[s] = *(ok ? p[s] : X);
            p[s] += kstep;
        }
    }
    __device__ __forceinline__ cx<T> finish(const cx<T> (&reg)[E], unsigned okmask, int s, T conj_sign) const {
        const bool ok = (okmask >> s) & 1u;
        return {ok ? reg[s].x : T(0), ok ? reg[s].y * conj_sign : T(0)};
    }
};

template <typename T, int BM, int BN, int BK, bool AKF, bool BKF, bool M3>
__global__ void __launch_bounds__(256, 2) cgemm_kernel(int conjA, int conjB, int64_t M, int64_t N, int64_t K, int64_t ksplit,
                                                       T alpha, const cx<T>* __restrict__ A, int64_t lda,
                                                       const cx<T>* __restrict__ B, int64_t ldb, cx<T>* __restrict__ C,
                                                       int64_t ldc, int64_t slab_stride) {
    using MT = MfmaTraits<T>;
    constexpr int TM = MT::TM, KS = MT::KS, NR = MT::NR;
    constexpr int WM = BM / 2, WN = BN / 2;      // 2 x 2 waves
    constexpr int TI = WM / TM, TJ = WN / TM;    // MFMA tiles per wave
    constexpr int LDA_S = BM + 1, LDB_S = BN + 1;
    using SA = Stager<T, BM, BK, AKF>;
    using SB = Stager<T, BN, BK, BKF>;

    // LDS holds the operand tiles k-major and INTERLEAVED (re, im): one ds_read_b64 (b128 for fp64) per operand
    // per k-step delivers both MFMA inputs of a lane; rows padded by one element so the k-fast staging writes
    // of a 16-lane group fall on distinct banks.
    extern __shared__ __attribute__((aligned(16))) char pm_gemm_smem[];
    typedef cx<T> (*ATile)[BK][LDA_S];
    typedef cx<T> (*BTile)[BK][LDB_S];
    ATile As = reinterpret_cast<ATile>(pm_gemm_smem);
    BTile Bs = reinterpret_cast<BTile>(pm_gemm_smem + sizeof(cx<T>) * 2 * BK * LDA_S);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t m0 = int64_t(blockIdx.y) * BM, n0 = int64_t(blockIdx.x) * BN;
    const int64_t kbeg = int64_t(blockIdx.z) * ksplit;
    const int64_t kend = (kbeg + ksplit < K) ? kbeg + ksplit : K;
    const T sa = conjA ? T(-1) : T(1), sb = conjB ? T(-1) : T(1);

    // M3 (three-multiplication complex product, "3M"): P1 = Ar Br, P2 = Ai Bi, P3 = (Ar + Ai)(Br + Bi);
    // Cr = P1 - P2, Ci = P3 - P1 - P2 -- three MFMA chains instead of four for two extra VALU adds per operand
    // fragment; acc_r = P1, acc_i = P2 (M3) and acc_3 = P3.
    typename MT::acc_t acc_r[TI][TJ], acc_i[TI][TJ], acc_3[M3 ? TI : 1][M3 ? TJ : 1];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                acc_r[i][j][r] = T(0);
                acc_i[i][j][r] = T(0);
                if constexpr (M3) acc_3[i][j][r] = T(0);
            }

    SA stA;
    SB stB;
    stA.init(A, lda, m0, M, kbeg, tid);
    stB.init(B, ldb, n0, N, kbeg, tid);
    cx<T> ra[SA::E], rb[SB::E];
    unsigned oka = 0, okb = 0;

    auto s_store = [&](int buf) {
#pragma unroll
        for (int s = 0; s < SA::E; ++s) As[buf][stA.kk[s]][stA.rr[s]] = stA.finish(ra, oka, s, sa);
#pragma unroll
        for (int s = 0; s < SB::E; ++s) Bs[buf][stB.kk[s]][stB.rr[s]] = stB.finish(rb, okb, s, sb);
    };

    int buf = 0;
    if (kbeg < kend) {
        stA.fetch(ra, oka, A, kbeg, kend);
        stB.fetch(rb, okb, B, kbeg, kend);
        s_store(0);
    }
    __syncthreads();
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) {   // prefetch the next K-tile into registers: in flight under the MFMAs of this tile
            stA.fetch(ra, oka, A, k0 + BK, kend);
            stB.fetch(rb, okb, B, k0 + BK, kend);
        }
#pragma unroll
        for (int ks = 0; ks < BK; ks += KS) {
            const int kk = ks + MT::op_k(lane);
            cx<T> a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = As[buf][kk][wr * WM + i * TM + MT::op_row(lane)];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = Bs[buf][kk][wc * WN + j * TM + MT::op_row(lane)];
            if constexpr (M3) {
                T as[TI], bs[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) as[i] = a[i].x + a[i].y;
#pragma unroll
                for (int j = 0; j < TJ; ++j) bs[j] = b[j].x + b[j].y;
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        acc_r[i][j] = MT::mfma(a[i].x, b[j].x, acc_r[i][j]);
                        acc_i[i][j] = MT::mfma(a[i].y, b[j].y, acc_i[i][j]);
                        acc_3[i][j] = MT::mfma(as[i], bs[j], acc_3[i][j]);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        acc_r[i][j] = MT::mfma(a[i].x, b[j].x, acc_r[i][j]);
                        acc_i[i][j] = MT::mfma(a[i].x, b[j].y, acc_i[i][j]);
                        acc_r[i][j] = MT::mfma(-a[i].y, b[j].y, acc_r[i][j]);
                        acc_i[i][j] = MT::mfma(a[i].y, b[j].x, acc_i[i][j]);
                    }
            }
        }
        if (more) s_store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // epilogue: lanes of a row are adjacent columns -> contiguous interleaved stores
    cx<T>* Cout = C + int64_t(blockIdx.z) * slab_stride;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int64_t row = m0 + wr * WM + i * TM + MT::row_of(r, lane);
                const int64_t col = n0 + wc * WN + j * TM + MT::col_of(lane);
                T cr = acc_r[i][j][r], ci = acc_i[i][j][r];
                if constexpr (M3) {
                    const T p1 = cr, p2 = ci;
                    cr = p1 - p2;
                    ci = acc_3[i][j][r] - p1 - p2;
                }
                if (row < M && col < N) Cout[row * ldc + col] = {cr * alpha, ci * alpha};
            }
}

template <typename T>
__global__ void splitk_reduce_kernel(int64_t M, int64_t N, int S, T alpha, const cx<T>* __restrict__ slabs,
                                     int64_t slab_stride, cx<T>* __restrict__ C, int64_t ldc) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= M * N) return;
    const int64_t r = g / N, c = g % N;
    T sr = T(0), si = T(0);
    for (int s = 0; s < S; ++s) {   // fixed order: bitwise reproducible
        const cx<T> v = slabs[int64_t(s) * slab_stride + g];
        sr += v.x;
        si += v.y;
    }
    C[r * ldc + c] = {sr * alpha, si * alpha};
}

static int gemm_bm(int64_t M, int dtype = PM_C64) {
    if (dtype != PM_C64) return 64;   // rows per workgroup tile: 128 (two MFMA tiles per wave along M) when M allows it
    const int t = tuning().gemm_bm;
    if (t == 64 || t == 128) return (t == 128 && M >= 128) ? 128 : 64;
    return 64;
}

size_t cgemm_workspace_bytes(int dtype, int64_t M, int64_t N, int64_t K, int* S_out) {
    const int BM = gemm_bm(M, dtype), BN = 64, BK = 32;   // slab depth multiple of the deepest K-tile
    const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    int S = 1;
    // aim for >= 2 workgroups per CU (512) while keeping >= 8 K-tiles per slab
    while (tiles * S < tuning().gemm_min_wgs && (K / (S * 2)) >= 4 * BK && S < 32) S *= 2;
    if (S_out) *S_out = S;
    if (S == 1) return 0;
    return size_t(S) * size_t(M) * size_t(N) * (dtype == PM_C64 ? 8 : 16);
}

template <typename T, int BK, int BM>
int cgemm_ws_bm(int opA, int opB, int64_t M, int64_t N, int64_t K, double alpha, const cx<T>* A, int64_t lda,
                const cx<T>* B, int64_t ldb, cx<T>* C, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int BN = 64;
    int S = 1;
    const size_t need = cgemm_workspace_bytes(sizeof(T) == 4 ? PM_C64 : PM_C128, M, N, K, &S);
    if (S > 1 && (!ws || ws_bytes < need)) S = 1;   // no workspace: fall back to unsplit (still correct)
    int64_t ksplit = (K + S - 1) / S;
    ksplit = (ksplit + BK - 1) / BK * BK;
    if (ksplit < BK) ksplit = BK;
    S = int((K + ksplit - 1) / ksplit);
    if (S < 1) S = 1;
    dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((M + BM - 1) / BM), (unsigned)S);
    // op 0/1: A stored M x K (k contiguous), B stored K x N (n contiguous); op 2/3: transposed storage
    const bool akf = !(opA & 2), bkf = (opB & 2) != 0;
    const int cA = opA & 1, cB = opB & 1;
    auto launch = [&](T al, cx<T>* out, int64_t ldo, int64_t slab) {
        constexpr size_t LDSB = sizeof(cx<T>) * 2 * BK * ((BM + 1) + (BN + 1));
#define PM_GEMM_(AK, BK_, M3_)                                                                                             \
    {                                                                                                                      \
        auto kern = cgemm_kernel<T, BM, BN, BK, AK, BK_, M3_>;                                                             \
        if (LDSB > 48 * 1024)                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB)); \
        hipLaunchKernelGGL(kern, grid, dim3(256), LDSB, st, cA, cB, M, N, K, ksplit, al, A, lda, B, ldb, out, ldo, slab);  \
    }
#define PM_GEMM(AK, BK_)                     \
    {                                        \
        if (tuning().gemm_3m) {              \
            PM_GEMM_(AK, BK_, true)          \
        } else {                             \
            PM_GEMM_(AK, BK_, false)         \
        }                                    \
    }
        if (akf && bkf) {
            PM_GEMM(true, true)
        } else if (akf) {
            PM_GEMM(true, false)
        } else if (bkf) {
            PM_GEMM(false, true)
        } else {
            PM_GEMM(false, false)
        }
#undef PM_GEMM
#undef PM_GEMM_
    };
    if (S == 1) {
        launch(T(alpha), C, ldc, int64_t(0));
        return int(hipGetLastError());
    }
    cx<T>* slabs = reinterpret_cast<cx<T>*>(ws);
    launch(T(1), slabs, N, M * N);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return int(e);
    const int64_t total = M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, M, N, S, T(alpha),
                       slabs, M * N, C, ldc);
    return int(hipGetLastError());
}

template <typename T, int BK>
int cgemm_ws_bk(int opA, int opB, int64_t M, int64_t N, int64_t K, double alpha, const cx<T>* A, int64_t lda,
                const cx<T>* B, int64_t ldb, cx<T>* C, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st) {
    if constexpr (sizeof(T) == 4) {   // fp64: 16x16 MFMA tiles, four per wave along M would spill
        if (gemm_bm(M) == 128) return cgemm_ws_bm<T, BK, 128>(opA, opB, M, N, K, alpha, A, lda, B, ldb, C, ldc, ws, ws_bytes, st);
    }
    return cgemm_ws_bm<T, BK, 64>(opA, opB, M, N, K, alpha, A, lda, B, ldb, C, ldc, ws, ws_bytes, st);
}

template <typename T>
int cgemm_ws(int opA, int opB, int64_t M, int64_t N, int64_t K, double alpha, const cx<T>* A, int64_t lda,
             const cx<T>* B, int64_t ldb, cx<T>* C, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st) {
    // K-tile depth: 16 (fp32) / 8 (fp64) = 33 KiB of LDS, 4 workgroups per CU; tuning gemm_bk = 32 / 16 doubles it
    // (66.5 KiB, 2 per CU, half the barriers per flop)
    const int bk = tuning().gemm_bk;
    if (sizeof(T) == 4) {
        if (bk == 32) return cgemm_ws_bk<T, 32>(opA, opB, M, N, K, alpha, A, lda, B, ldb, C, ldc, ws, ws_bytes, st);
        return cgemm_ws_bk<T, 16>(opA, opB, M, N, K, alpha, A, lda, B, ldb, C, ldc, ws, ws_bytes, st);
    }
    if (bk == 32) return cgemm_ws_bk<T, 16>(opA, opB, M, N, K, alpha, A, lda, B, ldb, C, ldc, ws, ws_bytes, st);
    return cgemm_ws_bk<T, 8>(opA, opB, M, N, K, alpha, A, lda, B, ldb, C, ldc, ws, ws_bytes, st);
}

}  // namespace pm

using namespace pm;

extern "C" {

size_t pm_cgemm_workspace(int32_t dtype, int64_t M, int64_t N, int64_t K) {
    return cgemm_workspace_bytes(dtype, M, N, K, nullptr);
}

int pm_cgemm(int32_t dtype, int32_t opA, int32_t opB, int64_t M, int64_t N, int64_t K, double alpha, const void* A,
             int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, void* workspace, size_t workspace_bytes,
             void* stream) {
    if (!A || !B || !C) return fail(PM_ERR_ARG, "pm_cgemm: null buffer");
    if (M < 0 || N < 0 || K < 0 || opA < 0 || opA > 3 || opB < 0 || opB > 3) return fail(PM_ERR_ARG, "pm_cgemm: bad argument");
    if (M == 0 || N == 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == PM_C64)
        return cgemm_ws<float>(opA, opB, M, N, K, alpha, (const cx<float>*)A, lda, (const cx<float>*)B, ldb, (cx<float>*)C, ldc,
                               workspace, workspace_bytes, st);
    if (dtype == PM_C128)
        return cgemm_ws<double>(opA, opB, M, N, K, alpha, (const cx<double>*)A, lda, (const cx<double>*)B, ldb, (cx<double>*)C,
                                ldc, workspace, workspace_bytes, st);
    return fail(PM_ERR_ARG, "pm_cgemm: dtype must be PM_C64 or PM_C128");
}

}  // extern "C"
