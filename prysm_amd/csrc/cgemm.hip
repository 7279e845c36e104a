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
//
// Everything per-thread is computed ONCE: the byte offset of each element from the (workgroup-uniform) base of the
// current K-tile, and its LDS slot.  Per K-tile the base advances by a scalar add and the loads are
// `global_load ... v_off, s[base]`; on interior tiles the registers go to LDS as they are.  (Measured with the
// timing builds -DPM_GEMM_DBG: the MFMA loop alone runs the 512 x 2048 x 2048 product in 110 us; the previous staging
// -- 64-bit pointer bumps, validity masks, select / conj multiplies, LDS address arithmetic, ~120 VALU instructions per
// K-tile and thread -- cost another 46 us because VALU work beside MFMAs steals their issue slots.)
// Conjugations are NOT applied here: they are folded into the sign of the imaginary parts at the MFMA operands and
// in the epilogue.  Out-of-range rows are clamped to the last valid row and zeroed when stored (edge tiles only);
// the K tail (last tile of a slab) takes the masked path too.
template <typename T, int ROWS, int BK, bool KFAST>
struct Stager {
    static constexpr int E = ROWS * BK / 256;
    unsigned boff[E];       // byte offset from the base of the K-tile (row block r0, first k of the tile)
    int lds[E];             // element slot inside one LDS buffer: kk * LD_S + rr
    int kk[E];              // k inside the tile (K-tail masking)
    unsigned rowmask;       // bit s: the row of element s exists
    __device__ __forceinline__ void init(int64_t ld, int64_t r0, int64_t R, int tid, int ld_s) {
        rowmask = 0;
#pragma unroll
        for (int s = 0; s < E; ++s) {
            const int e = tid + s * 256;
            kk[s] = KFAST ? e % BK : e / ROWS;
            const int rr = KFAST ? e / BK : e % ROWS;
            const bool ok = r0 + rr < R;
            rowmask |= ok ? (1u << s) : 0u;
            const int64_t rc = ok ? rr : (R - 1 - r0);        // clamp into the matrix (R > r0 always)
            const int64_t off = KFAST ? rc * ld + kk[s] : int64_t(kk[s]) * ld + rc;
            boff[s] = unsigned(off * int64_t(sizeof(cx<T>)));
            lds[s] = kk[s] * ld_s + rr;
        }
    }
    // base of a K-tile: first element of row block r0 at k = k0 (uniform over the workgroup)
    static __device__ __forceinline__ const char* tile_base(const cx<T>* X, int64_t ld, int64_t r0, int64_t k0) {
        return reinterpret_cast<const char*>(KFAST ? X + r0 * ld + k0 : X + k0 * ld + r0);
    }
    // interior tile: E unconditional loads from uniform base + per-thread offset
    __device__ __forceinline__ void fetch(cx<T> (&reg)[E], const char* base) const {
#pragma unroll
        for (int s = 0; s < E; ++s) reg[s] = *reinterpret_cast<const cx<T>*>(base + boff[s]);
    }
    // K tail: elements past kend read the tile's first k instead (always inside the matrix) and are zeroed at the store
    __device__ __forceinline__ void fetch_tail(cx<T> (&reg)[E], const char* base, int64_t ld, int krem) const {
#pragma unroll
        for (int s = 0; s < E; ++s) {
            const unsigned back = unsigned((KFAST ? int64_t(kk[s]) : int64_t(kk[s]) * ld) * int64_t(sizeof(cx<T>)));
            reg[s] = *reinterpret_cast<const cx<T>*>(base + (kk[s] < krem ? boff[s] : boff[s] - back));
        }
    }
    __device__ __forceinline__ void store(cx<T>* tile, const cx<T> (&reg)[E]) const {
#pragma unroll
        for (int s = 0; s < E; ++s) tile[lds[s]] = reg[s];
    }
    __device__ __forceinline__ void store_masked(cx<T>* tile, const cx<T> (&reg)[E], int krem) const {
#pragma unroll
        for (int s = 0; s < E; ++s) {
            const bool ok = ((rowmask >> s) & 1u) && kk[s] < krem;
            tile[lds[s]] = ok ? reg[s] : cx<T>{T(0), T(0)};
        }
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
    stA.init(lda, m0, M, tid, LDA_S);
    stB.init(ldb, n0, N, tid, LDB_S);
    cx<T> ra[SA::E], rb[SB::E];
    const bool rows_full = (m0 + BM <= M) && (n0 + BN <= N);     // uniform: no row / column of the tile is out of range
    cx<T>* const As0 = &As[0][0][0];
    cx<T>* const Bs0 = &Bs[0][0][0];
    constexpr int ABUF = BK * LDA_S, BBUF = BK * LDB_S;

    // fetch the K-tile starting at k (uniform) into registers; store the registers into LDS buffer `buf`
    auto fetch_tile = [&](int64_t k) {
        const char* ba = SA::tile_base(A, lda, m0, k);
        const char* bb = SB::tile_base(B, ldb, n0, k);
        if (k + BK <= kend) {
            stA.fetch(ra, ba);
            stB.fetch(rb, bb);
        } else {
            stA.fetch_tail(ra, ba, lda, int(kend - k));
            stB.fetch_tail(rb, bb, ldb, int(kend - k));
        }
    };
    auto store_tile = [&](int buf, int64_t k) {
        if (rows_full && k + BK <= kend) {
            stA.store(As0 + buf * ABUF, ra);
            stB.store(Bs0 + buf * BBUF, rb);
        } else {
            const int krem = int(kend - k < BK ? kend - k : BK);
            stA.store_masked(As0 + buf * ABUF, ra, krem);
            stB.store_masked(Bs0 + buf * BBUF, rb, krem);
        }
    };

    int buf = 0;
    if (kbeg < kend) {
        fetch_tile(kbeg);
        store_tile(0, kbeg);
    }
    __syncthreads();
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
#ifndef PM_GEMM_DBG
#define PM_GEMM_DBG 0   // timing builds (wrong results): 1 = no global loads / LDS staging in the loop, 2 = also no LDS operand
                        // reads, 3 = global loads but no LDS stores, 4 = LDS stores but no global loads, 5 = everything but the barrier
#endif
        if (more && (PM_GEMM_DBG == 0 || PM_GEMM_DBG == 3)) fetch_tile(k0 + BK);   // next K-tile into registers: in flight under the MFMAs of this one
        // operand fragments of k-step n + 1 are read from LDS BEFORE the MFMAs of k-step n are issued (register double
        // buffer), so the LDS latency -- longer while other workgroups stage their tiles -- hides under matrix work
        cx<T> a[2][TI], b[2][TJ];
        auto read_frags = [&](int slot, int ks) {
            const int kk = ks + MT::op_k(lane);
            if (PM_GEMM_DBG == 2) {
#pragma unroll
                for (int i = 0; i < TI; ++i) a[slot][i] = {T(lane + ks), T(i + 1)};
#pragma unroll
                for (int j = 0; j < TJ; ++j) b[slot][j] = {T(ks + 1), T(lane + j)};
            } else {
#pragma unroll
                for (int i = 0; i < TI; ++i) a[slot][i] = As[buf][kk][wr * WM + i * TM + MT::op_row(lane)];
#pragma unroll
                for (int j = 0; j < TJ; ++j) b[slot][j] = Bs[buf][kk][wc * WN + j * TM + MT::op_row(lane)];
            }
        };
        read_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < BK; ks += KS) {
            const int cur = (ks / KS) & 1;
            if (ks + KS < BK) read_frags(cur ^ 1, ks + KS);
            // conjugations: op(A) = Ar + i sa Ai, op(B) = Br + i sb Bi with sa, sb = +-1 (uniform)
            if constexpr (M3) {
                // P1 = Ar Br, P2 = Ai Bi (raw), P3 = (Ar + sa Ai)(Br + sb Bi); Cr = P1 - sa sb P2, Ci = P3 - P1 - sa sb P2
                T as[TI], bs[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) as[i] = a[cur][i].x + sa * a[cur][i].y;
#pragma unroll
                for (int j = 0; j < TJ; ++j) bs[j] = b[cur][j].x + sb * b[cur][j].y;
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        acc_r[i][j] = MT::mfma(a[cur][i].x, b[cur][j].x, acc_r[i][j]);
                        acc_i[i][j] = MT::mfma(a[cur][i].y, b[cur][j].y, acc_i[i][j]);
                        acc_3[i][j] = MT::mfma(as[i], bs[j], acc_3[i][j]);
                    }
            } else {
                T ay[TI], by[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) ay[i] = sa * a[cur][i].y;
#pragma unroll
                for (int j = 0; j < TJ; ++j) by[j] = sb * b[cur][j].y;
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        acc_r[i][j] = MT::mfma(a[cur][i].x, b[cur][j].x, acc_r[i][j]);
                        acc_i[i][j] = MT::mfma(a[cur][i].x, by[j], acc_i[i][j]);
                        acc_r[i][j] = MT::mfma(-ay[i], by[j], acc_r[i][j]);
                        acc_i[i][j] = MT::mfma(ay[i], b[cur][j].x, acc_i[i][j]);
                    }
            }
        }
        if (more && (PM_GEMM_DBG == 0 || PM_GEMM_DBG == 4)) store_tile(buf ^ 1, k0 + BK);
        if (PM_GEMM_DBG == 3 && more) {   // keep the loads alive without the LDS stores
            T sum = T(0);
#pragma unroll
            for (int s_ = 0; s_ < SA::E; ++s_) sum += ra[s_].x + rb[s_].y;
            if (sum == T(-12345.5)) As0[tid] = {sum, sum};
        }
        if (PM_GEMM_DBG != 5) __syncthreads();   // timing build 5: no barrier in the loop (wrong results)
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
                    const T p1 = cr, p2 = sa * sb * ci;
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
    if (lda >= (int64_t(1) << 22) || ldb >= (int64_t(1) << 22))   // per-thread 32-bit byte offsets inside a K-tile
        return fail(PM_ERR_UNSUPPORTED, "pm_cgemm: leading dimensions of 2^22 elements or more are not supported");
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
