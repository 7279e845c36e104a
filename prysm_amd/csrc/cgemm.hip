// Complex GEMM on the CDNA4 matrix cores -- the two dense contractions of the matrix DFT
// (fttools.MDFT.__call__ / .adjoint, prysm/fttools.py:201-228).
//
//   C (M x N) = alpha * opA(A) (M x K) @ opB(B) (K x N)
//
// complex64  : v_mfma_f32_32x32x2_f32  (exact f32, 157.3 TF peak = the whole f32 rate of the chip)
// complex128 : v_mfma_f64_16x16x4_f64
//
// A complex product is THREE real MFMA chains (the "3M" form: P1 = Ar Br, P2 = Ai Bi, P3 = (Ar + Ai)(Br + Bi);
// Cr = P1 - P2, Ci = P3 - P1 - P2; knob gemm_3m = 0: the four-chain form).  Operands stay interleaved (re, im) in global
// memory and in LDS; conjugations are signs on the imaginary operands and in the epilogue.
//
// Two kernels:
//  * cgemm_dma_kernel -- complex64, shapes that are multiples of 64 x 64 x 16 with 16-byte aligned operands (every matrix-DFT
//    product of the BASELINE configs): operand tiles go global -> LDS by LDS-DMA, three LDS buffers, one barrier per K-tile in
//    the middle of its MFMA stream, fragments read a quarter tile ahead.  64 x 64 or 128 x 128 workgroup tiles.
//  * cgemm_kernel -- everything else (complex128 on v_mfma_f64_16x16x4_f64, ragged shapes, unaligned operands): register-staged
//    64 x 64 tiles with masked edges.
// When the output has too few tiles for 256 CUs, K is split into slabs reduced by a second small kernel in a FIXED order --
// deterministic, unlike atomics.
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
// `global_load ... v_off, s[base]`; on interior tiles the registers go to LDS as they are.
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
        if (more) fetch_tile(k0 + BK);   // next K-tile into registers: in flight under the MFMAs of this one
        // operand fragments of k-step n + 1 are read from LDS BEFORE the MFMAs of k-step n are issued (register double
        // buffer), so the LDS latency -- longer while other workgroups stage their tiles -- hides under matrix work
        cx<T> a[2][TI], b[2][TJ];
        auto read_frags = [&](int slot, int ks) {
            const int kk = ks + MT::op_k(lane);
#pragma unroll
            for (int i = 0; i < TI; ++i) a[slot][i] = As[buf][kk][wr * WM + i * TM + MT::op_row(lane)];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[slot][j] = Bs[buf][kk][wc * WN + j * TM + MT::op_row(lane)];
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
        if (more) store_tile(buf ^ 1, k0 + BK);
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
                    const T p1 = cr, p2 = sa * sb * ci;
                    cr = p1 - p2;
                    ci = acc_3[i][j][r] - p1 - p2;
                }
                if (row < M && col < N) Cout[row * ldc + col] = {cr * alpha, ci * alpha};
            }
}

// ---------------------------------------------------------------------------------------------------------------
// complex64, large aligned shapes (the matrix-DFT products): 128 x 128 workgroup tiles, 64 x 64 per wave.
//
// Why a second kernel: with 32 x 32 per wave (above) a K-tile of 16 is 24 MFMAs = 0.64 us, shorter than the round trip of
// the global loads that must land before the next LDS stage, and every K-tile ends in a barrier: the MFMA pipe measured
// 61.6 % busy (profiles/r01/gemm_sq_counters.txt).  Here a wave owns 2 x 2 MFMA tiles and the three chains of the 3M
// product: 96 MFMAs = 2.6 us per K-tile and barrier, four 16-byte LDS reads per 24 MFMAs, and the next K-tile's global loads
// have the whole K-tile to arrive.  192 accumulator registers per lane, so ONE workgroup of four waves per CU (a wave has
// the 512-register file of its SIMD to itself).
//
// K inside a K-tile is PERMUTED: lane half h = lane >> 5 of an MFMA takes k = 8 h + s at step s (s < 8), for A and B alike,
// so the sum over the tile is the same and a lane's eight k values of one row are contiguous in LDS ([row][k], rows padded to
// 18 elements = 144 B: the 16-byte reads of 16 consecutive rows fall on 64 distinct banks).  Operands stay interleaved
// (re, im); conjugations are signs (sa, sb) applied to the 3M sums and in the epilogue.
// Requirements (host-checked, else the kernel above runs): M % 128 == 0, N % 128 == 0, slab depth % 16 == 0, 16-byte aligned
// bases, even leading dimensions.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// One operand tile (ROWS rows x 16 k, ROWS = 64 or 128) in LDS, filled by LDS-DMA (global_load_lds_dwordx4: no staging
// registers, no ds_write -- the eight ds_write_b128 per thread and K-tile of a register-staged version cost the MFMA stream
// 1600 of its 6144 cycles, experiments/scripts/exp_mfma2.cpp).  A DMA piece lands lane-linear (wave-uniform base + 16 B x lane), so the image is:
//   KFAST (memory [row][k]):  [row][8 slots of 16 B], slot = kpair ^ ((row >> 1) & 7) -- the swizzle is applied to the SOURCE
//                             address of each lane and again when reading; 16-byte fragment reads of the 16-lane LDS groups are
//                             conflict-free.  Piece j = rows 8 j .. 8 j + 7.
//   !KFAST (memory [k][row]): [k][ROWS rows] as in memory; a piece is 1 KiB of that image (one k of 128 rows, two of 64);
//                             fragments by 8-byte reads.
// ROWS / 8 pieces per tile, ROWS / 32 per wave.
// ROWS / 8 pieces per tile, shared by the NWS waves that stage it (ROWS / (8 NWS) each).
template <int ROWS, bool KFAST, int NWS>
struct DmaTile {
    static constexpr int PPW = ROWS / 8 / NWS;   // pieces per wave
    static_assert(PPW >= 1 && PPW * 8 * NWS == ROWS, "tile rows must divide over the staging waves");
    unsigned goff[PPW];   // this lane's source byte offset from the tile base, per piece
    __device__ __forceinline__ void init(int64_t ld, int wave, int lane) {
#pragma unroll
        for (int s = 0; s < PPW; ++s) {
            const int j = wave * PPW + s;
            if (KFAST) {
                const int row = 8 * j + (lane >> 3), slot = lane & 7;
                const int c = slot ^ ((row >> 1) & 7);
                goff[s] = unsigned((int64_t(row) * ld + 2 * c) * 8);
            } else {
                const int byte = j * 1024 + lane * 16;
                const int k = byte / (ROWS * 8), rp = (byte % (ROWS * 8)) / 16;
                goff[s] = unsigned((int64_t(k) * ld + 2 * rp) * 8);
            }
        }
    }
    __device__ __forceinline__ void issue(const char* base, char* tile, int wave) const {
#pragma unroll
        for (int s = 0; s < PPW; ++s)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + goff[s]),
                                             (__attribute__((address_space(3))) void*)(tile + (wave * PPW + s) * 1024), 16, 0, 0);
    }
};

// TI x TJ MFMA tiles (32 x 32) per wave; the four waves of a workgroup are arranged WM x WN x WK (rows, columns, K):
//   (2, 2) tiles, 2 x 2 x 1: 128 x 128, one workgroup per CU (192 accumulator registers per lane), the large-GEMM shape;
//   (1, 1) tiles, 2 x 2 x 1:  64 x 64;
//   (1, 1) tiles, 2 x 1 x 2:  64 x 32, the two wave PAIRS each take half of the K range            (experiment builds: lost)
//   (1, 1) tiles, 1 x 1 x 4:  32 x 32, every wave a quarter of the K range                         (experiment builds: no better than slabs)
//   (1, 1) tiles, 2 x 2 x 2:  64 x 64 with EIGHT waves -- two K-groups that are each the plain 64 x 64 arrangement (same staging cost
//                             per wave), one workgroup per CU.
// WK > 1 is split-K INSIDE the workgroup: each K-group has its own three-deep LDS ring (A rows 32 TI WM, B rows 32 TJ WN), the groups
// walk their K ranges in lockstep (same barriers) and their partial sums meet through LDS at the end, added in K order -- fixed
// order, bitwise reproducible, no slab round trip through HBM and no second launch.  Shipped: the eight-wave form for outputs of 256 ..
// 511 tiles (config 4's first product, 512 x 2048 x 2048: 108.9 us against 108.2 + 6.2 for two slabs + splitk_reduce_kernel); smaller
// outputs (its second product, 64 tiles) keep the slabs -- the 32 x 32 four-K-group form measured equal at K = 2048 and slower at 4096.
// EPI = 1: the epilogue stores w |alpha c|^2 into (or adds it to) a REAL matrix instead of the complex result (the incoherent sum of
// the polychromatic recipe: focus_dft + intensity + weighted accumulate without the 512^2 complex round trip).
template <int TI, int TJ, int WM, int WN, int WK, bool AKF, bool BKF, int EPI>
__global__ void __launch_bounds__(64 * WM * WN * WK) cgemm_dma_kernel(int conjA, int conjB, int ntm, int ntn, int64_t K, int64_t ksplit, float alpha,
                                                       const cx<float>* __restrict__ A, int64_t lda, const cx<float>* __restrict__ B,
                                                       int64_t ldb, cx<float>* __restrict__ C, int64_t ldc, int64_t slab_stride,
                                                       float weight, int accumulate) {
    static_assert(WM * WN * WK == 4 || WM * WN * WK == 8, "four waves, or eight: two K-groups of the 2 x 2 arrangement");
    constexpr int BK = 16, BM = 32 * TI * WM, BN = 32 * TJ * WN, NBUF = 3, NWS = WM * WN;
    constexpr int ABYTES = BM * 128, BBYTES = BN * 128, SGBYTES = NBUF * (ABYTES + BBYTES);
    extern __shared__ __attribute__((aligned(16))) char pm_gemm_smem[];

    // XCD-aware order: workgroups b, b + 8, ... run on one XCD; give each XCD a CONTIGUOUS run of the (slab, n-tile, m-tile)
    // list, m fastest, so the workgroups that share a B slab (and then an A slab) share that XCD's L2
    int l = blockIdx.x;
    const int nb = gridDim.x;
    if ((nb & 7) == 0) l = (l & 7) * (nb >> 3) + (l >> 3);
    const int tiles = ntm * ntn;
    const int slab = l / tiles, rr = l - slab * tiles;
    const int tn = rr / ntm, tm = rr - tn * ntm;
    const int64_t m0 = int64_t(tm) * BM, n0 = int64_t(tn) * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sg = wave / NWS, ws = wave - sg * NWS;       // K-group, wave inside it
    const int wr = ws / WN, wc = ws - wr * WN;
    const int row = lane & 31, h = lane >> 5;
    const float sa = conjA ? -1.f : 1.f, sb = conjB ? -1.f : 1.f;
    char* const As = pm_gemm_smem + sg * SGBYTES;           // [3] operand images of this K-group
    char* const Bs = As + NBUF * ABYTES;

    // K range of this workgroup's slab, then of this K-group inside it (the host makes both multiples of 16)
    const int64_t kslab = int64_t(slab) * ksplit;
    const int64_t kslab_end = (kslab + ksplit < K) ? kslab + ksplit : K;
    const int64_t kper = (kslab_end - kslab) / WK;
    const int64_t kbeg = kslab + sg * kper;
    const int nkt = int(kper / BK);

    DmaTile<BM, AKF, NWS> dA;
    DmaTile<BN, BKF, NWS> dB;
    dA.init(lda, ws, lane);
    dB.init(ldb, ws, lane);
    const char* pa = reinterpret_cast<const char*>(AKF ? A + m0 * lda + kbeg : A + kbeg * lda + m0);
    const char* pb = reinterpret_cast<const char*>(BKF ? B + n0 * ldb + kbeg : B + kbeg * ldb + n0);
    const int64_t stepA = (AKF ? int64_t(BK) : int64_t(BK) * lda) * 8, stepB = (BKF ? int64_t(BK) : int64_t(BK) * ldb) * 8;

    f32x16 p1[TI][TJ], p2[TI][TJ], p3[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) p1[i][j][r] = p2[i][j][r] = p3[i][j][r] = 0.f;

    // K inside a K-tile is permuted: lane half h takes k = 8 h + 2 q + e at step (quarter q, e), for both operands alike.
    // fragment byte offsets inside an operand image, row block 0 (block 1: + 32 rows):
    //   KFAST : row * 128 + ((4 h + q) ^ swz) * 16, swz = (row >> 1) & 7 (the row-block offsets are multiples of 32 rows: same swz)
    //   !KFAST: (8 h + 2 q + e) * ROWS * 8 + row * 8
    const int swz = (row >> 1) & 7;
    const int ra0 = AKF ? (wr * 32 * TI + row) * 128 + ((4 * h) ^ (swz & 4)) * 16 : h * 8 * BM * 8 + (wr * 32 * TI + row) * 8;
    const int rb0 = BKF ? (wc * 32 * TJ + row) * 128 + ((4 * h) ^ (swz & 4)) * 16 : h * 8 * BN * 8 + (wc * 32 * TJ + row) * 8;
    int qoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) qoff[q] = (q ^ (swz & 3)) * 16;
    // one quarter of fragments = the (re, im) of 2 k values per row block and operand: fr[slot][block] = (k0.re, k0.im, k1.re, k1.im)
    f32x4 fa[2][TI], fb[2][TJ];
    auto read_quarter = [&](int slot, int bufi, int q) {
        const char* at = As + bufi * ABYTES + ra0;
        const char* bt = Bs + bufi * BBYTES + rb0;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            if constexpr (AKF) {
                fa[slot][i] = *reinterpret_cast<const f32x4*>(at + i * 32 * 128 + qoff[q]);
            } else {
                const f32x2 u = *reinterpret_cast<const f32x2*>(at + i * 32 * 8 + (2 * q) * BM * 8);
                const f32x2 v = *reinterpret_cast<const f32x2*>(at + i * 32 * 8 + (2 * q + 1) * BM * 8);
                fa[slot][i] = f32x4{u[0], u[1], v[0], v[1]};
            }
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            if constexpr (BKF) {
                fb[slot][j] = *reinterpret_cast<const f32x4*>(bt + j * 32 * 128 + qoff[q]);
            } else {
                const f32x2 u = *reinterpret_cast<const f32x2*>(bt + j * 32 * 8 + (2 * q) * BN * 8);
                const f32x2 v = *reinterpret_cast<const f32x2*>(bt + j * 32 * 8 + (2 * q + 1) * BN * 8);
                fb[slot][j] = f32x4{u[0], u[1], v[0], v[1]};
            }
        }
    };

    // Pipeline (one barrier per K-tile, in the MIDDLE of its MFMA stream; two tiles of DMA in flight):
    //   tile t lives in LDS buffer t % 3.  In the middle of tile t: wait for this wave's DMA pieces of tile t + 1 (issued a whole
    //   tile ago), barrier -- every wave's pieces of tile t + 1 are now in LDS and every wave has left tile t - 1 --, then issue
    //   the DMA of tile t + 2 into buffer (t + 2) % 3 = the buffer of tile t - 1.  Fragments are read one quarter ahead of their
    //   use, the first quarter of tile t + 1 during the last quarter of tile t.
    if (nkt > 0) {
        dA.issue(pa, As, ws);
        dB.issue(pb, Bs, ws);
    }
    if (nkt > 1) {
        pa += stepA;
        pb += stepB;
        dA.issue(pa, As + ABYTES, ws);
        dB.issue(pb, Bs + BBYTES, ws);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();    // tiles 0 and 1 are in LDS
    if (nkt > 0) read_quarter(0, 0, 0);
    int buf = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const int nbuf = buf == NBUF - 1 ? 0 : buf + 1;
        const int pbuf = buf == 0 ? NBUF - 1 : buf - 1;     // = (kt + 2) % 3
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cur = q & 1;
            if (q < 3) read_quarter(cur ^ 1, buf, q + 1);
            else if (kt + 1 < nkt) read_quarter(0, nbuf, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 2; ++e) {     // the two k values of the quarter
                float ax[TI], ay[TI], as[TI], bx[TJ], by[TJ], bs[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    ax[i] = fa[cur][i][2 * e];
                    ay[i] = fa[cur][i][2 * e + 1];
                    as[i] = ax[i] + sa * ay[i];
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    bx[j] = fb[cur][j][2 * e];
                    by[j] = fb[cur][j][2 * e + 1];
                    bs[j] = bx[j] + sb * by[j];
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        p1[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[i], bx[j], p1[i][j], 0, 0, 0);
                        p2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[i], by[j], p2[i][j], 0, 0, 0);
                        p3[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[i], bs[j], p3[i][j], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q == 1 && kt + 1 < nkt) {
                // this wave's pieces of tile kt + 1 have landed (hipcc does not count an LDS-DMA issued in the previous trip of
                // the loop: the wait is written out), then the barrier: everybody's have
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (kt + 2 < nkt) {
                    pa += stepA;
                    pb += stepB;
                    dA.issue(pa, As + pbuf * ABYTES, ws);
                    dB.issue(pb, Bs + pbuf * BBYTES, ws);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        buf = nbuf;
    }

    if constexpr (WK > 1) {
        // partial sums of the K-groups meet through LDS (the rings are done with): group g > 0 parks its 48 TI TJ accumulator
        // registers lane-contiguously, group 0 adds them in the order g = 1, 2, ... and stores
        float* red = reinterpret_cast<float*>(pm_gemm_smem);
        constexpr int PER = TI * TJ * 48 * 64;       // floats per wave
        __syncthreads();
        if (sg > 0) {
            float* mine = red + ((sg - 1) * NWS + ws) * PER + lane;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int at = ((i * TJ + j) * 48 + r) * 64;
                        mine[at] = p1[i][j][r];
                        mine[at + 16 * 64] = p2[i][j][r];
                        mine[at + 32 * 64] = p3[i][j][r];
                    }
        }
        __syncthreads();
        if (sg > 0) return;
#pragma unroll
        for (int g = 1; g < WK; ++g) {
            const float* theirs = red + ((g - 1) * NWS + ws) * PER + lane;
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int at = ((i * TJ + j) * 48 + r) * 64;
                        p1[i][j][r] += theirs[at];
                        p2[i][j][r] += theirs[at + 16 * 64];
                        p3[i][j][r] += theirs[at + 32 * 64];
                    }
        }
    }
    // epilogue: Cr = P1 - sa sb P2, Ci = P3 - P1 - sa sb P2; lanes of a row are adjacent columns (256 B runs)
    cx<float>* Cout = C + int64_t(slab) * slab_stride;
    const float ss = sa * sb;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t orow = m0 + wr * 32 * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int64_t ocol = n0 + wc * 32 * TJ + j * 32 + row;
                const float q2 = ss * p2[i][j][r];
                const float cr = (p1[i][j][r] - q2) * alpha, ci = (p3[i][j][r] - p1[i][j][r] - q2) * alpha;
                if constexpr (EPI == 1) {
                    float* R = reinterpret_cast<float*>(C) + orow * ldc + ocol;
                    const float i2 = weight * (cr * cr + ci * ci);
                    *R = accumulate ? *R + i2 : i2;
                } else {
                    Cout[orow * ldc + ocol] = {cr, ci};
                }
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

// ... with the |.|^2 epilogue of pm_cgemm_abs2: R = (accumulate ? R : 0) + weight |alpha sum|^2
template <typename T>
__global__ void splitk_reduce_abs2_kernel(int64_t M, int64_t N, int S, T alpha, const cx<T>* __restrict__ slabs,
                                          int64_t slab_stride, T* __restrict__ R, int64_t ldr, T weight, int accumulate) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= M * N) return;
    const int64_t r = g / N, c = g % N;
    T sr = T(0), si = T(0);
    for (int s = 0; s < S; ++s) {
        const cx<T> v = slabs[int64_t(s) * slab_stride + g];
        sr += v.x;
        si += v.y;
    }
    sr *= alpha;
    si *= alpha;
    const T i2 = weight * (sr * sr + si * si);
    T* o = R + r * ldr + c;
    *o = accumulate ? *o + i2 : i2;
}

static int gemm_bm(int64_t M, int dtype = PM_C64) {
    if (dtype != PM_C64) return 64;   // rows per workgroup tile: 128 (two MFMA tiles per wave along M) when M allows it
    const int t = tuning().gemm_bm;
    if (t == 64 || t == 128) return (t == 128 && M >= 128) ? 128 : 64;
    return 64;
}

// plan of the LDS-DMA kernel: workgroup tile and wave arrangement, and the split of K across workgroups (slabs reduced in a fixed
// order by splitk_reduce_kernel) only when even 32 x 32 tiles leave CUs idle
struct DmaPlan {
    int tm, tn;    // workgroup tile: 128 x 128, 64 x 64 (128 x 64 measured in between: profiles/r02/exp_gemm_shapes.log), 64 x 32 or 32 x 32
    int wk;        // K-groups inside the workgroup (1, 2 with 64 x 32, 4 with 32 x 32)
    int S;
    int64_t ksplit;
};
static bool gemm_dma_plan(int64_t M, int64_t N, int64_t K, DmaPlan* out) {
    if (!tuning().gemm_dma || !tuning().gemm_3m || M < 64 || N < 64 || (M % 64) || (N % 64) || K < 16 || (K % 16)) return false;
    DmaPlan p{64, 64, 1, 1, K};
    // workgroups the launch aims for (128 x 128 tiles from that many, else K split across workgroups until it is reached): 512 measured
    // best on config 4's products (2^31 and 2^29 multiply-adds); the 2^26 .. 2^28 products of a 1024^2 <-> 256^2 model are launch bound and
    // take 256 -- half the slabs for the reduce launch to add (profiles/r06/exp_model7_knobs.log: the seven-plane model 131.3 us at 512,
    // 126.7 at 256, 189.8 at 128).  A knob set by hand is taken as it is.
    int want = tuning().gemm_dma_wgs;
    if (want == 512 && M * N * K <= (int64_t(1) << 28)) want = 256;
    const bool m128 = (M % 128) == 0, n128 = (N % 128) == 0;
    const int64_t t64 = (M / 64) * (N / 64);
    const int wkm = tuning().gemm_wk;     // bit 0: 64 x 64 with two K-groups (8 waves), bit 1: 64 x 32 with two, bit 2: 32 x 32 with four
    if (m128 && n128 && (M / 128) * (N / 128) >= want) p.tm = p.tn = 128;
    else if ((wkm & 1) && t64 < want && 2 * t64 >= want && (K % 32) == 0) p.wk = 2;                                  // 64 x 64, K halves
    const int t = tuning().gemm_tile;
    if (t == 64) { p.tm = p.tn = 64; p.wk = 1; }
    else if (t == 128 && m128 && n128) { p.tm = p.tn = 128; p.wk = 1; }
    const int64_t tiles = (M / p.tm) * (N / p.tn);
    // split K across workgroups until the launch fills the 256 CUs, keeping at least 4 K-tiles per slab (the in-workgroup forms exist
    // to avoid this: they are only chosen when they fill the chip by themselves)
    if (p.wk == 1)
        while (tiles * p.S < want && K / (p.S * 2) >= 64 && ((K / (p.S * 2)) % 16) == 0 && p.S < 64) p.S *= 2;
    p.ksplit = K / p.S;
    if (out) *out = p;
    return true;
}

static size_t cgemm_workspace_bytes_64(int dtype, int64_t M, int64_t N, int64_t K, int* S_out);
size_t cgemm_workspace_bytes(int dtype, int64_t M, int64_t N, int64_t K, int* S_out) {
    // the caller's operands decide between the two kernels (alignment), so the query covers both
    size_t a = cgemm_workspace_bytes_64(dtype, M, N, K, S_out), b = 0;
    DmaPlan dp;
    if (dtype == PM_C64 && gemm_dma_plan(M, N, K, &dp) && dp.S > 1) b = size_t(dp.S) * size_t(M) * size_t(N) * 8;
    return a > b ? a : b;
}
static size_t cgemm_workspace_bytes_64(int dtype, int64_t M, int64_t N, int64_t K, int* S_out) {
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
    const size_t need = cgemm_workspace_bytes_64(sizeof(T) == 4 ? PM_C64 : PM_C128, M, N, K, &S);
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
#define PM_GEMM_3M_OR_4(AK, BK_) PM_GEMM_(AK, BK_, true)
#define PM_GEMM(AK, BK_)                     \
    {                                        \
        PM_GEMM_3M_OR_4(AK, BK_)             \
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
#undef PM_GEMM_3M_OR_4
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
    return cgemm_ws_bm<T, BK, 64>(opA, opB, M, N, K, alpha, A, lda, B, ldb, C, ldc, ws, ws_bytes, st);
}

struct DmaEpi {
    int kind;          // 0 complex result, 1 real |.|^2 (pm_cgemm_abs2)
    float weight;
    int accumulate;
};

template <int TI, int TJ, int WM, int WN, int WK, int EPI>
static int cgemm_dma_launch(bool akf, bool bkf, int cA, int cB, int ntm, int ntn, int S, int64_t K, int64_t ksplit, float al,
                            const cx<float>* A, int64_t lda, const cx<float>* B, int64_t ldb, cx<float>* out, int64_t ldo, int64_t slab,
                            hipStream_t st, const DmaEpi& ep) {
    constexpr int RING = 3 * (32 * TI * WM + 32 * TJ * WN) * 128 * WK;
    constexpr int RED = WK > 1 ? (WK - 1) * WM * WN * TI * TJ * 48 * 64 * 4 : 0;
    constexpr int LDSB = RING > RED ? RING : RED;
#define PM_GD(AK, BK_)                                                                                                                \
    {                                                                                                                                 \
        auto kern = cgemm_dma_kernel<TI, TJ, WM, WN, WK, AK, BK_, EPI>;                                                               \
        if (LDSB > 48 * 1024)                                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);         \
        hipLaunchKernelGGL(kern, dim3(unsigned(ntm * ntn * S)), dim3(64 * WM * WN * WK), LDSB, st, cA, cB, ntm, ntn, K, ksplit, al, A, lda, B, ldb, out, \
                           ldo, slab, ep.weight, ep.accumulate);                                                                      \
    }
    if (akf && bkf) PM_GD(true, true)
    else if (akf) PM_GD(true, false)
    else if (bkf) PM_GD(false, true)
    else PM_GD(false, false)
#undef PM_GD
    return int(hipGetLastError());
}

// C: the complex result (ep.kind == 0, leading dimension ldc in complex elements) or the REAL image (ep.kind == 1, ldc in real elements)
static int cgemm_dma_run(int opA, int opB, int64_t M, int64_t N, int64_t K, double alpha, const cx<float>* A, int64_t lda,
                         const cx<float>* B, int64_t ldb, cx<float>* C, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st, DmaPlan p,
                         DmaEpi ep = DmaEpi{0, 1.f, 0}) {
    if (p.S > 1 && (!ws || ws_bytes < size_t(p.S) * size_t(M) * size_t(N) * 8)) {   // no workspace: unsplit
        p.S = 1;
        p.ksplit = K;
    }
    const bool akf = !(opA & 2), bkf = (opB & 2) != 0;
    const int cA = opA & 1, cB = opB & 1;
    const int ntm = int(M / p.tm), ntn = int(N / p.tn);
    cx<float>* out = p.S > 1 ? reinterpret_cast<cx<float>*>(ws) : C;
    const int64_t ldo = p.S > 1 ? N : ldc, slab = p.S > 1 ? M * N : 0;
    const float al = p.S > 1 ? 1.f : float(alpha);
    const DmaEpi none{0, 1.f, 0};
    int rc;
#define PM_RUN(TI, TJ, WM, WN, WK)                                                                                                            \
    rc = (ep.kind == 1 && p.S == 1)                                                                                                           \
             ? cgemm_dma_launch<TI, TJ, WM, WN, WK, 1>(akf, bkf, cA, cB, ntm, ntn, p.S, K, p.ksplit, al, A, lda, B, ldb, out, ldo, slab, st, ep) \
             : cgemm_dma_launch<TI, TJ, WM, WN, WK, 0>(akf, bkf, cA, cB, ntm, ntn, p.S, K, p.ksplit, al, A, lda, B, ldb, out, ldo, slab, st, none)
    if (p.tm == 128) PM_RUN(2, 2, 2, 2, 1);
    else if (p.wk == 2 && p.tn == 64) PM_RUN(1, 1, 2, 2, 2);
    else PM_RUN(1, 1, 2, 2, 1);
#undef PM_RUN
    if (rc || p.S == 1) return rc;
    const int64_t total = M * N;
    if (ep.kind == 1)
        hipLaunchKernelGGL(splitk_reduce_abs2_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, M, N, p.S, float(alpha),
                           reinterpret_cast<const cx<float>*>(ws), M * N, reinterpret_cast<float*>(C), ldc, ep.weight, ep.accumulate);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, M, N, p.S, float(alpha),
                           reinterpret_cast<const cx<float>*>(ws), M * N, C, ldc);
    return int(hipGetLastError());
}

template <typename T>
int cgemm_ws(int opA, int opB, int64_t M, int64_t N, int64_t K, double alpha, const cx<T>* A, int64_t lda,
             const cx<T>* B, int64_t ldb, cx<T>* C, int64_t ldc, void* ws, size_t ws_bytes, hipStream_t st) {
    if constexpr (sizeof(T) == 4) {
        DmaPlan dp;
        if (gemm_dma_plan(M, N, K, &dp) && (lda % 2) == 0 && (ldb % 2) == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0 &&
            reinterpret_cast<uintptr_t>(B) % 16 == 0)
            return cgemm_dma_run(opA, opB, M, N, K, alpha, A, lda, B, ldb, C, ldc, ws, ws_bytes, st, dp);
    }
    // K-tile depth: 16 (fp32) / 8 (fp64) = 33 KiB of LDS, 4 workgroups per CU; tuning gemm_bk = 32 / 16 doubles it
    // (66.5 KiB, 2 per CU, half the barriers per flop)
    if (sizeof(T) == 4) return cgemm_ws_bk<T, 16>(opA, opB, M, N, K, alpha, A, lda, B, ldb, C, ldc, ws, ws_bytes, st);
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

int pm_cgemm_abs2(int32_t dtype, int32_t opA, int32_t opB, int64_t M, int64_t N, int64_t K, double alpha, const void* A,
                  int64_t lda, const void* B, int64_t ldb, void* R, int64_t ldr, double weight, int32_t accumulate, void* workspace,
                  size_t workspace_bytes, void* stream) {
    if (!A || !B || !R) return fail(PM_ERR_ARG, "pm_cgemm_abs2: null buffer");
    if (M < 0 || N < 0 || K < 0 || opA < 0 || opA > 3 || opB < 0 || opB > 3) return fail(PM_ERR_ARG, "pm_cgemm_abs2: bad argument");
    if (M == 0 || N == 0) return 0;
    if (dtype != PM_C64 && dtype != PM_C128) return fail(PM_ERR_ARG, "pm_cgemm_abs2: dtype must be PM_C64 or PM_C128");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DmaPlan dp;
    const cx<float>* a = reinterpret_cast<const cx<float>*>(A);
    const cx<float>* b = reinterpret_cast<const cx<float>*>(B);
    if (dtype == PM_C64 && lda < (int64_t(1) << 22) && ldb < (int64_t(1) << 22) && gemm_dma_plan(M, N, K, &dp) && (lda % 2) == 0 &&
        (ldb % 2) == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(B) % 16 == 0)
        return cgemm_dma_run(opA, opB, M, N, K, alpha, a, lda, b, ldb, reinterpret_cast<cx<float>*>(R), ldr, workspace, workspace_bytes, st, dp,
                             DmaEpi{1, float(weight), accumulate ? 1 : 0});
    return fail(PM_ERR_UNSUPPORTED, "pm_cgemm_abs2: only the LDS-DMA shapes (complex64, multiples of 64 x 64 x 16, aligned operands); "
                                    "compose pm_cgemm and pm_abs2");
}

}  // extern "C"
