// HIP kernels wrapping the FFT engine + host launchers (explicitly instantiated
// per precision / mode in fft_row_f32.hip, fft_row_f64.hip, fft_col_f32.hip,
// fft_col_f64.hip so the four translation units compile in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "fft_io.h"

namespace pm {

constexpr int kMaxLogN = 13;  // 8192: the largest length one workgroup holds on chip

// ---- per-length configurations -------------------------------------------------------------
// Row pass: one sequence per N/16 threads, 256 threads per workgroup (512 at N = 8192); LDS is
// 8.5 B per point (complex64) so 4 workgroups / CU stay resident at N = 4096.
// VAR (row_variant() in pm_internal.h picks it per length / precision):
//   1: real and imaginary parts exchanged separately -> half the LDS per workgroup (complex128 rows of 2048 points)
//   4: two rows per thread -- stage twiddles, their products and the LDS addressing are shared by the pair (complex64 from
//      4096 points; the folded row pass pairs rows (i, i + M/2) this way)
//   5: one sequence per (small) workgroup -- twice the workgroups at 2048 points
template <typename T, int LOGN, int VAR>
struct RowCfgSel {
    static constexpr int N = 1 << LOGN, P = N >= 16 ? 16 : N, TPS = N / P;
    static constexpr int BO = (TPS >= 256 || VAR == 5) ? 1 : 256 / TPS;
    static constexpr int COMP = ((sizeof(T) == 8 && LOGN >= 12) || VAR == 1) ? 2 : 1;
    static constexpr int E = (VAR == 4) ? 2 : 1;
    using type = FftCfg<T, LOGN, 1, E, BO, COMP>;
};
// Column pass: a tile of 64 B rows (8 complex64 / 4 complex128 columns) per workgroup; at
// M = 4096 that is 256 KiB of field in the registers of 1024 threads, exchanged through LDS in
// two 136 KiB chunks (complex64: the thread's two columns; complex128: real then imaginary).
// VAR = 2: tiles of 128 B rows (8 complex128 columns, 1024 threads) for 2048-point columns -- the planes of a folded 4096-row
// complex128 transform store whole cache lines (216 -> 209 us at 4096^2, profiles/r02/exp_wide_col_tiles.log; complex64 loses
// the two-workgroups-per-CU overlap instead: 96 -> 108 us, so it keeps the 64 B tiles).
template <typename T, int LOGN, int VAR>
struct ColCfgSel {
    static constexpr int N = 1 << LOGN, P = N >= 16 ? 16 : N, TPS = N / P;
    static constexpr int E = sizeof(T) == 4 ? 2 : 1;
    static constexpr int CI = (VAR == 2 && LOGN == 11) ? 8 : (LOGN <= 12 ? 4 : 2);
    static constexpr int BO = (CI * TPS >= 256) ? 1 : 256 / (CI * TPS);
    static constexpr int COMP = (sizeof(T) == 8 && LOGN >= 11) ? 2 : 1;
    using type = FftCfg<T, LOGN, CI, E, BO, COMP>;
};

// log_g with the start-up stagger of fft_kernel packed above it: bits 8-15 the units, bits 16+ the workgroups per CU (those that start with
// the launch = 256 CUs x that); no stagger for a launch of a single round
int pm_stagger_group();             // capi.hip: the knob stagger_group
int pm_fft_stagger(int pass);       // capi.hip: the knobs fft_stagger (row kernels) / fft_stagger_col (column kernels) / fft_stagger_mid
static inline int engine_log_g(int log_g, int grid, size_t lds_bytes, int nt, int pass) {      // pass: 0 rows, 1 columns, 2 the middle pass of a fused chain, 3 / 4 the real-input row / Hermitian column kernels (fft_r2c.h)
    const int by_lds = lds_bytes ? int(size_t(160) * 1024 / lds_bytes) : 8, by_waves = 2048 / nt;
    const int per_cu = by_lds < by_waves ? (by_lds < 1 ? 1 : by_lds) : by_waves;
    int stg = pm_fft_stagger(pass);
    const bool force = stg >= 100;      // experiments: 100 + units staggers a single-round launch too (several workgroups per CU only)
    if (force) stg -= 100;
    if (grid <= 256 * per_cu && !(force && per_cu > 1)) return log_g;
    // auto (knob < 0): 8 units where a CU holds one workgroup, 1 where it holds more (they already overlap each other).  Measured
    // (profiles/r04/exp_fft_stagger.log, 2-D transform us without / with): 4096^2 complex64 93.5 / 92.8 (95.4 / 93.7 on another box),
    // complex128 210.3 / 206.0, 8192^2 complex64 493 / 474, complex128 1033 / 1013; twice the units already lose (4096^2 complex128 220)
    // (two column workgroups per CU: 93.4 against 93.3 without -- nothing; the Hermitian column kernel, one 1024-thread workgroup per CU:
    // 8192^2 real input 357 -> 344 us at 8 and 338 at 16 on one box, 392 -> 385 at 8 and 391 at 16 on another -- exp_mtf_stagger*.log;
    // the real-input row kernel: nothing at any setting)
    if (stg < 0) stg = pass == 3 ? 0 : (per_cu == 1 ? 8 : (pass == 0 ? 1 : 0));
    if (stg == 0) return log_g;
    // bit 30 (knob stagger_group, column kernels): the delay is hashed from the SIBLING GROUP of a workgroup (the 2^log_g
    // workgroups group_remap puts on one XCD with adjacent tiles) instead of from the workgroup -- siblings then stay in step, and the
    // 64 B pieces they write into the same 128 B lines (the mirrored half: one element off alignment) can meet in that XCD's L2
    return log_g | ((stg & 255) << 8) | ((per_cu & 255) << 16) | ((pass != 0 && pass != 3 && pm_stagger_group()) ? (1 << 30) : 0);
}

__device__ __forceinline__ int engine_stagger(int log_g_packed) {
    const int stg = (log_g_packed >> 8) & 255;
    unsigned lin = blockIdx.x + blockIdx.y * gridDim.x;     // dispatch order: x first
    if (stg && int(lin) < (((log_g_packed >> 16) & 255) << 8)) {
        if (log_g_packed & (1 << 30)) lin = ((lin >> 3) >> (log_g_packed & 255)) * 8 + (lin & 7);
        const unsigned h = (lin * 2654435761u) >> 29;
        for (unsigned i = 0; i < h * unsigned(stg); ++i) __builtin_amdgcn_s_sleep(8);
    }
    return log_g_packed & 255;
}

// Minimum waves per SIMD the register allocation must leave room for (the second __launch_bounds__ argument of hipcc).  The paired
// rows of a folded transform (VAR 4, E = 2) are the kernels that sit at a register boundary: complex64 fits four 256-thread
// workgroups per CU at 128 VGPRs.  complex128 (128 registers of data alone) runs two per CU at 182 - 215 registers; a build under a
// 168-register cap (three per CU) spills 24 - 186 registers and was measured in round 5: 4096^2 focus 201 -> 289 us,
// angular spectrum 305 -> 404 us, 2048^2 unchanged (profiles/r05/exp_knob_row_cap.log) -- not built any more.
template <typename C, bool COL, int VAR, typename S>
constexpr int fft_kernel_min_waves() {
    constexpr bool chirp = std::is_same<S, RowStoreChirp<typename C::T>>::value;     // (the Bluestein chain's last pass: 141 - 149 registers, left alone)
    if (!COL && VAR == 4 && C::E == 2 && !chirp) return sizeof(typename C::T) == 4 ? 4 : 1;
    // complex64 column tiles of 512 threads: two workgroups per CU need 128 registers (the allocation sits at 128 - 130)
    if (COL && sizeof(typename C::T) == 4 && C::NT == 512 && C::LDS_BYTES <= 80 * 1024) return 4;
    return 1;
}

template <typename C, bool COL, int VAR, typename L, typename S>
__global__ void __launch_bounds__(C::NT, (fft_kernel_min_waves<C, COL, VAR, S>())) fft_kernel(const L lp, const S sp, const cx<typename C::T>* __restrict__ tw, const int log_g_packed) {
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    // start-up stagger (knob fft_stagger, packed above the group shift by engine_log_g): the workgroups that start with the launch wait
    // 0 .. 7 x units x 512 cycles by a hash of their index, so that the load / transform / store phases of a CU's workgroups overlap
    const int log_g = engine_stagger(log_g_packed);
    int unit = group_remap(blockIdx.x, gridDim.x, log_g);
    if (COL) unit = unit * C::BO + pos.bo;
    cx<typename C::T> v[C::E][C::P];
    const L lpb = at_batch(lp, blockIdx.y);   // blockIdx.y: field of a batch
    const S spb = at_batch(sp, blockIdx.y);
    load<C>(lpb, unit, pos, v);
    // two sequences per thread (complex64 columns, paired rows): exchanges pipelined against the butterflies of the other
    // sequence (measured: 4096^2 column pass 57.2 -> 56.1 us, 8192^2 419 -> 358 us); row variant 5 keeps the plain order
    if constexpr (VAR != 5 && C::E == 2 && C::COMP == 1 && C::NSTAGE > 1)
        fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else
        fft_run<C>(v, pos, pm_smem, tw);
    store<C>(spb, unit, pos, v);
}


// Fused spectral-multiply column pass: forward transform, multiply by H, inverse transform -- all on the
// registers of the workgroup (the engine returns natural order, so the inverse starts where the forward ended);
// reads the tiled intermediate of the row pass and writes a tiled buffer for the inverse row pass.  This is the
// middle pass of ifft2(fft2(x) * H) in 3 passes / 6 N^2 s bytes instead of 4 passes / 8 N^2 s.
// Measured (profiles/r01/fused_as.log): 4096^2 complex128 448 vs 473 us, complex64 210 vs 239 us, 2048^2 complex64
// 55 vs 75 us against two pm_fft2 calls.  (Built without the SLP vectorizer -- with it this kernel spilled > 150
// VGPRs under the 128-register cap of the 1024-thread workgroup and lost.)
// MINW: minimum waves per SIMD the register allocation must leave room for (1: whatever the kernel wants -- 184 VGPRs for
// 2048-point complex128 tiles, ONE 512-thread workgroup per CU; 4: two of them per CU, 128 VGPRs).
template <typename C, typename S, int MINW, int KIND = -1>
__device__ __forceinline__ void col_mul_body(cx<typename C::T> (&v)[C::E][C::P], const MidMul<typename C::T>& mp, int unit, ThreadPos pos,
                                             char* smem, const cx<typename C::T>* __restrict__ tw) {
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, smem, tw);
    else fft_run<C>(v, pos, smem, tw);
    if constexpr (KIND < 0) mid_multiply_conj<C>(mp, unit, pos, v);
    else mid_multiply_conj_kind<C, KIND>(mp, unit, pos, v);
    __syncthreads();   // LDS of the forward exchange is reused by the inverse
    // opaque copy of the slot: otherwise the twiddles (and their products) of the first transform are CSE'd
    // with the second and kept live across it -- hundreds of spilled registers at 1024 threads
    ThreadPos pos2 = pos;
    asm volatile("" : "+v"(pos2.t), "+v"(pos2.cl), "+v"(pos2.bo));
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos2, smem, tw);
    else fft_run<C>(v, pos2, smem, tw);
#pragma unroll
    for (int e = 0; e < C::E; ++e)
#pragma unroll
        for (int m = 0; m < C::P; ++m) v[e][m].y = -v[e][m].y;
}

template <typename C, typename S = ColStoreTiled<typename C::T>, int MINW = 1>
__global__ void __launch_bounds__(C::NT, (C::NT == 512 ? MINW : 1))   // 2nd argument: min waves per SIMD
    fft_col_mul_kernel(const ColLoadTiled<typename C::T> lp0, const MidMul<typename C::T> mp0,
                                                            const S sp0,
                                                            const cx<typename C::T>* __restrict__ tw, const int log_g) {
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = group_remap(blockIdx.x, gridDim.x, log_g) * C::BO + pos.bo;
    const auto lp = at_batch(lp0, blockIdx.y);
    const auto mp = at_batch(mp0, blockIdx.y);
    const auto sp = at_batch(sp0, blockIdx.y);
    cx<typename C::T> v[C::E][C::P];
    load<C>(lp, unit, pos, v);
    col_mul_body<C, S, MINW>(v, mp, unit, pos, pm_smem, tw);
    store<C>(sp, unit, pos, v);
}

// Lean addressing for the tiled column passes of the fused chain (whole, unrotated tiles -- host-checked): ONE uniform 64-bit base
// per register slot (scalar registers) + ONE 32-bit per-thread byte offset.  With the generic loaders the compiler keeps the 16 +
// 16 + 16 per-slot 64-bit vector addresses of load / multiplier / store live (96 VGPRs): the difference between 184 registers (one
// 512-thread workgroup per CU) and 126 (two).
template <typename C>
struct PfAddr {
    using T = typename C::T;
    static constexpr int TC = C::CI * C::E;
    static constexpr int ES = int(sizeof(cx<T>));
    uint32_t vdata;   // byte offset of this thread's first element inside a tile block (load and store layouts are the same)
    int row0;         // the thread's first row (slot 0)
    int TL;
    int64_t mstep;    // bytes between register slots m and m + 1
    __device__ __forceinline__ PfAddr(ThreadPos pos, int log_k) {
        TL = TC << log_k;
        row0 = pos.t;
        vdata = uint32_t(pos.t * TL + pos.cl * C::E) * ES;
        mstep = int64_t(C::TPS) * TL * ES;
    }
    __device__ __forceinline__ int64_t tile_off(int tile, int nrows, int log_k) const {   // bytes, uniform
        const int tl = tile >> log_k, sub = tile & ((1 << log_k) - 1);
        return (int64_t(tl) * nrows * TL + sub * TC) * ES;
    }
};

// rows [ay.off, ay.off + ay.len) of the length-N columns are stored (memory row = logical row - off), the rest is zero padding;
// ay.shift must be 0 (host-checked)
template <typename C>
__device__ __forceinline__ void pf_load(const ColLoadTiled<typename C::T>& p, int tile, const PfAddr<C>& A, cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    const char* tb = reinterpret_cast<const char*>(p.src) + A.tile_off(tile, p.nrows, p.log_k) - int64_t(p.ay.off) * A.TL * PfAddr<C>::ES;
    const int lo = p.ay.off, hi = p.ay.off + p.ay.len;
    const bool full = p.ay.len == C::N;      // uniform
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const char* a = tb + m * A.mstep + A.vdata;
        const int pp = A.row0 + m * C::TPS;
        if (full || (pp >= lo && pp < hi)) {
            if constexpr (C::E == 2) {
                const Vec4<T> w = *reinterpret_cast<const Vec4<T>*>(a);
                v[0][m] = {w.a, w.b};
                v[1][m] = {w.c, w.d};
            } else {
                v[0][m] = *reinterpret_cast<const cx<T>*>(a);
            }
        } else {
#pragma unroll
            for (int e = 0; e < C::E; ++e) v[e][m] = {T(0), T(0)};
        }
    }
}

template <typename C>
__device__ __forceinline__ void pf_store(const ColStoreTiled<typename C::T>& p, int tile, const PfAddr<C>& A, const cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    char* tb = reinterpret_cast<char*>(p.dst) + A.tile_off(tile, p.nrows, p.log_k);
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        char* a = tb + m * A.mstep + A.vdata;
        if constexpr (C::E == 2) *reinterpret_cast<Vec4<T>*>(a) = Vec4<T>{v[0][m].x, v[0][m].y, v[1][m].x, v[1][m].y};
        else *reinterpret_cast<cx<T>*>(a) = v[0][m];
    }
}
// ... only rows [0, nrows) of the results (the cropped convolution of the Bluestein chain)
template <typename C>
__device__ __forceinline__ void pf_store(const ColStoreTiledCrop<typename C::T>& p, int tile, const PfAddr<C>& A, const cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    char* tb = reinterpret_cast<char*>(p.dst) + A.tile_off(tile, p.nrows, p.log_k);
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        if (A.row0 + m * C::TPS >= p.nrows) continue;
        char* a = tb + m * A.mstep + A.vdata;
        if constexpr (C::E == 2) *reinterpret_cast<Vec4<T>*>(a) = Vec4<T>{v[0][m].x, v[0][m].y, v[1][m].x, v[1][m].y};
        else *reinterpret_cast<cx<T>*>(a) = v[0][m];
    }
}


// Mode 3: one tile per workgroup like mode 0, but built to fit TWO 512-thread workgroups per CU (128 VGPRs) without spilling: the
// lean addressing of the persistent kernel (one uniform base + one 32-bit offset per stream instead of 16 vector addresses each),
// the separable multiplier fetched four slots at a time.  Sixteen waves per CU issue memory operations instead of eight, and the
// load / transform / store phases of the two workgroups overlap.
template <typename C, int KIND, typename S = ColStoreTiled<typename C::T>>
__global__ void __launch_bounds__(C::NT, 4)
    fft_col_mul_lean_kernel(const ColLoadTiled<typename C::T> lp0, const MidMul<typename C::T> mp0, const S sp0,
                            const cx<typename C::T>* __restrict__ tw, const int log_g_packed) {
    using T = typename C::T;
    static_assert(C::BO == 1, "one tile per workgroup");
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const int log_g = engine_stagger(log_g_packed);
    constexpr int TC = C::CI * C::E, ES = int(sizeof(cx<T>));
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const auto lp = at_batch(lp0, blockIdx.y);
    const auto mp = at_batch(mp0, blockIdx.y);
    const auto sp = at_batch(sp0, blockIdx.y);
    const PfAddr<C> A(pos, lp.log_k);
    const int unit = group_remap(blockIdx.x, gridDim.x, log_g);
    cx<T> v[C::E][C::P];
    pf_load<C>(lp, unit, A, v);
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else fft_run<C>(v, pos, pm_smem, tw);
    if constexpr (KIND == MUL_SEPARABLE) {
        cx<T> hx[C::E];
#pragma unroll
        for (int e = 0; e < C::E; ++e) hx[e] = mp.mul_x[unit * TC + pos.cl * C::E + e];
        const int ys = mp.ystep > 1 ? mp.ystep : 1;
        const uint32_t voff = uint32_t(pos.t * ys) * ES;
        const int64_t mstep = int64_t(C::TPS) * ys * ES;
        const char* hb = reinterpret_cast<const char*>(mp.mul);
#pragma unroll
        for (int m0 = 0; m0 < C::P; m0 += 4) {
            cx<T> hy[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) hy[j] = *reinterpret_cast<const cx<T>*>(hb + (m0 + j) * mstep + voff);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < C::E; ++e) {
                    const cx<T> h = cmul(hy[j], hx[e]);
                    const cx<T> x = mp.conj ? cmulc(v[e][m0 + j], h) : cmul(v[e][m0 + j], h);
                    v[e][m0 + j] = {x.x, -x.y};
                }
            __builtin_amdgcn_sched_barrier(0);      // keep the four-slot batches apart: all sixteen hy at once are 64 VGPRs
        }
    } else {
        // full multiplier H[k][c] (tf=, convolution kernels): this thread's columns of rows k = t + m TPS, four rows at a time
        const uint32_t voff = uint32_t(int64_t(pos.t) * mp.ld + pos.cl * C::E) * ES;
        const int64_t mstep = int64_t(C::TPS) * mp.ld * ES;
        const char* hb = reinterpret_cast<const char*>(mp.mul + unit * TC);
#pragma unroll
        for (int m0 = 0; m0 < C::P; m0 += 4) {
            cx<T> hf[4][C::E];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* a = hb + (m0 + j) * mstep + voff;
                if constexpr (C::E == 2) {
                    if (mp.vec_ok) {
                        const Vec4<T> w = *reinterpret_cast<const Vec4<T>*>(a);
                        hf[j][0] = {w.a, w.b};
                        hf[j][1] = {w.c, w.d};
                    } else {
                        hf[j][0] = reinterpret_cast<const cx<T>*>(a)[0];
                        hf[j][1] = reinterpret_cast<const cx<T>*>(a)[1];
                    }
                } else {
                    hf[j][0] = *reinterpret_cast<const cx<T>*>(a);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < C::E; ++e) {
                    const cx<T> x = mp.conj ? cmulc(v[e][m0 + j], hf[j][e]) : cmul(v[e][m0 + j], hf[j][e]);
                    v[e][m0 + j] = {x.x, -x.y};
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();   // LDS of the forward exchange is reused by the inverse
    ThreadPos pos2 = pos;   // opaque copy: see col_mul_body
    asm volatile("" : "+v"(pos2.t), "+v"(pos2.cl), "+v"(pos2.bo));
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos2, pm_smem, tw);
    else fft_run<C>(v, pos2, pm_smem, tw);
#pragma unroll
    for (int e = 0; e < C::E; ++e)
#pragma unroll
        for (int m = 0; m < C::P; ++m) v[e][m].y = -v[e][m].y;
    pf_store<C>(sp, unit, A, v);
}

int pm_num_cus();   // capi.hip: compute units of the current device (cached)

// mode (tuning colmul_mode; 512-thread tiles = 2048-point columns only): 0 one tile per workgroup, 1 the same under a 128-VGPR cap
// (two workgroups per CU), 2 persistent workgroups that prefetch the next tile (fft_col_mul_pf_kernel)
template <typename T, int LOGN, typename S>
int launch_col_mul_one(const ColLoadTiled<T>& lp, const MidMul<T>& mp, const S& sp, const cx<T>* tw, int ntiles,
                       int log_g, hipStream_t st, int nbatch, int mode) {
    using C = typename ColCfgSel<T, LOGN, 0>::type;
    const int grid = (ntiles + C::BO - 1) / C::BO;
    if (grid <= 0) return 0;
    if constexpr (C::BO == 1 && C::NT >= 512) {
        if (mode == 3) {
            // an unrotated window of stored rows (zero padding around it is synthesised), every tile present
            const bool whole = lp.ay.shift == 0 && lp.ay.n == C::N && lp.ntiles == grid * C::BO && sp.ntiles == lp.ntiles;
            // (a full multiplier's per-thread byte offset must fit 32 bits: rows t < TPS of an ld-element array)
            const bool off32 = mp.kind != MUL_FULL || int64_t(C::TPS) * mp.ld * int64_t(sizeof(cx<T>)) < (int64_t(1) << 31);
            if (whole && off32 && (mp.kind == MUL_SEPARABLE || mp.kind == MUL_FULL) && mp.ncols >= grid * C::CI * C::E) {
                auto kern = mp.kind == MUL_FULL ? fft_col_mul_lean_kernel<C, MUL_FULL, S> : fft_col_mul_lean_kernel<C, MUL_SEPARABLE, S>;
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   int(C::LDS_BYTES));
                if (e != hipSuccess) return int(e);
                hipLaunchKernelGGL(kern, dim3(grid, nbatch), dim3(C::NT), C::LDS_BYTES, st, lp, mp, sp, tw, engine_log_g(log_g, grid, C::LDS_BYTES, C::NT, 2));
                return int(hipGetLastError());
            }
        }
    }
    auto kern = fft_col_mul_kernel<C, S>;
    if (C::LDS_BYTES > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           int(C::LDS_BYTES));
        if (e != hipSuccess) return int(e);
    }
    hipLaunchKernelGGL(kern, dim3(grid, nbatch), dim3(C::NT), C::LDS_BYTES, st, lp, mp, sp, tw, log_g);
    return int(hipGetLastError());
}

template <typename T, typename S>
int launch_col_mul_impl(int logm, const ColLoadTiled<T>& lp, const MidMul<T>& mp, const S& sp, const cx<T>* tw,
                        int ntiles, int log_g, hipStream_t st, int nbatch, int mode = 0) {
    switch (logm) {
#define PM_CASE(k) \
    case k:        \
        return launch_col_mul_one<T, k, S>(lp, mp, sp, tw, ntiles, log_g, st, nbatch, mode);
        PM_CASE(1) PM_CASE(2) PM_CASE(3) PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7)
        PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASE(11) PM_CASE(12) PM_CASE(13)
#undef PM_CASE
        default:
            return -2;
    }
}

template <typename T, bool COL, int LOGN, int VAR, typename L, typename S>
int launch_one(const L& lp, const S& sp, const cx<T>* tw, int units, int log_g, hipStream_t st, int nbatch) {
    using Sel = typename std::conditional<COL, ColCfgSel<T, LOGN, VAR>, RowCfgSel<T, LOGN, VAR>>::type;
    using C = typename Sel::type;
    auto kern = fft_kernel<C, COL, VAR, L, S>;
    constexpr size_t LDSB = C::LDS_BYTES;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int per_wg = C::BO * (COL ? 1 : C::E);     // row mode: a thread owns E consecutive rows
    const int grid = (units + per_wg - 1) / per_wg;
    if (grid <= 0 || nbatch <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid, nbatch), dim3(C::NT), LDSB, st, lp, sp, tw, engine_log_g(log_g, grid, LDSB, C::NT, COL ? 1 : 0));
    return int(hipGetLastError());
}


// folded row pass (RowStoreFold): units are row PAIRS (i, i + M/2); complex64 / complex128 rows of 2048 .. 8192 points
template <typename T, int LOGN>
int launch_fold_one(const RowLoadNat<T>& lp, const RowStoreFold<T>& sp, const cx<T>* tw, int npairs, int log_g, hipStream_t st,
                    int nbatch) {
    using C = typename RowCfgSel<T, LOGN, 4>::type;
    auto kern = fft_kernel<C, false, 4, RowLoadNat<T>, RowStoreFold<T>>;
    constexpr size_t LDSB = C::LDS_BYTES;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int grid = (npairs + C::BO - 1) / C::BO;
    if (grid <= 0 || nbatch <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid, nbatch), dim3(C::NT), LDSB, st, lp, sp, tw, engine_log_g(log_g, grid, LDSB, C::NT, 0));
    return int(hipGetLastError());
}
template <typename T>
int launch_fold_impl(int logn, const RowLoadNat<T>& lp, const RowStoreFold<T>& sp, const cx<T>* tw, int npairs, int log_g,
                     hipStream_t st, int nbatch) {
    switch (logn) {
        case 11: return launch_fold_one<T, 11>(lp, sp, tw, npairs, log_g, st, nbatch);
        case 12: return launch_fold_one<T, 12>(lp, sp, tw, npairs, log_g, st, nbatch);
        case 13: return launch_fold_one<T, 13>(lp, sp, tw, npairs, log_g, st, nbatch);
        default: return -2;
    }
}

// last pass of the folded fused operation: rows rebuilt from the two planes (RowLoadFold), natural output
template <typename T, int LOGN>
int launch_unfold_one(const RowLoadFold<T>& lp, const RowStoreNat<T>& sp, const cx<T>* tw, int npairs, hipStream_t st, int nbatch) {
    using C = typename RowCfgSel<T, LOGN, 4>::type;
    auto kern = fft_kernel<C, false, 4, RowLoadFold<T>, RowStoreNat<T>>;
    constexpr size_t LDSB = C::LDS_BYTES;
    if (LDSB > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(LDSB));
        if (e != hipSuccess) return int(e);
    }
    const int grid = (npairs + C::BO - 1) / C::BO;
    if (grid <= 0 || nbatch <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(grid, nbatch), dim3(C::NT), LDSB, st, lp, sp, tw, 0);
    return int(hipGetLastError());
}
template <typename T>
int launch_unfold_impl(int logn, const RowLoadFold<T>& lp, const RowStoreNat<T>& sp, const cx<T>* tw, int npairs, hipStream_t st,
                       int nbatch) {
    switch (logn) {
        case 11: return launch_unfold_one<T, 11>(lp, sp, tw, npairs, st, nbatch);
        case 12: return launch_unfold_one<T, 12>(lp, sp, tw, npairs, st, nbatch);
        case 13: return launch_unfold_one<T, 13>(lp, sp, tw, npairs, st, nbatch);
        default: return -2;
    }
}

// The row-pass tiling per length and precision (what pm_internal.h row_variant() returns; measured in rounds 1 - 2): only THAT one is
// built.  Until round 5 every length from 2048 points carried all four tilings of both precisions for the knob row_var -- 12 dead
// kernel families per loader / storer pair, 7 MB of the library.
template <typename T, int LOGN>
constexpr int built_row_var() {
    if (sizeof(T) == 4) return LOGN >= 12 ? 4 : (LOGN == 11 ? 5 : 0);
    return LOGN == 11 ? 1 : 0;
}

template <typename T, bool COL, typename L, typename S>
int launch_fft(int logn, int var, const L& lp, const S& sp, const cx<T>* tw, int units, int log_g, hipStream_t st, int nbatch) {
    switch (logn) {
#define PM_CASE(k) \
    case k:        \
        return launch_one<T, COL, k, 0, L, S>(lp, sp, tw, units, log_g, st, nbatch);
// lengths with more than one tiling (see RowCfgSel / ColCfgSel): rows take the built one; columns the 128 B tiles (var 2) for
// 2048-point complex128 tiles, else the 64 B tiles
#define PM_CASEV(k)                                                                                              \
    case k:                                                                                                      \
        if constexpr (!COL) return launch_one<T, COL, k, (COL ? 0 : built_row_var<T, k>()), L, S>(lp, sp, tw, units, log_g, st, nbatch); \
        if constexpr (COL && k == 11 && sizeof(T) == 8) {                                                        \
            if (var == 2) return launch_one<T, COL, k, (COL ? 2 : 0), L, S>(lp, sp, tw, units, log_g, st, nbatch); \
        }                                                                                                        \
        return launch_one<T, COL, k, 0, L, S>(lp, sp, tw, units, log_g, st, nbatch);
        PM_CASE(1) PM_CASE(2) PM_CASE(3) PM_CASE(4) PM_CASE(5) PM_CASE(6) PM_CASE(7)
        PM_CASE(8) PM_CASE(9) PM_CASE(10) PM_CASEV(11) PM_CASEV(12) PM_CASEV(13)
#undef PM_CASE
#undef PM_CASEV
        default:
            return -2;
    }
}

// entry points, one explicit instantiation per .hip file
template <typename T> int launch_row_tiled(int logn, int var, const RowLoadNat<T>&, const RowStoreTiled<T>&, const cx<T>* tw, int nseq, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_row_nat(int logn, int var, const RowLoadNat<T>&, const RowStoreNat<T>&, const cx<T>* tw, int nseq, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_col_tiled(int logm, int var, const ColLoadTiled<T>&, const ColStoreNat<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_col_nat(int logm, int var, const ColLoadNat<T>&, const ColStoreNat<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_col_mul(int logm, const ColLoadTiled<T>&, const MidMul<T>&, const ColStoreTiled<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t, int nbatch = 1, int mode = 0);
template <typename T> int launch_col_mul_crop(int logm, const ColLoadTiled<T>&, const MidMul<T>&, const ColStoreTiledCrop<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t);
template <typename T> int launch_row_from_tiled(int logn, int var, const RowLoadTiled<T>&, const RowStoreNat<T>&, const cx<T>* tw, int nseq, hipStream_t, int nbatch = 1);
template <typename T> int launch_row_fold(int logn, const RowLoadNat<T>&, const RowStoreFold<T>&, const cx<T>* tw, int npairs, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_row_unfold(int logn, const RowLoadFold<T>&, const RowStoreNat<T>&, const cx<T>* tw, int npairs, hipStream_t, int nbatch = 1);
template <typename T> int launch_row_chirp_tiled(int logn, int var, const RowLoadChirp<T>&, const RowStoreTiled<T>&, const cx<T>* tw, int nseq, int log_g, hipStream_t);
template <typename T> int launch_row_tiled_chirp(int logn, int var, const RowLoadTiled<T>&, const RowStoreChirp<T>&, const cx<T>* tw, int nseq, hipStream_t);

}  // namespace pm
