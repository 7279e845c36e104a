// Global-memory loaders / storers for the FFT engine.
//
// "Row" mode: the transform axis is contiguous in memory, one sequence per row
// (CI = E = 1).  "Col" mode: the transform axis is strided; a workgroup owns a
// tile of TC = CI*E adjacent columns so that every row of the tile is one
// TC*sizeof(complex) = 64 B segment (complex64: 8 columns, complex128: 4).
//
// The private intermediate of the 2-D transform is stored TILED:
//     W[tile = c / TC][row q][c % TC]
// so the column pass streams one contiguous block per workgroup, and the row
// pass writes 64 B segments whose 128 B line partner (row q^1) is written by
// the same or an adjacent workgroup on the same XCD (merged in that XCD's L2).
//
// Register discipline: a thread's P points sit at logical indices t + m*TPS.
// The rotation of fftshift / ifftshift is by N/2 = (P/2)*TPS, i.e. a rotation
// of the REGISTER SLOT index m by P/2 -- resolved at compile time (ROT
// template argument), so addresses stay affine (one base + constant offsets)
// and no per-element modulo arithmetic is executed.  Arbitrary rotations take
// the generic (ROT = -1) path.
//
// All functions are __host__ __device__: tools/emu_fft.cpp runs them on the CPU.
#pragma once
#include <type_traits>

#include "fft_engine.h"

namespace pm {

enum : int { EPI_NONE = 0, EPI_ABS2 = 1, EPI_ABS2_ACCUM = 2 };
enum : int { MUL_NONE = 0, MUL_FULL = 1, MUL_SEPARABLE = 2 };

template <typename T>
struct RowLoadNat {
    const cx<T>* src;
    int64_t ld;     // elements between consecutive sequences
    AxisMap ax;     // along the transform axis
    int nseq;       // number of sequences (memory rows)
    int conj;
    int nt;         // non-temporal loads (input is read exactly once)
    int64_t bstride;   // elements between consecutive fields of a batch (blockIdx.y); 0 / absent: one field
    int real;       // the array is REAL (T, not cx<T>): src points at T, ld / bstride count real elements, imag = 0
    int eoff;       // E > 1: sequence of slot e is unit*E + e (eoff == 0, consecutive rows) or unit + e*eoff (rows eoff apart:
                    // the pair (i, i + M/2) of a folded column transform)
    // real == 3: the same synthesis from PACKED (amplitude, OPD) pairs -- src is read as complex (re = amplitude, im = OPD): one
    // 8-byte load per element instead of two 4-byte loads from two arrays (a wavelength loop packs the two maps once)
    // real == 2: PUPIL SYNTHESIS on the fly -- src is the real OPD map and the element is amp * exp(2 pi i * k2 * opd)
    // (Wavefront.from_amp_and_phase, prysm/propagation/wavefront.py:58-79, fused into the load: the complex pupil is
    // never written to memory).  amp_kind: 0 unit amplitude, 1 float, 2 double, 3 bool / uint8.
    const void* amp;
    int amp_kind;
    int64_t amp_ld;
    double k2;      // phase in turns per OPD unit: k / (2 pi)
};

template <typename T>
struct RowStoreTiled {
    cx<T>* dst;
    int nseq;       // rows of the intermediate
    int log_tc;     // log2(LAYOUT tile width TL): one row of a layout tile is TL*sizeof(complex) contiguous bytes
    int64_t bstride;   // elements between consecutive fields of a batch (blockIdx.y); 0 / absent: one field
};

// Row pass of a FOLDED 2-D transform: the thread's two rows are the pair (i, i + M/2) of the column axis; one radix-2
// decimation-in-frequency step of the length-M column transform is taken here,
//     plane 0 row i = y[i] + y[i + M/2]          (-> even output rows, an M/2-point column transform)
//     plane 1 row i = (y[i] - y[i + M/2]) W_M^i  (-> odd output rows)
// so the column pass runs M/2-point tiles: 64 B pieces instead of 32 B at M = 8192, twice the register budget per
// thread at M = 4096 complex128.  Both planes are stored tiled like RowStoreTiled.
template <typename T>
struct RowStoreFold {
    cx<T>* dst;             // plane 0; plane 1 at dst + plane_stride
    int64_t plane_stride;   // elements
    int npairs;             // M / 2 = rows per plane
    int log_tc;
    const cx<T>* twm;       // twiddle table of length M (W_M^k)
    int swap;               // input rows were rotated by M/2 (ifftshift): slot 0 holds logical row i + M/2
    int64_t bstride;
};

template <typename T>
struct RowStoreNat {
    cx<T>* dst;
    int64_t ld;
    AxisMap ax;
    int nseq;
    int conj;
    T scale;
    int use_ay;     // 0: sequence s is written to memory row s; 1: to row ay.map(s) (rotation / crop of rows)
    AxisMap ay;
    int64_t bstride;   // elements between consecutive fields of a batch (blockIdx.y); 0 / absent: one field
    int eoff;          // E > 1: sequence of slot e is unit*E + e (0) or unit + e*eoff (the pair (n, n + M/2) of an unfolded transform)
    int nt;            // non-temporal stores (the output is written exactly once and is about the size of the Infinity Cache)
};

// row pass reading the tiled intermediate (third pass of the fused fft2 -> multiply -> ifft2)
template <typename T>
struct RowLoadTiled {
    const cx<T>* src;
    int nrows;      // rows stored in the tiled buffer
    int log_tl;     // log2(layout tile width)
    int row0;       // first stored row to transform (sequence s reads stored row row0 + s)
    int nseq;
    int conj;
    int64_t bstride;   // elements between consecutive fields of a batch (blockIdx.y); 0 / absent: one field
};

// Row pass UNDOING a fold (last pass of the folded fused operation): the two planes hold the M/2-point inverse column
// transforms A0[n], A1[n] of the even / odd bins; one radix-2 decimation-in-time step rebuilds the two rows
//     z[n] = A0[n] + conj(W_M^n) A1[n],     z[n + M/2] = A0[n] - conj(W_M^n) A1[n]
// in the registers of the thread that then transforms both along the row.
template <typename T>
struct RowLoadFold {
    const cx<T>* src;       // plane 0; plane 1 at src + plane_stride
    int64_t plane_stride;
    int npairs;             // M / 2 = rows per plane
    int log_tl;
    const cx<T>* twm;       // W_M^k
    int conj;               // conjugate the rebuilt rows (inverse row transform = conj-in / conj-out)
    int64_t bstride;
};

// column pass writing back into the tiled layout (second pass of the fused operation)
template <typename T>
struct ColStoreTiled {
    cx<T>* dst;
    int nrows;      // rows of the destination (the full transform length)
    int ntiles;
    int log_k;
    int64_t bstride;   // elements between consecutive fields of a batch (blockIdx.y); 0 / absent: one field
};

// spectral multiplier applied between the forward and inverse column transforms, indexed by the
// unshifted bin (row k, column c)
// ... storing only rows [0, nrows) of the column results (the cropped convolution of the Bluestein chain): the buffer holds
// nrows rows per layout tile, rows beyond are never written (nor read by the row pass that follows)
template <typename T>
struct ColStoreTiledCrop {
    cx<T>* dst;
    int nrows;
    int ntiles;
    int log_k;
};

template <typename T>
struct MidMul {
    int kind;       // MUL_FULL / MUL_SEPARABLE
    int conj;
    const cx<T>* mul;     // FULL: mul[k*ld + c]; SEPARABLE: hy[k]
    const cx<T>* mul_x;   // SEPARABLE: hx[c]
    int64_t ld;
    int ncols;
    int64_t bstride;   // elements between the multipliers (FULL) / the hy vectors (SEPARABLE) of consecutive fields;
    int64_t bstride_x; // ... between the hx vectors.  0: one multiplier for the whole batch
    int ystep;         // SEPARABLE: row factor of bin k is mul[k * ystep] (0 means 1); 2 for the planes of a folded transform
    int vec_ok;        // FULL, complex64: the multipliers of a thread's two columns may be read as one 16-byte access
};

template <typename T>
struct ColLoadTiled {
    const cx<T>* src;
    int nrows;      // rows stored in the intermediate (memory rows)
    AxisMap ay;     // logical row -> stored row
    int ntiles;     // number of TC-wide tiles (workgroup units)
    int log_k;      // layout tile = 2^log_k workgroup tiles wide (TL = TC << log_k)
    int64_t bstride;   // elements between consecutive fields of a batch (blockIdx.y); 0 / absent: one field
};

template <typename T>
struct ColLoadNat {
    const cx<T>* src;
    int64_t ld;
    AxisMap ay;     // along the transform axis (rows)
    int ncols;      // number of columns in memory
    int conj;
    int vec_ok;     // base and ld allow 16-byte loads of column pairs
    int64_t bstride;   // elements between consecutive fields of a batch (blockIdx.y); 0 / absent: one field
};

template <typename T>
struct ColStoreNat {
    void* dst;      // cx<T>* or T* (abs2 epilogues)
    int64_t ld;
    AxisMap ay;     // output rows (shift / crop)
    AxisMap ax;     // output columns (shift / crop)
    int conj;
    int epilogue;
    T scale;        // multiplies the complex value
    T weight;       // EPI_ABS2_ACCUM: dst += weight * |scale * v|^2
    int mul_kind;   // multiplier indexed by the LOGICAL (unshifted) bin (row k, col c)
    int mul_conj;
    const cx<T>* mul;     // MUL_FULL: mul[k*mul_ld + c]; MUL_SEPARABLE: row factor hy[k]
    const cx<T>* mul_x;   // MUL_SEPARABLE: column factor hx[c]
    int64_t mul_ld;
    int vec_ok;
    int nt;         // non-temporal stores on the fast path (output is written exactly once)
    int64_t bstride;       // OUTPUT elements (complex, or real for the |.|^2 epilogues) between fields of a batch
    int64_t mul_bstride;   // elements between per-field multipliers (FULL) / hy vectors (SEPARABLE); 0: shared
    int64_t mul_bstride_x; // elements between per-field hx vectors
};

// A caller's 2-D array seen through its pm_fft2 view (window / rotation per axis, real arrays, conjugation): the input of the
// both-axes Bluestein form and of the big power-of-two path.
template <typename T>
struct Blue2dIn {
    const void* src;   // cx<T>* or T* (real)
    int64_t ld;
    AxisMap ay, ax;
    int conj, real;
};
// element (i, j) of the LOGICAL array behind the view: zero outside the stored window
template <typename T>
PM_HD cx<T> fetch2d(const Blue2dIn<T>& in, int i, int j) {
    const int qy = in.ay.map(i), qx = in.ax.map(j);
    cx<T> x{T(0), T(0)};
    if (qy >= 0 && qx >= 0) {
        const int64_t at = int64_t(qy) * in.ld + qx;
        if (in.real)
            x.x = reinterpret_cast<const T*>(in.src)[at];
        else
            x = reinterpret_cast<const cx<T>*>(in.src)[at];
        if (in.conj) x.y = -x.y;
    }
    return x;
}

// Row pass of the both-axes Bluestein chain with the FIRST chirp multiply in its load: sequence i (i < n1) is the logical row
// x(i, j) w1[i] w2[j], j < n2, zero padded to the convolution length of the transform (bluestein.h).
template <typename T>
struct RowLoadChirp {
    Blue2dIn<T> in;
    const cx<T>* w1;
    const cx<T>* w2;
    int nseq;       // n1
};
// ... and with the LAST chirp multiply and the caller's epilogue in its store: bin (k, c), k < n1, c < n2, goes out as
// epilogue(conj?(v) w1[k] w2[c]) through store_one (scale, conj, rotation / crop, multiplier, |.|^2)
template <typename T>
struct RowStoreChirp {
    ColStoreNat<T> out;
    const cx<T>* w1;
    const cx<T>* w2;
    int n1, n2;
    int conj;       // conj-out of the inverse row transform
};

// slot rotation helper: memory index (before the window offset) of register slot m.
//   ROT >= 0 : shift == ROT * TPS  ->  p = t + ((m + ROT) mod P) * TPS      (compile-time slot)
//   ROT <  0 : generic             ->  p = (t + m*TPS + shift) mod N
template <typename C, int ROT>
PM_HD int slot_pos(int t, int m, int shift) {
    if constexpr (ROT >= 0) {
        return t + ((m + ROT) & (C::P - 1)) * C::TPS;
    } else {
        int p = t + m * C::TPS + shift;
        if (p >= C::N) p -= C::N;
        return p;
    }
}

// classify a shift for length-N transforms of config C: returns ROT (>= 0) or -1
template <typename C>
PM_HD int rot_of(int shift) {
    if (shift == 0) return 0;
    if (C::P >= 2 && shift == C::N / 2) return C::P / 2;
    return -1;
}

template <typename T>
struct alignas(16) Vec4 {
    T a, b, c, d;
};

// non-temporal (streaming) accesses: data touched exactly once should not displace the intermediate of
// the 2-D transform from L2 / Infinity Cache.  The builtins need scalar or ext-vector types.
#if defined(__HIPCC__)
template <typename T, int NW> struct NtVec { typedef T type __attribute__((ext_vector_type(NW))); };
template <typename T>
__device__ __forceinline__ cx<T> nt_load_cx(const cx<T>* p) {
    typedef typename NtVec<T, 2>::type V;
    const V w = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
    return {w[0], w[1]};
}
template <typename T>
__device__ __forceinline__ void nt_store_cx(cx<T>* p, cx<T> v) {
    typedef typename NtVec<T, 2>::type V;
    V w;
    w[0] = v.x;
    w[1] = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<V*>(p));
}
template <typename T>
__device__ __forceinline__ void nt_store_v4(Vec4<T>* p, Vec4<T> v) {
    typedef typename NtVec<T, 4>::type V;
    V w;
    w[0] = v.a; w[1] = v.b; w[2] = v.c; w[3] = v.d;
    __builtin_nontemporal_store(w, reinterpret_cast<V*>(p));
}
template <typename T>
__device__ __forceinline__ void nt_store_s(T* p, T v) { __builtin_nontemporal_store(v, p); }
template <typename T>
__device__ __forceinline__ T nt_load_s(const T* p) { return __builtin_nontemporal_load(p); }
#else
template <typename T> inline T nt_load_s(const T* p) { return *p; }
template <typename T> inline cx<T> nt_load_cx(const cx<T>* p) { return *p; }
template <typename T> inline void nt_store_cx(cx<T>* p, cx<T> v) { *p = v; }
template <typename T> inline void nt_store_v4(Vec4<T>* p, Vec4<T> v) { *p = v; }
template <typename T> inline void nt_store_s(T* p, T v) { *p = v; }
#endif

// amp * exp(2 pi i * turns): the phase is reduced to [-0.5, 0.5] turns in fp64 (OPD * k reaches thousands of radians),
// then the hardware sine / cosine (argument in revolutions; measured max abs error 1.3e-7 on that interval,
// experiments/scripts/exp_hwsin.cpp) for float, sincospi for double.
template <typename T>
PM_HD cx<T> synth_value(T opd, T a, double k2) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (sizeof(T) == 4) {
        // fp32 OPD map: the product opd * k2 carried as TWO floats (the rounding error of the first product is exact in an FMA; k2's
        // low part adds the bits a float k2 lacks), reduced to [-0.5, 0.5] turns by an exact subtraction -- 3e-8 turns of error, below
        // the hardware sine's 1.3e-7, in six fp32 instructions where the fp64 form took five at half rate plus two conversions
        // (round 5: the synthesising row pass of the wavelength loop is VALU co-limited)
        const float khi = float(k2), klo = float(k2 - double(khi));
        const float th = opd * khi;
        const float tl = __builtin_fmaf(opd, klo, __builtin_fmaf(opd, khi, -th));
        const float rf = (th - __builtin_rintf(th)) + tl;
        return {a * __builtin_amdgcn_cosf(rf), a * __builtin_amdgcn_sinf(rf)};
    }
#endif
    const double turns = double(opd) * k2;
    const double r = turns - rint(turns);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (sizeof(T) == 4) {
        const float rf = float(r);
        return {a * __builtin_amdgcn_cosf(rf), a * __builtin_amdgcn_sinf(rf)};
    } else {
        double sn, cs;
        sincospi(2.0 * r, &sn, &cs);
        return {T(a * cs), T(a * sn)};
    }
#else
    const double ang = 6.283185307179586476925286766559 * r;
    return {T(double(a) * cos(ang)), T(double(a) * sin(ang))};
#endif
}
template <typename T>
PM_HD T synth_amp(const void* amp, int kind, int64_t idx) {
    if (kind == 1) return T(reinterpret_cast<const float*>(amp)[idx]);
    if (kind == 2) return T(reinterpret_cast<const double*>(amp)[idx]);
    if (kind == 3) return T(reinterpret_cast<const unsigned char*>(amp)[idx] ? 1 : 0);
    return T(1);
}

// ------------------------------------------------------------------ batches
// Field b of a batch (blockIdx.y) is the same problem at an offset: a copy of the parameter block with
// the base pointers advanced.  The parameter blocks live in SGPRs, so this is a handful of scalar ops.
template <typename T> PM_HD RowLoadNat<T> at_batch(RowLoadNat<T> p, int b) {
    if (p.real == 1 || p.real == 2)
        p.src = reinterpret_cast<const cx<T>*>(reinterpret_cast<const T*>(p.src) + int64_t(b) * p.bstride);
    else
        p.src += int64_t(b) * p.bstride;
    return p;
}
template <typename T> PM_HD RowStoreTiled<T> at_batch(RowStoreTiled<T> p, int b) { p.dst += int64_t(b) * p.bstride; return p; }
template <typename T> PM_HD RowStoreFold<T> at_batch(RowStoreFold<T> p, int b) { p.dst += int64_t(b) * p.bstride; return p; }
template <typename T> PM_HD RowStoreNat<T> at_batch(RowStoreNat<T> p, int b) { p.dst += int64_t(b) * p.bstride; return p; }
template <typename T> PM_HD RowLoadFold<T> at_batch(RowLoadFold<T> p, int b) { p.src += int64_t(b) * p.bstride; return p; }
template <typename T> PM_HD RowLoadTiled<T> at_batch(RowLoadTiled<T> p, int b) { p.src += int64_t(b) * p.bstride; return p; }
template <typename T> PM_HD RowLoadChirp<T> at_batch(RowLoadChirp<T> p, int) { return p; }    // one field per launch
template <typename T> PM_HD RowStoreChirp<T> at_batch(RowStoreChirp<T> p, int) { return p; }
template <typename T> PM_HD ColStoreTiled<T> at_batch(ColStoreTiled<T> p, int b) { p.dst += int64_t(b) * p.bstride; return p; }
template <typename T> PM_HD ColStoreTiledCrop<T> at_batch(ColStoreTiledCrop<T> p, int) { return p; }   // one field per launch
template <typename T> PM_HD ColLoadTiled<T> at_batch(ColLoadTiled<T> p, int b) { p.src += int64_t(b) * p.bstride; return p; }
template <typename T> PM_HD ColLoadNat<T> at_batch(ColLoadNat<T> p, int b) { p.src += int64_t(b) * p.bstride; return p; }
template <typename T> PM_HD MidMul<T> at_batch(MidMul<T> p, int b) {
    p.mul += int64_t(b) * p.bstride;
    if (p.mul_x) p.mul_x += int64_t(b) * p.bstride_x;
    return p;
}
template <typename T> PM_HD ColStoreNat<T> at_batch(ColStoreNat<T> p, int b) {
    if (p.epilogue == EPI_NONE)
        p.dst = reinterpret_cast<cx<T>*>(p.dst) + int64_t(b) * p.bstride;
    else
        p.dst = reinterpret_cast<T*>(p.dst) + int64_t(b) * p.bstride;
    if (p.mul) p.mul += int64_t(b) * p.mul_bstride;
    if (p.mul_x) p.mul_x += int64_t(b) * p.mul_bstride_x;
    return p;
}

// ------------------------------------------------------------------ row mode
// FULL: the window covers the whole axis and the sequence exists -> no per-element predicates at all
// MODE: 0 complex input, 1 real input, 2 pupil synthesis (real OPD + amplitude)
// (p.nt stays a runtime test inside the unrolled loop: every load then sits in a basic block of its own behind a scalar branch, which
// looks wasteful and is what keeps the sixteen 64-bit addresses from being formed -- and held -- all at once: as a template argument
// the row kernels grew by 12 - 40 registers and several lost a workgroup per CU, round 5, tools/kernel_table.py)
template <typename C, int ROT, bool FULL = false, int MODE = 0>
PM_HD void load_rot(const RowLoadNat<typename C::T>& p, int blk, ThreadPos pos,
                    cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    static_assert(C::CI == 1, "row mode");
    // a thread owns E sequences (rows): consecutive ones, or the pair (i, i + eoff) of a folded transform
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        const int unit = blk * C::BO + pos.bo;
        const int seq = p.eoff ? unit + e * p.eoff : unit * C::E + e;
        const bool ok = seq < p.nseq && (!p.eoff || unit < p.eoff);
        const cx<T>* row = p.src + int64_t(ok ? seq : 0) * p.ld - p.ax.off;
        const T* rrow = reinterpret_cast<const T*>(p.src) + int64_t(ok ? seq : 0) * p.ld - p.ax.off;   // real input / OPD
        const int64_t arow = int64_t(ok ? seq : 0) * p.amp_ld - p.ax.off;                               // amplitude row
        const int lo = p.ax.off, hi = ok ? p.ax.off + p.ax.len : -1;
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const int pp = slot_pos<C, ROT>(pos.t, m, p.ax.shift);
            cx<T> val = {T(0), T(0)};
            if (FULL || (pp >= lo && pp < hi)) {
                if constexpr (MODE == 3) {
                    const cx<T> ao = p.nt ? nt_load_cx(row + pp) : row[pp];
                    val = synth_value<T>(ao.y, ao.x, p.k2);
                } else if constexpr (MODE == 2)
                    val = synth_value<T>(rrow[pp], synth_amp<T>(p.amp, p.amp_kind, arow + pp), p.k2);
                else if constexpr (MODE == 1)
                    val.x = p.nt ? nt_load_s(rrow + pp) : rrow[pp];
                else
                    val = p.nt ? nt_load_cx(row + pp) : row[pp];
            }
            v[e][m] = val;
        }
        if (MODE == 0 && p.conj) {
#pragma unroll
            for (int m = 0; m < C::P; ++m) v[e][m].y = -v[e][m].y;
        }
    }
}

template <typename C, int MODE>
PM_HD void load_sel(const RowLoadNat<typename C::T>& p, int blk, ThreadPos pos, cx<typename C::T> (&v)[C::E][C::P]) {
    const int rot = rot_of<C>(p.ax.shift);
    const int unit_ = blk * C::BO + pos.bo;
    const bool full = p.ax.off == 0 && p.ax.len == C::N &&
                      (p.eoff ? (unit_ < p.eoff && unit_ + (C::E - 1) * p.eoff < p.nseq) : (unit_ * C::E + C::E - 1) < p.nseq);
    if (rot == 0) {
        if (full) load_rot<C, 0, true, MODE>(p, blk, pos, v);
        else load_rot<C, 0, false, MODE>(p, blk, pos, v);
    } else if (rot > 0) {
        if (full) load_rot<C, (C::P >= 2 ? C::P / 2 : 0), true, MODE>(p, blk, pos, v);
        else load_rot<C, (C::P >= 2 ? C::P / 2 : 0), false, MODE>(p, blk, pos, v);
    } else {
        load_rot<C, -1, false, MODE>(p, blk, pos, v);
    }
}

template <typename C>
PM_HD void load(const RowLoadNat<typename C::T>& p, int blk, ThreadPos pos, cx<typename C::T> (&v)[C::E][C::P]) {
    {
        if (p.real == 2) {
            load_sel<C, 2>(p, blk, pos, v);
            return;
        }
        if (p.real == 3) {
            load_sel<C, 3>(p, blk, pos, v);
            return;
        }
    }
    if (p.real) load_sel<C, 1>(p, blk, pos, v);
    else load_sel<C, 0>(p, blk, pos, v);
}

// Addresses of the tiled intermediate for row kernels.  Column c = t + m TPS of stored row `row` sits at element
//     (((c >> ltc) * nrows + row) << ltc) + (c & tcm),         TL = 1 << ltc the layout tile width, tcm = TL - 1.
// TPS and TL are powers of two, so whichever is larger the address splits into a per-thread part and a part that depends on the register
// slot m only (no carry between them: TL <= TPS makes (m TPS) & tcm zero, TL > TPS makes t + ((m TPS) & tcm) stay below TL):
//     thread:  (((t >> ltc) * nrows + row) << ltc) + (t & tcm)                     -> ONE 32-bit byte offset (vector register)
//     slot:    ((((m TPS) >> ltc) * nrows) << ltc) + ((m TPS) & tcm)               -> uniform, scalar registers
// The general form computed a 64-bit product, two 64-bit shifts and three 64-bit adds per point in vector registers -- 18 integer
// instructions beside the 8 floating-point ones of a fold (round 5; the row + fold kernel of the headline issued twice the VALU
// work of the column kernel, profiles/r04/headline_sq_counters.txt).  Byte offsets fit 32 bits: a slot's rows end below
// TPS * nrows * sizeof(complex) <= 512 * 8192 * 16 B.
template <typename C>
struct TiledRowAddr {
    using T = typename C::T;
    static constexpr int ES = int(sizeof(cx<T>));
    uint32_t voff;      // bytes
    int ltc, nrows;
    PM_HD TiledRowAddr(int t, int row, int ltc_, int nrows_) : ltc(ltc_), nrows(nrows_) {
        const int tcm = (1 << ltc) - 1;
        voff = uint32_t((((t >> ltc) * nrows + row) << ltc) + (t & tcm)) * uint32_t(ES);
    }
    PM_HD int64_t slot(int m) const {   // bytes, uniform
        const int mt = m * C::TPS, tcm = (1 << ltc) - 1;
        return ((int64_t(mt >> ltc) * nrows << ltc) + (mt & tcm)) * ES;
    }
    template <typename P> PM_HD P* at(P* base, int m) const {
        typedef typename std::conditional<std::is_const<P>::value, const char, char>::type B;
        return reinterpret_cast<P*>(reinterpret_cast<B*>(base) + slot(m) + voff);
    }
};

template <typename C>
PM_HD void store(const RowStoreTiled<typename C::T>& p, int blk, ThreadPos pos,
                 const cx<typename C::T> (&v)[C::E][C::P]) {
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        const int seq = (blk * C::BO + pos.bo) * C::E + e;
        if (seq >= p.nseq) continue;
        const TiledRowAddr<C> A(pos.t, seq, p.log_tc, p.nseq);
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            *A.at(p.dst, m) = v[e][m];
            if ((m & 3) == 3) PM_SCHED_FENCE();     // four slot bases (scalar register pairs) at a time: all sixteen spill scalars into vector registers
        }
    }
}

template <typename C>
PM_HD void store(const RowStoreFold<typename C::T>& p, int blk, ThreadPos pos,
                 const cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    static_assert(C::E == 2, "the fold pairs the two rows of a thread");
    const int i = blk * C::BO + pos.bo;       // pair index = logical row of the lower half
    if (i >= p.npairs) return;
    // rows that came in rotated by M/2 (swap): register set 0 holds the UPPER row of the pair -- the sum does not care and the
    // difference changes sign, which the twiddle absorbs (a runtime select of the register sets cost four v_cndmask per point)
    cx<T> w = p.twm[i];
    if (p.swap) w = {-w.x, -w.y};
    const TiledRowAddr<C> A(pos.t, i, p.log_tc, p.npairs);
    cx<T>* const d1 = p.dst + p.plane_stride;
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const cx<T> s = {v[0][m].x + v[1][m].x, v[0][m].y + v[1][m].y};
        const cx<T> d = {v[0][m].x - v[1][m].x, v[0][m].y - v[1][m].y};
        *A.at(p.dst, m) = s;
        *A.at(d1, m) = cmul(d, w);
        if ((m & 3) == 3) PM_SCHED_FENCE();     // or the products of all sixteen points are formed (and held) before the first store
    }
}

template <typename C, int ROT>
PM_HD void store_rot(const RowStoreNat<typename C::T>& p, int blk, ThreadPos pos,
                     const cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        const int unit = blk * C::BO + pos.bo;
        const int seq = p.eoff ? unit + e * p.eoff : unit * C::E + e;
        if (seq >= p.nseq || (p.eoff && unit >= p.eoff)) continue;
        int mrow = seq;
        if (p.use_ay) {
            mrow = p.ay.map(seq);
            if (mrow < 0) continue;
        }
        cx<T>* row = p.dst + int64_t(mrow) * p.ld - p.ax.off;
        const int lo = p.ax.off, hi = p.ax.off + p.ax.len;
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const int pp = slot_pos<C, ROT>(pos.t, m, p.ax.shift);
            if (pp < lo || pp >= hi) continue;
            cx<T> val = cscale(v[e][m], p.scale);
            if (p.conj) val.y = -val.y;
            if (p.nt)
                nt_store_cx(row + pp, val);
            else
                row[pp] = val;
        }
    }
}

template <typename C>
PM_HD void store(const RowStoreNat<typename C::T>& p, int blk, ThreadPos pos,
                 const cx<typename C::T> (&v)[C::E][C::P]) {
    const int rot = rot_of<C>(p.ax.shift);
    if (rot == 0)
        store_rot<C, 0>(p, blk, pos, v);
    else if (rot > 0)
        store_rot<C, (C::P >= 2 ? C::P / 2 : 0)>(p, blk, pos, v);
    else
        store_rot<C, -1>(p, blk, pos, v);
}

template <typename C>
PM_HD void load(const RowLoadTiled<typename C::T>& p, int blk, ThreadPos pos, cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    static_assert(C::CI == 1, "row mode");
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        const int seq = (blk * C::BO + pos.bo) * C::E + e;
        const bool ok = seq < p.nseq;
        const TiledRowAddr<C> A(pos.t, p.row0 + (ok ? seq : 0), p.log_tl, p.nrows);
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            cx<T> val = {T(0), T(0)};
            if (ok) val = *A.at(p.src, m);
            if (p.conj) val.y = -val.y;
            v[e][m] = val;
            if ((m & 3) == 3) PM_SCHED_FENCE();
        }
    }
}

template <typename C>
PM_HD void load(const RowLoadChirp<typename C::T>& p, int blk, ThreadPos pos, cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    static_assert(C::CI == 1, "row mode");
    const int n2 = p.in.ax.n;
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        const int seq = (blk * C::BO + pos.bo) * C::E + e;
        const bool ok = seq < p.nseq;
        const cx<T> wy = p.w1[ok ? seq : 0];
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const int c = pos.t + m * C::TPS;
            cx<T> val = {T(0), T(0)};
            if (ok && c < n2) val = cmul(fetch2d(p.in, seq, c), cmul(wy, p.w2[c]));
            v[e][m] = val;
        }
    }
}

template <typename C>
PM_HD void load(const RowLoadFold<typename C::T>& p, int blk, ThreadPos pos, cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    static_assert(C::CI == 1 && C::E == 2, "the unfold rebuilds the two rows of a thread");
    const int n = blk * C::BO + pos.bo;
    const bool ok = n < p.npairs;
    const cx<T> w = p.twm[ok ? n : 0];
    const TiledRowAddr<C> A(pos.t, ok ? n : 0, p.log_tl, p.npairs);
    const cx<T>* const s1 = p.src + p.plane_stride;
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        cx<T> a0 = {T(0), T(0)}, a1 = {T(0), T(0)};
        if (ok) {
            a0 = *A.at(p.src, m);
            a1 = *A.at(s1, m);
        }
        const cx<T> b = cmulc(a1, w);      // conj(W_M^n) A1[n]
        cx<T> lo = {a0.x + b.x, a0.y + b.y}, hi = {a0.x - b.x, a0.y - b.y};
        if (p.conj) {
            lo.y = -lo.y;
            hi.y = -hi.y;
        }
        v[0][m] = lo;
        v[1][m] = hi;
        if ((m & 3) == 3) PM_SCHED_FENCE();
    }
}

// ------------------------------------------------------------------ col mode
// unit handled by workgroup g: XCD-aware sibling grouping.  Workgroups g, g+8, ..., g+8(G-1) run on the
// same XCD (block b -> XCD b % 8) back to back; they get G ADJACENT units, so the pieces of one cache line
// (or of one layout-tile row) that different workgroups touch meet in that XCD's L2.  Speed only: any
// placement gives the same results.
PM_HD int group_remap(int g, int total, int log_g) {
    const int G = 1 << log_g;
    if (log_g == 0 || (total % (8 * G)) != 0) return g;
    const int xcd = g & 7, idx = g >> 3;
    return (((idx >> log_g) << 3) + xcd) * G + (idx & (G - 1));
}
PM_HD int pair_remap(int g, int total) { return group_remap(g, total, 1); }

template <typename C, int ROT, bool FULL = false>
PM_HD void load_rot(const ColLoadTiled<typename C::T>& p, int tile, ThreadPos pos,
                    cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    constexpr int TC = C::CI * C::E;
    const bool ok = tile < p.ntiles;
    // element (row q, col) of workgroup tile `tile` lives in layout tile (tile >> log_k) at
    // [(q)*TL + (tile & (k-1))*TC + col]; q = p - off
    const int TL = TC << p.log_k;
    const int tl = (ok ? tile : 0) >> p.log_k, sub = (ok ? tile : 0) & ((1 << p.log_k) - 1);
    const cx<T>* base = p.src + (int64_t(tl) * p.nrows - p.ay.off) * TL + sub * TC + pos.cl * C::E;
    const int lo = p.ay.off, hi = ok ? p.ay.off + p.ay.len : -1;
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int pp = slot_pos<C, ROT>(pos.t, m, p.ay.shift);
        if (FULL || (pp >= lo && pp < hi)) {
            const cx<T>* a = base + int64_t(pp) * TL;
            if constexpr (C::E == 2 && sizeof(T) == 4) {
                const Vec4<T> w = *reinterpret_cast<const Vec4<T>*>(a);  // two adjacent complex64 columns
                v[0][m] = {w.a, w.b};
                v[1][m] = {w.c, w.d};
            } else {
#pragma unroll
                for (int e = 0; e < C::E; ++e) v[e][m] = a[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < C::E; ++e) v[e][m] = {T(0), T(0)};
        }
    }
}

template <typename C>
PM_HD void load(const ColLoadTiled<typename C::T>& p, int tile, ThreadPos pos,
                cx<typename C::T> (&v)[C::E][C::P]) {
    const int rot = rot_of<C>(p.ay.shift);
    const bool full = p.ay.off == 0 && p.ay.len == C::N && tile < p.ntiles;
    if (rot == 0) {
        if (full) load_rot<C, 0, true>(p, tile, pos, v);
        else load_rot<C, 0>(p, tile, pos, v);
    } else if (rot > 0) {
        if (full) load_rot<C, (C::P >= 2 ? C::P / 2 : 0), true>(p, tile, pos, v);
        else load_rot<C, (C::P >= 2 ? C::P / 2 : 0)>(p, tile, pos, v);
    } else {
        load_rot<C, -1>(p, tile, pos, v);
    }
}

template <typename C>
PM_HD void load(const ColLoadNat<typename C::T>& p, int tile, ThreadPos pos,
                cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    constexpr int TC = C::CI * C::E;
    const int col0 = tile * TC + pos.cl * C::E;
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int q = p.ay.map(pos.t + m * C::TPS);
        const cx<T>* a = p.src + int64_t(q >= 0 ? q : 0) * p.ld + col0;
        bool done = false;
        if constexpr (C::E == 2 && sizeof(T) == 4) {
            if (p.vec_ok && q >= 0 && col0 + 1 < p.ncols) {
                const Vec4<T> w = *reinterpret_cast<const Vec4<T>*>(a);
                v[0][m] = {w.a, w.b};
                v[1][m] = {w.c, w.d};
                done = true;
            }
        }
        if (!done) {
#pragma unroll
            for (int e = 0; e < C::E; ++e) {
                cx<T> val = {T(0), T(0)};
                if (q >= 0 && col0 + e < p.ncols) val = a[e];
                v[e][m] = val;
            }
        }
        if (p.conj) {
#pragma unroll
            for (int e = 0; e < C::E; ++e) v[e][m].y = -v[e][m].y;
        }
    }
}

template <typename C>
PM_HD void store(const ColStoreTiled<typename C::T>& p, int tile, ThreadPos pos,
                 const cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    constexpr int TC = C::CI * C::E;
    if (tile >= p.ntiles) return;
    const int TL = TC << p.log_k;
    const int tl = tile >> p.log_k, sub = tile & ((1 << p.log_k) - 1);
    cx<T>* base = p.dst + int64_t(tl) * p.nrows * TL + sub * TC + pos.cl * C::E;
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        cx<T>* a = base + int64_t(pos.t + m * C::TPS) * TL;
        if constexpr (C::E == 2 && sizeof(T) == 4) {
            *reinterpret_cast<Vec4<T>*>(a) = Vec4<T>{v[0][m].x, v[0][m].y, v[1][m].x, v[1][m].y};
        } else {
#pragma unroll
            for (int e = 0; e < C::E; ++e) a[e] = v[e][m];
        }
    }
}

template <typename C>
PM_HD void store(const ColStoreTiledCrop<typename C::T>& p, int tile, ThreadPos pos,
                 const cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    constexpr int TC = C::CI * C::E;
    if (tile >= p.ntiles) return;
    const int TL = TC << p.log_k;
    const int tl = tile >> p.log_k, sub = tile & ((1 << p.log_k) - 1);
    cx<T>* base = p.dst + int64_t(tl) * p.nrows * TL + sub * TC + pos.cl * C::E;
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int row = pos.t + m * C::TPS;
        if (row >= p.nrows) continue;
        cx<T>* a = base + int64_t(row) * TL;
        if constexpr (C::E == 2 && sizeof(T) == 4) {
            *reinterpret_cast<Vec4<T>*>(a) = Vec4<T>{v[0][m].x, v[0][m].y, v[1][m].x, v[1][m].y};
        } else {
#pragma unroll
            for (int e = 0; e < C::E; ++e) a[e] = v[e][m];
        }
    }
}

// v[e][m] *= H[k = t + m*TPS][col], then conjugate (the inverse transform that follows is conj(FFT(conj .))).
// The multiplier kind is resolved OUTSIDE the unrolled loops (template argument): a runtime test inside them made
// the compiler issue the loads of both kinds for all 32 elements and spill.
template <typename C, int KIND>
PM_HD void mid_multiply_conj_kind(const MidMul<typename C::T>& p, int tile, ThreadPos pos,
                                  cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    constexpr int TC = C::CI * C::E;
    const int col0 = tile * TC + pos.cl * C::E;
    cx<T> hx[C::E];
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        hx[e] = {T(1), T(0)};
        if (KIND == MUL_SEPARABLE && col0 + e < p.ncols) hx[e] = p.mul_x[col0 + e];
    }
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int k = pos.t + m * C::TPS;
        cx<T> hy = {T(1), T(0)};
        if (KIND == MUL_SEPARABLE) hy = p.mul[p.ystep > 1 ? k * p.ystep : k];
        cx<T> hf[C::E];
        if constexpr (KIND == MUL_FULL) {
            bool done = false;
            if constexpr (C::E == 2 && sizeof(T) == 4) {
                if (p.vec_ok && col0 + 1 < p.ncols) {
                    const Vec4<T> w = *reinterpret_cast<const Vec4<T>*>(p.mul + int64_t(k) * p.ld + col0);
                    hf[0] = {w.a, w.b};
                    hf[1] = {w.c, w.d};
                    done = true;
                }
            }
            if (!done) {
#pragma unroll
                for (int e = 0; e < C::E; ++e)
                    hf[e] = (col0 + e < p.ncols) ? p.mul[int64_t(k) * p.ld + col0 + e] : cx<T>{T(1), T(0)};
            }
        }
#pragma unroll
        for (int e = 0; e < C::E; ++e) {
            const cx<T> h = (KIND == MUL_FULL) ? hf[e] : cmul(hy, hx[e]);
            const cx<T> x = p.conj ? cmulc(v[e][m], h) : cmul(v[e][m], h);
            v[e][m] = {x.x, -x.y};
        }
    }
}

template <typename C>
PM_HD void mid_multiply_conj(const MidMul<typename C::T>& p, int tile, ThreadPos pos,
                             cx<typename C::T> (&v)[C::E][C::P]) {
    if (p.kind == MUL_FULL)
        mid_multiply_conj_kind<C, MUL_FULL>(p, tile, pos, v);
    else
        mid_multiply_conj_kind<C, MUL_SEPARABLE>(p, tile, pos, v);
}

// one output element (row bin k, column bin c) through the full epilogue -- shared by the
// generic store path and the direct-DFT kernels
template <typename T>
PM_HD void store_one(const ColStoreNat<T>& p, int k, int c, cx<T> x) {
    const int qy = p.ay.map(k);
    const int qx = p.ax.map(c);
    if (qy < 0 || qx < 0) return;
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

template <typename C>
PM_HD void store(const RowStoreChirp<typename C::T>& p, int blk, ThreadPos pos, const cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
#pragma unroll
    for (int e = 0; e < C::E; ++e) {
        const int seq = (blk * C::BO + pos.bo) * C::E + e;
        if (seq >= p.n1) continue;
        const cx<T> wy = p.w1[seq];
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const int c = pos.t + m * C::TPS;
            if (c >= p.n2) continue;
            cx<T> val = v[e][m];
            if (p.conj) val.y = -val.y;
            store_one(p.out, seq, c, cmul(val, cmul(wy, p.w2[c])));
        }
    }
}

// fast path: no multiplier, full-width aligned columns (no crop along x, even rotation), complex or
// |.|^2 output.  Row window handled by a compare; addresses affine in the slot index.
template <typename C, int ROT, bool FULL = false>
PM_HD void store_fast(const ColStoreNat<typename C::T>& p, int tile, ThreadPos pos,
                      const cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    constexpr int TC = C::CI * C::E;
    const int col0 = tile * TC + pos.cl * C::E;
    if (col0 >= p.ax.n) return;
    int qx = col0 + p.ax.shift;              // columns: rotation only (len == n, off == 0)
    if (qx >= p.ax.n) qx -= p.ax.n;
    const int lo = p.ay.off, hi = p.ay.off + p.ay.len;
    if (p.epilogue == EPI_NONE) {
        cx<T>* base = reinterpret_cast<cx<T>*>(p.dst) - int64_t(p.ay.off) * p.ld + qx;
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const int pp = slot_pos<C, ROT>(pos.t, m, p.ay.shift);
            if (!FULL && (pp < lo || pp >= hi)) continue;
            cx<T>* a = base + int64_t(pp) * p.ld;
            cx<T> val[C::E];
#pragma unroll
            for (int e = 0; e < C::E; ++e) {
                val[e] = cscale(v[e][m], p.scale);
                if (p.conj) val[e].y = -val[e].y;
            }
            if constexpr (C::E == 2 && sizeof(T) == 4) {
                const Vec4<T> w{val[0].x, val[0].y, val[1].x, val[1].y};
                if (p.nt)
                    nt_store_v4(reinterpret_cast<Vec4<T>*>(a), w);
                else
                    *reinterpret_cast<Vec4<T>*>(a) = w;
            } else {
#pragma unroll
                for (int e = 0; e < C::E; ++e) {
                    if (p.nt)
                        nt_store_cx(a + e, val[e]);
                    else
                        a[e] = val[e];
                }
            }
        }
    } else {
        T* base = reinterpret_cast<T*>(p.dst) - int64_t(p.ay.off) * p.ld + qx;
        const T s2 = p.scale * p.scale;
        // the thread's E = 2 adjacent columns go out as ONE 8-byte access when the output allows it (vec_ok bit 1): two
        // 4-byte stores per row doubled the store instructions of the |.|^2 epilogues (measured 65 -> 5x us at 4096^2)
        const bool pair = C::E == 2 && (p.vec_ok & 2);
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const int pp = slot_pos<C, ROT>(pos.t, m, p.ay.shift);
            if (!FULL && (pp < lo || pp >= hi)) continue;
            T* a = base + int64_t(pp) * p.ld;
            T i2[C::E];
#pragma unroll
            for (int e = 0; e < C::E; ++e) i2[e] = (v[e][m].x * v[e][m].x + v[e][m].y * v[e][m].y) * s2;
            if constexpr (C::E == 2) {
                if (pair) {
                    cx<T>* a2 = reinterpret_cast<cx<T>*>(a);       // a pair of reals, same size and alignment as one complex
                    if (p.epilogue == EPI_ABS2) {
                        if (p.nt)
                            nt_store_cx(a2, cx<T>{i2[0], i2[1]});
                        else
                            *a2 = cx<T>{i2[0], i2[1]};
                    } else {
                        const cx<T> old = *a2;
                        *a2 = cx<T>{old.x + p.weight * i2[0], old.y + p.weight * i2[1]};
                    }
                    continue;
                }
            }
#pragma unroll
            for (int e = 0; e < C::E; ++e) {
                if (p.epilogue == EPI_ABS2) {
                    if (p.nt)
                        nt_store_s(a + e, i2[e]);
                    else
                        a[e] = i2[e];
                } else {
                    a[e] += p.weight * i2[e];
                }
            }
        }
    }
}

template <typename C>
PM_HD void store(const ColStoreNat<typename C::T>& p, int tile, ThreadPos pos,
                 const cx<typename C::T> (&v)[C::E][C::P]) {
    constexpr int TC = C::CI * C::E;
    const int rot = rot_of<C>(p.ay.shift);
    const bool fast = p.mul_kind == MUL_NONE && p.vec_ok && p.ax.off == 0 && p.ax.len == p.ax.n &&
                      (p.ax.n % TC) == 0 && (p.ax.shift % TC) == 0 && rot >= 0;
    if (fast) {
        const bool full = p.ay.off == 0 && p.ay.len == C::N;
        if (rot == 0) {
            if (full) store_fast<C, 0, true>(p, tile, pos, v);
            else store_fast<C, 0>(p, tile, pos, v);
        } else {
            if (full) store_fast<C, (C::P >= 2 ? C::P / 2 : 0), true>(p, tile, pos, v);
            else store_fast<C, (C::P >= 2 ? C::P / 2 : 0)>(p, tile, pos, v);
        }
        return;
    }
    const int col0 = tile * TC + pos.cl * C::E;
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int k = pos.t + m * C::TPS;
#pragma unroll
        for (int e = 0; e < C::E; ++e)
            if (col0 + e < p.ax.n) store_one(p, k, col0 + e, v[e][m]);
    }
}

}  // namespace pm
