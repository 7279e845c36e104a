// Grouped-wavelength transform pair that keeps FOUR waves per SIMD (round 4) -- the form of fft_spectral.h that survives at 4096-point
// rows.  The polychromatic recipe (docs/source/how-tos/Polychromatic Propagation.ipynb cell 3: for each wavelength pupil = amp exp(i k opd),
// focus, |.|^2, weighted sum) as a loop of transform pairs moves 8 (packed map) + 8 + 8 (intermediate) + 4 + 4 (accumulator) = 32 bytes per
// sample and wavelength; in groups of G it needs 16 + 16 / G.  fft_spectral.h's kernels get there by holding the packed map AND the field of
// a row pair (256 VGPRs + AGPR spills: one wave per SIMD) and 32 accumulators beside a generic column pass (256 VGPRs): below 4096^2 the
// saved bytes win anyway, at 4096^2 they lost to the plain loop.  Here
//   rows     the two sequences a thread owns are TWO WAVELENGTHS OF ONE ROW, not two rows: the (amplitude, OPD) pairs are loaded once (32
//            registers), both pupils are synthesised from them and the raw values are dead before the transform starts -- the register
//            footprint of the plain two-rows-per-thread row pass, with the stage twiddles shared by the pair exactly as there.  The
//            radix-2 fold of the column transform (RowStoreFold) needs rows i and i + M/2 together: they sit in the two halves of a
//            512-thread workgroup and meet through LDS before the row transform (the fold is linear, so it commutes with it):
//            plane 0 row i = x_i + x_{i+M/2}, plane 1 row i = (x_i - x_{i+M/2}) W_M^i.
//   columns  lean addressing (one uniform base per register slot + one 32-bit per-thread offset, PfAddr) frees the registers the 32
//            accumulators need: two 512-thread workgroups per CU, the accumulator read and written once per group.
// complex64 only (the BASELINE's polychromatic configuration is fp32); everything else stays on fft_spectral.h / the loop.
// The per-thread pieces are __host__ __device__: tools/emu_fft.cpp runs them on the CPU.
#pragma once
#include "fft_io.h"
#if defined(__HIPCC__)
#include "fft_spectral_types.h"
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define PM_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define PM_UNIFORM(x) (x)
#endif

namespace pm {

// ------------------------------------------------------------------ parameter blocks (host-filled, kernel arguments)
template <typename T>
struct Sp2Row {
    const cx<T>* src;       // packed (amplitude, OPD) map, one pair per sample
    int64_t ld;             // pairs between consecutive rows
    int off, len;           // stored window of the length-N row: position p -> memory index p - off, zero amplitude outside
    int rot;                // slot rotation of the load: 0, or P/2 for a rotation by N/2 (ifftshift)
    int nrows;              // stored rows (memory rows of the map)
    int nt;                 // non-temporal loads (the group covers the whole wavelength list: the map is read once)
    cx<T>* dst;             // tiled intermediates: wavelength b of the group at dst + b * fstride
    int64_t fstride;
    int64_t plane_stride;   // fold: elements between plane 0 and plane 1
    int drows;              // rows per layout tile block: M/2 (fold) or nrows
    int log_tl;             // log2(layout tile width)
    int swap;               // fold: the input rows are rotated by M/2 -- memory row i is logical row i + M/2
    const cx<T>* twm;       // fold: W_M^k
};

template <typename T>
struct Sp2Col {
    const cx<T>* src;       // intermediate of wavelength 0; plane y (blockIdx.y) at + y * plane_stride, wavelength b at + b * fstride
    int64_t fstride, plane_stride;
    int nrows;              // rows stored per layout tile block
    int off, len;           // stored window of the length-N column (zero rows outside are synthesised)
    int rot;                // slot rotation of the load (input rows rotated by N/2): 0 or P/2
    int log_k;
    T* dst;                 // accumulator image; plane y at + y * out_plane (the odd rows of a folded transform: one row further)
    int64_t ld, out_plane;
    int orot;               // slot rotation of the output rows: 0 or P/2
    int qshift, ncols;      // output column of bin c: (c + qshift) mod ncols
    T s2;                   // scale^2
};

// ------------------------------------------------------------------ rows
// raw[m] = (amplitude, OPD) at position t + ((m + ROT) mod P) TPS of memory row `memrow` (zero outside the window)
template <typename C, int ROT, bool FULL>
PM_HD void sp2_row_load(const Sp2Row<typename C::T>& g, int memrow, bool ok, int t, cx<typename C::T> (&raw)[C::P]) {
    using T = typename C::T;
    constexpr int ES = int(sizeof(cx<T>));
    const char* rb = reinterpret_cast<const char*>(g.src) + (int64_t(ok ? memrow : 0) * g.ld - g.off) * ES;   // uniform
    const uint32_t voff = uint32_t(t) * ES;
    const int lo = g.off, hi = ok ? g.off + g.len : -1;
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int mm = (m + ROT) & (C::P - 1);
        const int pp = t + mm * C::TPS;
        cx<T> val = {T(0), T(0)};
        if (FULL || (pp >= lo && pp < hi)) {
            const cx<T>* a = reinterpret_cast<const cx<T>*>(rb + int64_t(mm) * C::TPS * ES + voff);
            val = g.nt ? nt_load_cx(a) : *a;
        }
        raw[m] = val;
    }
}

template <typename C>
PM_HD void sp2_row_load_sel(const Sp2Row<typename C::T>& g, int memrow, bool ok, int t, cx<typename C::T> (&raw)[C::P]) {
    const bool full = ok && g.off == 0 && g.len == C::N;
    if (g.rot == 0) {
        if (full) sp2_row_load<C, 0, true>(g, memrow, ok, t, raw);
        else sp2_row_load<C, 0, false>(g, memrow, ok, t, raw);
    } else {
        if (full) sp2_row_load<C, (C::P >= 2 ? C::P / 2 : 0), true>(g, memrow, ok, t, raw);
        else sp2_row_load<C, (C::P >= 2 ? C::P / 2 : 0), false>(g, memrow, ok, t, raw);
    }
}

// both pupils of a row from one set of raw values (Wavefront.from_amp_and_phase, prysm/propagation/wavefront.py:58-79)
template <typename C>
PM_HD void sp2_synth(const cx<typename C::T> (&raw)[C::P], double k2a, double k2b, cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        v[0][m] = synth_value<T>(raw[m].y, raw[m].x, k2a);
        v[1][m] = synth_value<T>(raw[m].y, raw[m].x, k2b);
    }
}

// the fold before the row transform, through LDS: half `bo` of the workgroup holds memory row i + bo M/2
template <typename C>
PM_HD void sp2_fold_write(const cx<typename C::T> (&v)[C::P], int t, int bo, cx<typename C::T>* lds) {
#pragma unroll
    for (int m = 0; m < C::P; ++m) lds[(bo * C::P + m) * C::TPS + t] = v[m];
}
template <typename C>
PM_HD void sp2_fold_combine(cx<typename C::T> (&v)[C::P], int t, int bo, const cx<typename C::T>* lds, cx<typename C::T> w, int swap) {
    using T = typename C::T;
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const cx<T> o = lds[((1 - bo) * C::P + m) * C::TPS + t];
        if (bo == 0) {
            v[m] = v[m] + o;                                    // plane 0: y[i] + y[i + M/2]
        } else {
            const cx<T> d = swap ? v[m] - o : o - v[m];         // plane 1: (y[i] - y[i + M/2]) W_M^i; this half holds memory row i + M/2
            v[m] = cmul(d, w);
        }
    }
}

// element (row `unit` of a plane, column c = t + m TPS) -> tiled address ((c >> ltl) * drows + unit) << ltl + (c & (TL - 1))
template <typename C>
PM_HD void sp2_row_store(const Sp2Row<typename C::T>& g, int unit, int plane, int wl, int t, const cx<typename C::T> (&v)[C::P]) {
    using T = typename C::T;
    constexpr int ES = int(sizeof(cx<T>));
    char* db = reinterpret_cast<char*>(g.dst + int64_t(wl) * g.fstride + int64_t(plane) * g.plane_stride);   // uniform
    const int ltl = g.log_tl, tlm = (1 << ltl) - 1;
    if (C::TPS <= (1 << ltl)) {         // the layout tile is a multiple of TPS columns wide
        const uint32_t voff = uint32_t(t) * ES;
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const int c0 = m * C::TPS;
            const int64_t a = ((int64_t(c0 >> ltl) * g.drows + unit) << ltl) + (c0 & tlm);
            *reinterpret_cast<cx<T>*>(db + a * ES + voff) = v[m];
        }
    } else {                            // TPS is a multiple of the tile width
        const uint32_t voff = uint32_t((((t >> ltl) * g.drows) << ltl) + (t & tlm)) * ES;
        const int64_t a0 = int64_t(unit) << ltl, step = (int64_t(C::TPS >> ltl) * g.drows) << ltl;
#pragma unroll
        for (int m = 0; m < C::P; ++m) *reinterpret_cast<cx<T>*>(db + (a0 + m * step) * ES + voff) = v[m];
    }
}

// ------------------------------------------------------------------ columns
// lean tile addressing (the PfAddr of fft_kernels.h, restated here without device-only code so that the emulator can run it)
template <typename C>
struct Sp2Addr {
    using T = typename C::T;
    static constexpr int TC = C::CI * C::E;
    static constexpr int ES = int(sizeof(cx<T>));
    uint32_t vdata;   // byte offset of this thread's first element inside a tile block
    int row0, TL;
    int64_t mstep;    // bytes between register slots
    PM_HD Sp2Addr(ThreadPos pos, int log_k) {
        TL = TC << log_k;
        row0 = pos.t;
        vdata = uint32_t(pos.t * TL + pos.cl * C::E) * ES;
        mstep = int64_t(C::TPS) * TL * ES;
    }
    PM_HD int64_t tile_off(int tile, int nrows, int log_k) const {
        const int tl = tile >> log_k, sub = tile & ((1 << log_k) - 1);
        return (int64_t(tl) * nrows * TL + sub * TC) * ES;
    }
};

template <typename C, int ROT>
PM_HD void sp2_col_load(const Sp2Col<typename C::T>& g, const cx<typename C::T>* src, int tile, const Sp2Addr<C>& A,
                        cx<typename C::T> (&v)[C::E][C::P]) {
    using T = typename C::T;
    static_assert(C::E == 2 && sizeof(T) == 4, "two adjacent complex64 columns per thread");
    const char* tb = reinterpret_cast<const char*>(src) + A.tile_off(tile, g.nrows, g.log_k) - int64_t(g.off) * A.TL * Sp2Addr<C>::ES;
    const int lo = g.off, hi = g.off + g.len;
    const bool full = g.len == C::N;
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int mm = (m + ROT) & (C::P - 1);
        const int pp = A.row0 + mm * C::TPS;
        if (full || (pp >= lo && pp < hi)) {
            const Vec4<T> w = *reinterpret_cast<const Vec4<T>*>(tb + mm * A.mstep + A.vdata);
            v[0][m] = {w.a, w.b};
            v[1][m] = {w.c, w.d};
        } else {
            v[0][m] = {T(0), T(0)};
            v[1][m] = {T(0), T(0)};
        }
    }
}

template <typename C>
PM_HD void sp2_accumulate(typename C::T (&acc)[C::E][C::P], const cx<typename C::T> (&v)[C::E][C::P], typename C::T wb, typename C::T s2) {
#pragma unroll
    for (int e = 0; e < C::E; ++e)
#pragma unroll
        for (int m = 0; m < C::P; ++m) acc[e][m] += wb * ((v[e][m].x * v[e][m].x + v[e][m].y * v[e][m].y) * s2);
}

// image[row t + ((m + OROT) mod P) TPS][qx .. qx + 1] += acc[.][m]: one 8-byte read-modify-write per slot
template <typename C, int OROT>
PM_HD void sp2_col_store(const Sp2Col<typename C::T>& g, typename C::T* dst, int tile, ThreadPos pos, const typename C::T (&acc)[C::E][C::P]) {
    using T = typename C::T;
    constexpr int TC = C::CI * C::E;
    int qx = tile * TC + pos.cl * C::E + g.qshift;
    if (qx >= g.ncols) qx -= g.ncols;
    char* ob = reinterpret_cast<char*>(dst);
    const uint32_t voff = uint32_t(int64_t(pos.t) * g.ld + qx) * uint32_t(sizeof(T));
    const int64_t mstep = int64_t(C::TPS) * g.ld * int64_t(sizeof(T));
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int mm = (m + OROT) & (C::P - 1);
        cx<T>* a = reinterpret_cast<cx<T>*>(ob + mm * mstep + voff);
        const cx<T> old = *a;
        *a = cx<T>{old.x + acc[0][m], old.y + acc[1][m]};
    }
}

#if defined(__HIPCC__)
// ------------------------------------------------------------------ entry points (fft_spectral2_f32.hip)
// rows: logn = log2(row length), fold as in the plan; grid = row pairs (fold) or ceil(nrows / rows per workgroup)
int launch_row_spectral2(int logn, bool fold, bool keep, const Sp2Row<float>& g, const cx<float>* tw, const Spectral& w, hipStream_t st);
// columns: logm = log2(tile length), ntiles workgroups x nplanes
int launch_col_spectral2(int logm, const Sp2Col<float>& g, const cx<float>* tw, int ntiles, int log_g, const Spectral& w, hipStream_t st,
                         int nplanes);
#endif

}  // namespace pm
