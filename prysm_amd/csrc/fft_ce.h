// Composite lengths on the register engine: 1000, 1500, 2000, 3000 ... -- the decimal grids and Q = 1.5 pads of the reference's users
// (prysm/propagation/fft.py:7-25 through prysm/fttools.py:23-31) -- as the power-of-two engine runs them (fft_engine.h): the sequence
// lives in the REGISTERS of its threads from the global load to the global store, the LDS is only the exchange fabric between stages,
// and every index is a compile-time constant of the plan.  The general kernel (fft_mixed.h) keeps the sequence in LDS, reads its plan at
// run time and pays ~20 integer instructions per point and stage for it (an all-off ablation of its 3000-point kernels ran 17 us of the
// 62; profiles/r04/exp_mix_ablate.log); it stays the route of every length that has no plan here.
//
// Plan: N = R0 R1 .. R_{S-1}, every later factor a divisor of the first one, P = R0 points per thread, TS = N / P threads per sequence.
// Decimation in frequency.  Thread t loads v[m] = x[t + TS m].  Before stage s the thread holds P / R_s butterflies u of R_s points m:
//
//     v[m Q_s + u] = A_{s-1}[c = g + G_s u][i' + M_s m],      i' = t mod M_s,  g = t div M_s,  Q_s = P / R_s,  G_s = TS / M_s
//
// with M_s = N / (R0 .. R_s) the length that is left after stage s and A_{s-1}[c][.] the M_{s-1}-point sub-problem of the output bins
// congruent to c.  The stage is a twiddle-free DFT over m, then the twiddle W_{M_{s-1}}^{i' k}; register k Q_s + u then holds
// A_s[c + C_s k][i'] (C_s = R0 .. R_{s-1}).  Because C_s = G_s Q_s, the REGISTER INDEX r = k Q_s + u is exactly the part of the new
// bin prefix the thread index does not carry, so the exchange is
//
//     write:  lds[r SB + T] = v[r]                                   (T: the thread's index in the workgroup -- consecutive lanes)
//     read :  v[m Q + u]   = lds[(r1 + R u) SB + pos(gw M_{s-1} + i'' + M_s m)]     g' = t div M_s = gw + G_{s-1} r1
//
// and after the last stage v[r] = X[t + TS r]: natural order in the layout of the input, stores as coalesced as the loads.
// SB (slots between the rows of the exchange) is chosen per exchange so that the reads spread over the banks (tools/ce_banks.py).
//
// Small DFTs: 2, 4, 8, 16 from the engine, 3 and 5 by the symmetric half sums (fft_mixed.h), and the composite factors 6 .. 30 by the
// prime-factor map (Good-Thomas: their factors are coprime, so there are NO inner twiddles, only a renaming of registers).
//
// Everything here is __host__ __device__: tools/emu_ce.cpp runs the same functions on the CPU, thread by thread and phase by phase.
#pragma once
#include "fft_mixed.h"

namespace pm {

template <int R0_, int R1_, int R2_ = 1, int R3_ = 1>
struct CePlan {
    static constexpr int S = R3_ > 1 ? 4 : (R2_ > 1 ? 3 : 2);
    static constexpr int P = R0_, N = R0_ * R1_ * R2_ * R3_, TS = N / R0_;
    static constexpr int radix(int s) { return s == 0 ? R0_ : (s == 1 ? R1_ : (s == 2 ? R2_ : R3_)); }
    static constexpr int m(int s) {      // points left after stage s
        int v = N;
        for (int i = 0; i <= s; ++i) v /= radix(i);
        return v;
    }
    static constexpr int g(int s) { return TS / m(s); }
    static constexpr int q(int s) { return P / radix(s); }
    static constexpr int tw_step(int s) { return s == 0 ? 1 : N / m(s - 1); }     // W_{M_{s-1}} = W_N^step
    static_assert(R0_ % R1_ == 0 && R0_ % R2_ == 0 && R0_ % R3_ == 0, "later factors divide the first one");
    static_assert(R1_ > 1 && (R3_ == 1 || R2_ > 1), "factors in order");
};

// ---------------------------------------------------------------------------
// small DFTs
// ---------------------------------------------------------------------------
template <typename T, int R>
struct CeDft {
    static PM_HD void run(cx<T>* a) { MixDft<T, R>::run(a); }      // 2 4 8 16 (engine)
};
// acc + a s, a -+ i b on whole complex values: one packed instruction each in the translation units built with PM_PACKED_F32
template <typename T> PM_HD cx<T> ce_fma(cx<T> a, T s, cx<T> acc) { return {acc.x + a.x * s, acc.y + a.y * s}; }
template <typename T> PM_HD cx<T> ce_add_mi(cx<T> a, cx<T> b) { return {a.x + b.y, a.y - b.x}; }
template <typename T> PM_HD cx<T> ce_sub_mi(cx<T> a, cx<T> b) { return {a.x - b.y, a.y + b.x}; }
#if defined(__HIP_DEVICE_COMPILE__) && defined(PM_PACKED_F32)
__device__ __forceinline__ cx<float> ce_fma(cx<float> a, float s, cx<float> acc) {
    pm_v2f r, k;      // s is a compile-time constant: the pair sits in scalar registers
    k[0] = s;
    k[1] = s;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(pm_pk(a)), "s"(k), "v"(pm_pk(acc)));
    return pm_un(r);
}
__device__ __forceinline__ cx<float> ce_add_mi(cx<float> a, cx<float> b) { return add_mi(a, b); }
__device__ __forceinline__ cx<float> ce_sub_mi(cx<float> a, cx<float> b) { return sub_mi(a, b); }
#endif
// odd primes by the symmetric half sums (fft_mixed.h mix_dft_odd) written on complex values
template <typename T, int R>
PM_HD void ce_dft_odd(cx<T>* a) {
    constexpr int H = (R - 1) / 2;
    cx<T> p[H + 1], q[H + 1];
    cx<T> sum = a[0];
#pragma unroll
    for (int m = 1; m <= H; ++m) {
        p[m] = a[m] + a[R - m];
        q[m] = a[m] - a[R - m];
        sum = sum + p[m];
    }
    const cx<T> x0 = a[0];
    a[0] = sum;
#pragma unroll
    for (int k = 1; k <= H; ++k) {
        cx<T> A = x0, B = cscale(q[1], T(MixRoots<R>::tab.s[k % R]));
#pragma unroll
        for (int m = 1; m <= H; ++m) {
            A = ce_fma(p[m], T(MixRoots<R>::tab.c[(m * k) % R]), A);
            if (m > 1) B = ce_fma(q[m], T(MixRoots<R>::tab.s[(m * k) % R]), B);
        }
        a[k] = ce_add_mi(A, B);
        a[R - k] = ce_sub_mi(A, B);
    }
}
template <typename T> struct CeDft<T, 3> { static PM_HD void run(cx<T>* a) { ce_dft_odd<T, 3>(a); } };
template <typename T> struct CeDft<T, 5> { static PM_HD void run(cx<T>* a) { ce_dft_odd<T, 5>(a); } };
template <typename T> struct CeDft<T, 7> { static PM_HD void run(cx<T>* a) { ce_dft_odd<T, 7>(a); } };
constexpr int ce_inv_mod(int a, int m) {
    for (int x = 1; x < m; ++x)
        if ((a * x) % m == 1) return x;
    return 1;
}
// R = R1 R2 coprime: n = (R2 n1 + R1 n2) mod R, k = (E1 k1 + E2 k2) mod R with E1 = 1 (mod R1), 0 (mod R2) -- W_R^{nk} = W_R1^{n1 k1} W_R2^{n2 k2}
template <typename T, int R1, int R2>
PM_HD void ce_dft_pfa(cx<T>* a) {
    constexpr int R = R1 * R2;
    constexpr int E1 = R2 * ce_inv_mod(R2 % R1, R1), E2 = R1 * ce_inv_mod(R1 % R2, R2);
    cx<T> t[R1 > R2 ? R1 : R2];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) {
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) t[n1] = a[(R2 * n1 + R1 * n2) % R];
        CeDft<T, R1>::run(t);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) a[(R2 * k1 + R1 * n2) % R] = t[k1];
    }
    cx<T> o[R];
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) t[n2] = a[(R2 * k1 + R1 * n2) % R];
        CeDft<T, R2>::run(t);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) o[(E1 * k1 + E2 * k2) % R] = t[k2];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) a[k] = o[k];
}
template <typename T> struct CeDft<T, 6> { static PM_HD void run(cx<T>* a) { ce_dft_pfa<T, 2, 3>(a); } };
template <typename T> struct CeDft<T, 10> { static PM_HD void run(cx<T>* a) { ce_dft_pfa<T, 2, 5>(a); } };
template <typename T> struct CeDft<T, 12> { static PM_HD void run(cx<T>* a) { ce_dft_pfa<T, 4, 3>(a); } };
template <typename T> struct CeDft<T, 15> { static PM_HD void run(cx<T>* a) { ce_dft_pfa<T, 3, 5>(a); } };
template <typename T> struct CeDft<T, 20> { static PM_HD void run(cx<T>* a) { ce_dft_pfa<T, 4, 5>(a); } };
template <typename T> struct CeDft<T, 24> { static PM_HD void run(cx<T>* a) { ce_dft_pfa<T, 8, 3>(a); } };
template <typename T> struct CeDft<T, 30> { static PM_HD void run(cx<T>* a) { ce_dft_pfa<T, 5, 6>(a); } };
template <typename T> struct CeDft<T, 40> { static PM_HD void run(cx<T>* a) { ce_dft_pfa<T, 8, 5>(a); } };

// The twiddles w^k, k < R <= 48, of a thread's butterflies in one stage: w, w^4 and w^16 from the table, the others as at most three
// products (fft_mixed.h MixTw, extended)
template <typename T, int R>
struct CeTw {
    cx<T> lo[4], hi[12];
    PM_HD CeTw(const cx<T>* __restrict__ tw, uint32_t idx) {
        static_assert(R <= 48, "twiddle powers up to w^47");
        lo[0] = hi[0] = cx<T>{T(1), T(0)};
        lo[1] = mix_ld(tw + idx);
        hi[1] = R > 4 ? mix_ld(tw + 4u * idx) : lo[1];
        hi[4] = R > 16 ? mix_ld(tw + 16u * idx) : lo[1];
        if (R > 2) lo[2] = cmul(lo[1], lo[1]);
        if (R > 3) lo[3] = cmul(lo[2], lo[1]);
        if (R > 8) hi[2] = cmul(hi[1], hi[1]);
        if (R > 12) hi[3] = cmul(hi[2], hi[1]);
        if (R > 20) hi[5] = cmul(hi[4], hi[1]);
        if (R > 24) hi[6] = cmul(hi[4], hi[2]);
        if (R > 28) hi[7] = cmul(hi[4], hi[3]);
        if (R > 32) hi[8] = cmul(hi[4], hi[4]);
        if (R > 36) hi[9] = cmul(hi[8], hi[1]);
        if (R > 40) hi[10] = cmul(hi[8], hi[2]);
        if (R > 44) hi[11] = cmul(hi[8], hi[3]);
    }
    PM_HD cx<T> operator()(int k) const {     // k is a constant after unrolling
        if (k < 4) return lo[k];
        if ((k & 3) == 0) return hi[k >> 2];
        return cmul(hi[k >> 2], lo[k & 3]);
    }
};

// ---------------------------------------------------------------------------
// shape of a workgroup: SEQS sequences, rows ([sequence][point]: lanes run along one sequence) or columns (lanes across the SEQS
// adjacent columns first); COMP 1: complex exchange, 2: real parts then imaginary parts through half the LDS.  PAD_s: slots added to the
// row stride of the exchange INTO stage s.
// ---------------------------------------------------------------------------
template <typename T_, typename PL_, int SEQS_, bool COL_, int COMP_, int PAD1_ = 0, int PAD2_ = 0, int PAD3_ = 0, int WPE_ = 0, int ABL_ = 0>
struct CeCfg {
    static constexpr int ABL = ABL_;     // timing studies only (results wrong): 1 no global loads, 2 no stores, 4 no butterflies, 8 no exchanges
    static constexpr int WPE = WPE_;     // waves per SIMD the kernels are compiled for (0: fft_ce_kernels.h ce_waves_per_eu)
    using T = T_;
    using PL = PL_;
    static constexpr int SEQS = SEQS_, COMP = COMP_, NT = SEQS_ * PL_::TS, P = PL_::P;
    static constexpr bool COL = COL_;
    static constexpr int sb(int s) { return NT + (s == 1 ? PAD1_ : (s == 2 ? PAD2_ : PAD3_)); }
    static constexpr int lds_elems() {
        int v = 0;
        for (int s = 1; s < PL_::S; ++s) v = sb(s) > v ? sb(s) : v;
        return v * P;
    }
    static constexpr size_t LDS_BYTES = size_t(lds_elems()) * sizeof(T_) * (COMP_ == 1 ? 2 : 1);
};
template <typename C> struct CeLds { using type = cx<typename C::T>; };
template <typename T, typename PL, int SEQS, bool COL, int P1, int P2, int P3, int W, int A>
struct CeLds<CeCfg<T, PL, SEQS, COL, 2, P1, P2, P3, W, A>> { using type = T; };

struct CePos {
    int tid, t, sl;     // thread of the workgroup, thread of its sequence, sequence slot
};
template <typename C>
PM_HD CePos ce_pos(int tid) {
    return C::COL ? CePos{tid, tid / C::SEQS, tid % C::SEQS} : CePos{tid, tid % C::PL::TS, tid / C::PL::TS};
}

// stage s on the registers of a thread
template <typename C, int s>
PM_HD void ce_stage(cx<typename C::T> (&v)[C::P], int t, const cx<typename C::T>* __restrict__ tw) {
    using T = typename C::T;
    using PL = typename C::PL;
    constexpr int R = PL::radix(s), Q = PL::q(s);
    constexpr bool last = s + 1 == PL::S;
    if constexpr (last) {
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            cx<T> a[R];
#pragma unroll
            for (int k = 0; k < R; ++k) a[k] = v[k * Q + u];
            CeDft<T, R>::run(a);
#pragma unroll
            for (int k = 0; k < R; ++k) v[k * Q + u] = a[k];
        }
    } else {
        const CeTw<T, R> w(tw, uint32_t(t % PL::m(s)) * uint32_t(PL::tw_step(s)));
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            cx<T> a[R];
#pragma unroll
            for (int k = 0; k < R; ++k) a[k] = v[k * Q + u];
            CeDft<T, R>::run(a);
            v[u] = a[0];
#pragma unroll
            for (int k = 1; k < R; ++k) v[k * Q + u] = cmul(a[k], w(k));
        }
    }
}

// exchange into stage s (1 <= s < S): every register to row r of the fabric at the thread's own slot ...
template <typename C, int s, typename LT>
PM_HD void ce_exch_write(const cx<typename C::T> (&v)[C::P], int comp, CePos pos, LT* lds) {
#pragma unroll
    for (int r = 0; r < C::P; ++r) {
        const int a = r * C::sb(s) + pos.tid;
        if constexpr (C::COMP == 1)
            lds[a] = v[r];
        else
            lds[a] = comp == 0 ? v[r].x : v[r].y;
    }
}
// ... and the butterflies of stage s gathered from it
template <typename C, int s, typename LT>
PM_HD void ce_exch_read(cx<typename C::T> (&v)[C::P], int comp, CePos pos, const LT* lds) {
    using PL = typename C::PL;
    constexpr int R = PL::radix(s), Q = PL::q(s), M = PL::m(s), MP = PL::m(s - 1), GP = PL::g(s - 1);
    const int g = pos.t / M, i2 = pos.t % M, gw = g % GP, r1 = g / GP;
    const int tw0 = gw * MP + i2;
    const int base = r1 * C::sb(s) + (C::COL ? tw0 * C::SEQS + pos.sl : pos.sl * PL::TS + tw0);
#pragma unroll
    for (int u = 0; u < Q; ++u) {
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int a = base + R * u * C::sb(s) + M * m * (C::COL ? C::SEQS : 1);
            if constexpr (C::COMP == 1) {
                v[m * Q + u] = lds[a];
            } else {
                if (comp == 0)
                    v[m * Q + u].x = lds[a];
                else
                    v[m * Q + u].y = lds[a];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// global side.  Rows: sequence `row` at src + row pitch, element q at [q].  Columns: column c at src + c, element q at [q pitch].
// The view of the transform axis (window / rotation) is the library's AxisMap; WIN = false: the view keeps every element.
// ---------------------------------------------------------------------------
template <typename T>
struct CeIn {
    const cx<T>* src;
    int64_t pitch;
    AxisMap ax;
    int nseq;
    T ysign;        // -1: conjugated input
    int64_t bstride;      // a stack of fields (grid.y): elements between their inputs
};
// pupil synthesis in the row loads (DirectIn::synth, fft_io.h synth_value): the element is amp exp(2 pi i k2 opd) from packed (amplitude, OPD)
// pairs (kind 3) or from the OPD map and a separate amplitude array (kind 2) -- Wavefront.from_amp_and_phase(...).focus() on a composite
// grid never writes the complex pupil (prysm/propagation/wavefront.py:58-79)
struct CeSynth {
    int kind;
    double k2;
    const void* amp;
    int amp_kind;
    int64_t amp_ld;
};
template <typename T>
struct CeRowOut {
    cx<T>* dst;
    int64_t ld;
    // mapped: the 1-D API's view of the bins (window / rotation, scale, conjugation -- fft_io.h RowStoreNat); else natural rows
    int mapped;
    AxisMap ax;
    T sr, si;
    int64_t bstride;      // ... between the outputs of the fields of a stack
};
// multiplier of the middle pass (fft_io.h MidMul): full H[k ld + c] or separable hy[k] hx[c], optionally conjugated
template <typename T>
struct CeMul {
    int kind, conj;
    const cx<T>* mul;
    const cx<T>* mul_x;
    int64_t ld;
};
// the ColStoreNat view of the 2-D transform without crop or multiplier: every bin kept, rotations on both axes, scale, conjugation, and the
// library's epilogues (fft_io.h: complex, |.|^2 into a real array, weight |.|^2 added to a real array)
template <typename T>
struct CeColOut {
    void* dst;      // cx<T>*, or T* under an epilogue
    int64_t ld;
    int ny, sy, nx, sx;
    T sr, si;
    int epilogue;
    T weight;
    int64_t bstride;      // output elements (complex, or real under an epilogue) between the fields of a stack
};

// Addresses: a base that is uniform in the workgroup (scalar registers; the per-element constants of the plan fold into it) plus ONE
// 32-bit byte offset per lane -- the saddr + voffset form of the global instructions, no 64-bit vector arithmetic per element.  The
// launchers check that the arrays stay below 4 GiB (ce_fits32).  seq0: the first sequence of the workgroup, sl: the lane's sequence slot
// (clamped by the caller to one that exists).
template <typename T>
PM_HD const cx<T>* ce_at(const cx<T>* ubase, int64_t uelems, uint32_t vbytes) {
    return reinterpret_cast<const cx<T>*>(reinterpret_cast<const char*>(ubase + uelems) + vbytes);
}
template <typename T>
PM_HD cx<T>* ce_at(cx<T>* ubase, int64_t uelems, uint32_t vbytes) {
    return reinterpret_cast<cx<T>*>(reinterpret_cast<char*>(ubase + uelems) + vbytes);
}
// A rotated axis: logical index t + TS m sits at position p = q0 + TS m - (wrapped ? N : 0), q0 = (t + shift) mod N.  With `unit` elements
// between positions: address = [base + (TS m - N) unit] + [q0 unit + (wrapped ? 0 : N unit)] -- uniform part, lane part (never negative).
struct CeRot {
    int q0;
    uint32_t v0, vn;      // bytes: lane offset of position q0 (plus what the caller adds), N unit
};
template <int N>
PM_HD int ce_rot0(int t, int shift) {
    const int q = t + shift;
    return q >= N ? q - N : q;
}

// v[m] = x[t + TS m] of sequence seq0 + sl
template <typename C, bool WIN>
PM_HD void ce_load(cx<typename C::T> (&v)[C::P], const CeIn<typename C::T>& in, int seq0, int sl, int t) {
    using T = typename C::T;
    using PL = typename C::PL;
    constexpr uint32_t ES = sizeof(cx<T>);
    const int q0 = ce_rot0<PL::N>(t, in.ax.shift);
    if (WIN) {      // a window of the axis (zero padding, Q > 1): positions outside it are zero and their loads are not issued
        const int64_t unit = C::COL ? in.pitch : 1;
        const cx<T>* ubase = C::COL ? in.src + seq0 : in.src + int64_t(seq0) * in.pitch;
        const uint32_t vn = uint32_t(PL::N) * uint32_t(unit) * ES;
        const uint32_t v0 = (uint32_t(q0) * uint32_t(unit) + uint32_t(sl) * uint32_t(C::COL ? 1 : in.pitch)) * ES;
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const bool wrapped = q0 >= PL::N - PL::TS * m;
            const int q = q0 + PL::TS * m - (wrapped ? PL::N : 0) - in.ax.off;
            v[m] = cx<T>{T(0), T(0)};
            if (unsigned(q) < unsigned(in.ax.len)) v[m] = mix_ld(ce_at(ubase, int64_t(PL::TS * m - PL::N - in.ax.off) * unit, v0 + (wrapped ? 0u : vn)));
        }
    } else {
        const int64_t unit = C::COL ? in.pitch : 1;
        const cx<T>* ubase = C::COL ? in.src + seq0 : in.src + int64_t(seq0) * in.pitch;
        const uint32_t vn = uint32_t(PL::N) * uint32_t(unit) * ES;
        const uint32_t v0 = (uint32_t(q0) * uint32_t(unit) + uint32_t(sl) * uint32_t(C::COL ? 1 : in.pitch)) * ES;
#pragma unroll
        for (int m = 0; m < C::P; ++m) {
            const bool wrapped = q0 >= PL::N - PL::TS * m;
            v[m] = mix_ld(ce_at(ubase, int64_t(PL::TS * m - PL::N) * unit, v0 + (wrapped ? 0u : vn)));
        }
    }
#pragma unroll
    for (int m = 0; m < C::P; ++m) v[m].y *= in.ysign;
}
// ... synthesised: the raw pairs first (all loads in flight), then the sine / cosine
template <typename C, int SYN>
PM_HD void ce_load_synth(cx<typename C::T> (&v)[C::P], const CeIn<typename C::T>& in, const CeSynth& sy, int seq0, int sl, int t) {
    using T = typename C::T;
    using PL = typename C::PL;
    const int64_t row = seq0 + sl;
    const int q0 = ce_rot0<PL::N>(t, in.ax.shift);
    const cx<T>* pk = in.src + row * in.pitch;                                        // SYN 3: packed pairs, pitch in pairs
    const T* od = reinterpret_cast<const T*>(in.src) + row * in.pitch;                // SYN 2: OPD map, pitch in real elements
    // positions outside the stored window: amplitude 0 (their loads are not issued)
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        int p = q0 + PL::TS * m;
        p = p >= PL::N ? p - PL::N : p;
        const int q = p - in.ax.off;
        v[m] = cx<T>{T(0), T(0)};
        if (unsigned(q) < unsigned(in.ax.len)) {
            if (SYN == 3) {
                v[m] = mix_ld(pk + q);
            } else {
                v[m].y = od[q];
                v[m].x = synth_amp<T>(sy.amp, sy.amp_kind, row * sy.amp_ld + q);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < C::P; ++m) v[m] = synth_value<T>(v[m].y, v[m].x, sy.k2);
}
template <typename C>
PM_HD void ce_store_row(const cx<typename C::T> (&v)[C::P], const CeRowOut<typename C::T>& out, int row0, int sl, int t) {
    using T = typename C::T;
    using PL = typename C::PL;
    constexpr uint32_t ES = sizeof(cx<T>);
    cx<T>* ubase = out.dst + int64_t(row0) * out.ld;
    const uint32_t vrow = uint32_t(sl) * uint32_t(out.ld) * ES;
    if (out.mapped) {
        const int q0 = ce_rot0<PL::N>(t, out.ax.shift);
        const uint32_t v0 = vrow + uint32_t(q0) * ES, vn = uint32_t(PL::N) * ES;
#pragma unroll
        for (int r = 0; r < C::P; ++r) {
            const bool wrapped = q0 >= PL::N - PL::TS * r;
            const int q = q0 + PL::TS * r - (wrapped ? PL::N : 0) - out.ax.off;
            if (unsigned(q) < unsigned(out.ax.len))
                mix_st(ce_at(ubase, int64_t(PL::TS * r - PL::N) - out.ax.off, v0 + (wrapped ? 0u : vn)), cx<T>{v[r].x * out.sr, v[r].y * out.si});
        }
        return;
    }
    const uint32_t v0 = vrow + uint32_t(t) * ES;
#pragma unroll
    for (int r = 0; r < C::P; ++r) mix_st(ce_at(ubase, PL::TS * r, v0), v[r]);
}
// column c0 + sl of the output view: rows rotated by sy, columns by sx
template <typename C>
PM_HD void ce_store_col(const cx<typename C::T> (&v)[C::P], const CeColOut<typename C::T>& out, int c0, int sl, int t) {
    using T = typename C::T;
    using PL = typename C::PL;
    int qx = c0 + sl + out.sx;
    qx = qx >= out.nx ? qx - out.nx : qx;
    const int k0 = ce_rot0<PL::N>(t, out.sy);
    if (out.epilogue == 0) {
        constexpr uint32_t ES = sizeof(cx<T>);
        cx<T>* ubase = reinterpret_cast<cx<T>*>(out.dst);
        const uint32_t vn = uint32_t(PL::N) * uint32_t(out.ld) * ES, v0 = (uint32_t(k0) * uint32_t(out.ld) + uint32_t(qx)) * ES;
#pragma unroll
        for (int r = 0; r < C::P; ++r) {
            const bool wrapped = k0 >= PL::N - PL::TS * r;
            mix_st(ce_at(ubase, int64_t(PL::TS * r - PL::N) * out.ld, v0 + (wrapped ? 0u : vn)), cx<T>{v[r].x * out.sr, v[r].y * out.si});
        }
    } else {
        constexpr uint32_t ES = sizeof(T);
        T* ubase = reinterpret_cast<T*>(out.dst);
        const uint32_t vn = uint32_t(PL::N) * uint32_t(out.ld) * ES, v0 = (uint32_t(k0) * uint32_t(out.ld) + uint32_t(qx)) * ES;
        const T s2 = out.sr * out.sr * (out.epilogue == 2 ? out.weight : T(1));
#pragma unroll
        for (int r = 0; r < C::P; ++r) {
            const bool wrapped = k0 >= PL::N - PL::TS * r;
            T* o = reinterpret_cast<T*>(reinterpret_cast<char*>(ubase + int64_t(PL::TS * r - PL::N) * out.ld) + (v0 + (wrapped ? 0u : vn)));
            const T i2 = (v[r].x * v[r].x + v[r].y * v[r].y) * s2;
            *o = out.epilogue == 2 ? *o + i2 : i2;
        }
    }
}
// middle pass: v[r] = conj(v[r] h(k, col)), k = t + TS r -- the spectrum times the multiplier, conjugated for the inverse that follows
// (KIND is a template argument: with both forms behind a run-time branch the register allocation of the tile spilled 50 .. 100 registers)
template <typename C, int KIND>
PM_HD void ce_mul_col(cx<typename C::T> (&v)[C::P], const CeMul<typename C::T>& mm, int c0, int sl, int t) {
    using T = typename C::T;
    using PL = typename C::PL;
    constexpr uint32_t ES = sizeof(cx<T>);
    const T sg = mm.conj ? T(-1) : T(1);      // a conjugated multiplier by the sign of its imaginary part: no second product to select from
    if constexpr (KIND == 1) {       // MUL_FULL (the compiler hoists as many of the loads as the registers allow; batching them by hand spilled)
        const cx<T>* ubase = mm.mul + c0;
        const uint32_t v0 = (uint32_t(t) * uint32_t(mm.ld) + uint32_t(sl)) * ES;
#pragma unroll
        for (int r = 0; r < C::P; ++r) {
            cx<T> h = mix_ld(ce_at(ubase, int64_t(PL::TS * r) * mm.ld, v0));
            h.y *= sg;
            const cx<T> x = cmul(v[r], h);
            v[r] = cx<T>{x.x, -x.y};
        }
    } else {
        cx<T> hx = mix_ld(mm.mul_x + c0 + sl);
        const uint32_t v0 = uint32_t(t) * ES;
#pragma unroll
        for (int r = 0; r < C::P; ++r) {
            cx<T> h = cmul(mix_ld(ce_at(mm.mul, PL::TS * r, v0)), hx);
            h.y *= sg;
            const cx<T> x = cmul(v[r], h);
            v[r] = cx<T>{x.x, -x.y};
        }
    }
}
// ... and its store: conj(v) to natural rows of the intermediate, dst[k pitch + col]
template <typename C>
PM_HD void ce_store_mid(const cx<typename C::T> (&v)[C::P], cx<typename C::T>* dst, int64_t pitch, int c0, int sl, int t) {
    using T = typename C::T;
    constexpr uint32_t ES = sizeof(cx<T>);
    cx<T>* ubase = dst + c0;
    const uint32_t v0 = (uint32_t(t) * uint32_t(pitch) + uint32_t(sl)) * ES;
#pragma unroll
    for (int r = 0; r < C::P; ++r) mix_st(ce_at(ubase, int64_t(C::PL::TS * r) * pitch, v0), cx<T>{v[r].x, -v[r].y});
}

}  // namespace pm
