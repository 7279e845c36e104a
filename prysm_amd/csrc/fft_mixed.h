// Composite lengths on their own factors: n = r0 r1 ... with every factor <= 20 (all lengths whose primes are <= 19: 1000, 3000,
// 2592, 1001 ...) as ONE kernel per axis, in place of Bluestein's convolution of >= 2n - 1 points per axis (>= 4x the area in 2-D).
// The reference reaches these lengths through scipy.fft / pocketfft, which factors them the same way (prysm/propagation/fft.py:24,
// prysm/fttools.py:23-31, prysm/propagation/angular_spectrum.py:35-42).
//
// Algorithm: decimation in frequency, IN PLACE in LDS (a butterfly writes the slots it read, so one buffer serves any number of
// butterflies per thread), one barrier per stage:
//
//     stage s works on blocks of L_s = n / (r0 .. r_{s-1}) points:  y[j + k L_s/r] = W_{L_s}^{jk} sum_m x[j + m L_s/r] W_r^{mk}
//
// after the last stage slot p = sum_s k_s L_{s+1} holds bin k0 + r0 k1 + r0 r1 k2 + ...  The first stage reads the caller's array
// (window / rotation / real / conjugate: the DirectIn view) and the last one writes the destination, so the data crosses LDS
// (stages - 1) times.  The last stage has no twiddles (the planner puts the largest factor there) and enumerates its butterflies by the
// LOW digits of the bin index, so that adjacent lanes store adjacent bins.
//
// The factor of each stage is a run-time value (one kernel per class of largest factor -- 10, 16, 20 -- serves every length); the small
// DFTs are compile-time (a switch over the factors of the class).
// Row mode: a workgroup holds `seqs` memory rows, lanes run along the row.  Column mode: `seqs` adjacent columns (a power of two),
// lanes run across the columns first -- pieces of seqs elements per row of the array.
//
// Everything below the kernels is __host__ __device__ so that tools/emu_mix.cpp can run the index arithmetic on the CPU.
#pragma once
#include "bluestein.h"
#include "fft_engine.h"
#include "fft_io.h"

namespace pm {

constexpr int kMixMaxStages = 6;
constexpr int kMixMaxN = 8192;
constexpr int kMixMaxRadix = 20;
// factors with a small DFT below.  Factors up to 32 were built and measured (profiles/r03/exp_mix_maxr.log): the kernel class that contains
// them needs 171 VGPRs (complex64) / 256 + spills (complex128) and lost to plans of one more stage in a leaner class at every length tried
// (625 = 25 x 25: 22.2 us against 14.4 as 5 x 5 x 5 x 5; 5000^2 414 against 321; complex128 900^2 42.4 against 22.7)
// (17 and 19 -- round 4 -- by the symmetric half sums like the other odd primes: O(R^2), but lengths such as 1020 = 3 x 4 x 5 x 17 or 1900 = 19 x 10 x 10 leave Bluestein's
// convolution for it; the class of 20 holds them at no extra registers in complex64)
constexpr bool mix_radix_ok(int r) { return r >= 2 && r <= 20; }

// ---------------------------------------------------------------------------
// compile-time roots of unity (octant reduction + Taylor series on [0, pi/4]; ~1 ulp)
// ---------------------------------------------------------------------------
constexpr double mix_sin_small(double x) {
    double x2 = x * x, t = x, s = x;
    for (int i = 1; i < 14; ++i) {
        t *= -x2 / double((2 * i) * (2 * i + 1));
        s += t;
    }
    return s;
}
constexpr double mix_cos_small(double x) {
    double x2 = x * x, t = 1.0, s = 1.0;
    for (int i = 1; i < 14; ++i) {
        t *= -x2 / double((2 * i - 1) * (2 * i));
        s += t;
    }
    return s;
}
// cos / sin of 2 pi num / den
constexpr double mix_root(int num, int den, bool want_sin) {
    const double quarter_pi = 0.78539816339744830961566084581988;
    num %= den;
    if (num < 0) num += den;
    const int oct = (8 * num) / den, rem = 8 * num - oct * den;
    const double r = quarter_pi * double(rem) / double(den), rc = quarter_pi * double(den - rem) / double(den);
    double c = 0, s = 0;
    switch (oct) {
        case 0: c = mix_cos_small(r); s = mix_sin_small(r); break;
        case 1: c = mix_sin_small(rc); s = mix_cos_small(rc); break;
        case 2: c = -mix_sin_small(r); s = mix_cos_small(r); break;
        case 3: c = -mix_cos_small(rc); s = mix_sin_small(rc); break;
        case 4: c = -mix_cos_small(r); s = -mix_sin_small(r); break;
        case 5: c = -mix_sin_small(rc); s = -mix_cos_small(rc); break;
        case 6: c = mix_sin_small(r); s = -mix_cos_small(r); break;
        default: c = mix_cos_small(rc); s = -mix_sin_small(rc); break;
    }
    return want_sin ? s : c;
}
template <int R>
struct MixRootTab {
    double c[R], s[R];
    constexpr MixRootTab() : c{}, s{} {
        for (int k = 0; k < R; ++k) {
            c[k] = mix_root(k, R, false);
            s[k] = mix_root(k, R, true);
        }
    }
};
template <int R>
struct MixRoots {
    static constexpr MixRootTab<R> tab{};
};

// ---------------------------------------------------------------------------
// small DFTs beyond the engine's 2 / 4 / 8 / 16: odd primes by the symmetric half sums, composites by one Cooley-Tukey split with
// compile-time twiddles.  Forward sign, natural order in and out.
// ---------------------------------------------------------------------------
template <typename T, int R>
struct MixDft {
    static PM_HD void run(cx<T>* a) { Dft<T, R>::run(a); }
};

template <typename T, int R>
PM_HD void mix_dft_odd(cx<T>* a) {
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
        cx<T> A = x0, B = {T(0), T(0)};
#pragma unroll
        for (int m = 1; m <= H; ++m) {
            const T c = T(MixRoots<R>::tab.c[(m * k) % R]), s = T(MixRoots<R>::tab.s[(m * k) % R]);
            A.x += p[m].x * c;
            A.y += p[m].y * c;
            B.x += q[m].x * s;
            B.y += q[m].y * s;
        }
        a[k] = {A.x + B.y, A.y - B.x};
        a[R - k] = {A.x - B.y, A.y + B.x};
    }
}
template <typename T> struct MixDft<T, 3> { static PM_HD void run(cx<T>* a) { mix_dft_odd<T, 3>(a); } };
template <typename T> struct MixDft<T, 5> { static PM_HD void run(cx<T>* a) { mix_dft_odd<T, 5>(a); } };
template <typename T> struct MixDft<T, 7> { static PM_HD void run(cx<T>* a) { mix_dft_odd<T, 7>(a); } };
template <typename T> struct MixDft<T, 11> { static PM_HD void run(cx<T>* a) { mix_dft_odd<T, 11>(a); } };
template <typename T> struct MixDft<T, 13> { static PM_HD void run(cx<T>* a) { mix_dft_odd<T, 13>(a); } };
template <typename T> struct MixDft<T, 17> { static PM_HD void run(cx<T>* a) { mix_dft_odd<T, 17>(a); } };
template <typename T> struct MixDft<T, 19> { static PM_HD void run(cx<T>* a) { mix_dft_odd<T, 19>(a); } };

// R = R1 R2, n = R2 n1 + n2, k = k1 + R1 k2
template <typename T, int R1, int R2>
PM_HD void mix_dft_ct(cx<T>* a) {
    constexpr int R = R1 * R2;
    cx<T> t[R1 > R2 ? R1 : R2];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) {
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) t[n1] = a[R2 * n1 + n2];
        MixDft<T, R1>::run(t);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            if (n2 * k1 % R != 0) {
                const cx<T> w = {T(MixRoots<R>::tab.c[(n2 * k1) % R]), T(-MixRoots<R>::tab.s[(n2 * k1) % R])};
                a[R2 * k1 + n2] = cmul(t[k1], w);
            } else {
                a[R2 * k1 + n2] = t[k1];
            }
        }
    }
    cx<T> o[R];
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
#pragma unroll
        for (int n2 = 0; n2 < R2; ++n2) t[n2] = a[R2 * k1 + n2];
        MixDft<T, R2>::run(t);
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) o[k1 + R1 * k2] = t[k2];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) a[k] = o[k];
}
template <typename T> struct MixDft<T, 6> { static PM_HD void run(cx<T>* a) { mix_dft_ct<T, 2, 3>(a); } };
template <typename T> struct MixDft<T, 9> { static PM_HD void run(cx<T>* a) { mix_dft_ct<T, 3, 3>(a); } };
template <typename T> struct MixDft<T, 10> { static PM_HD void run(cx<T>* a) { mix_dft_ct<T, 2, 5>(a); } };
template <typename T> struct MixDft<T, 12> { static PM_HD void run(cx<T>* a) { mix_dft_ct<T, 4, 3>(a); } };
template <typename T> struct MixDft<T, 14> { static PM_HD void run(cx<T>* a) { mix_dft_ct<T, 2, 7>(a); } };
template <typename T> struct MixDft<T, 15> { static PM_HD void run(cx<T>* a) { mix_dft_ct<T, 3, 5>(a); } };
template <typename T> struct MixDft<T, 18> { static PM_HD void run(cx<T>* a) { mix_dft_ct<T, 2, 9>(a); } };
template <typename T> struct MixDft<T, 20> { static PM_HD void run(cx<T>* a) { mix_dft_ct<T, 4, 5>(a); } };

// ---------------------------------------------------------------------------
// plan: factors (largest last), block lengths, exact division by multiply-high
// ---------------------------------------------------------------------------
struct MixPlan {
    int n, nstage;
    int radix[kMixMaxStages];
    int len[kMixMaxStages + 1];     // len[s] = n / (radix[0] .. radix[s-1]); len[nstage] = 1
    uint32_t mg_radix[kMixMaxStages];   // magic of radix[s]
    uint32_t mg_sub[kMixMaxStages];     // magic of len[s + 1]
    uint32_t mg_nb[kMixMaxStages];      // magic of n / radix[s] (butterflies of one sequence in stage s)
    int maxr;                       // largest factor (selects the kernel class)
};
// The kernels read the plan through a pointer to a device-resident copy (one per length, cached beside the twiddles): a plan passed by
// value lands in scratch memory once the kernel is large (the compiler keeps the copy of the kernel argument), and its fields then sit in
// VGPRs.  What varies per launch travels by value:
struct MixShape {
    int seqs, log_seqs;             // sequences per workgroup (column mode: a power of two)
    // LDS slots of a sequence (round 4): point p sits in slot p + pad1 (p / len[2]) + pad0 (p / len[1]) -- a few unused slots after
    // every block of the first two levels.  Unpadded, the last stage reads at the digit-reversed strides len[1], len[2] (300 and 20
    // elements at 3000 = 10 x 15 x 20: only 8 distinct bank pairs for 32 lanes, a 4-way conflict on every read) and the middle stages
    // cross short blocks inside a lane group: SQ_LDS_BANK_CONFLICT was 32 % (columns) to 56 % (rows) of the LDS cycles
    // (profiles/r03/exp_mix_pmc.txt).  s0 / s1: slot strides of the first two digits, npad: slots per sequence.  The host picks the
    // pads per (plan, precision, mode, sequences per workgroup) with a model of the LDS banks (fft_mixed.hip mix_pick_pads).
    int pad0, pad1, s0, s1, npad;
    // start-up stagger (knob mix_stagger): the workgroups of the first wave of the launch (index < first_round) wait 0 .. 7 x stagger x
    // 512 cycles by a hash of their index, so that the load / butterfly / store phases of the workgroups of a CU -- which otherwise
    // start together and stay in step -- overlap
    int stagger, first_round;
    int pers_tiles;     // > 0: the persistent column kernel (fft_mixed_kernels.h mix_cols_pers_kernel) over this many tiles
};
#define PM_MIX_ABLATE(sh, bit) false
inline void mix_shape_pads(const MixPlan& p, MixShape& sh, int c0, int c1);

// floor(a / d) for a d < 2^16 as (a * magic) >> 32, magic = floor(2^32 / d) + 1 (exact while a d < 2^32); d == 1: magic 0 = identity
inline uint32_t mix_magic(int d) { return d <= 1 ? 0u : uint32_t((uint64_t(1) << 32) / uint64_t(d)) + 1u; }
PM_HD int mix_div(int a, uint32_t magic) {
    if (magic == 0) return a;
#if defined(__HIP_DEVICE_COMPILE__)
    return int(__umulhi(uint32_t(a), magic));
#else
    return int((uint64_t(uint32_t(a)) * uint64_t(magic)) >> 32);
#endif
}

// The plan of a length: among all factorisations into factors of the table above, the cheapest by a measured cost model -- stages x
// weight of the kernel class the largest factor selects (10: 1, 16: 1.15, 20: `w20` = 1.3 for complex64, 1.4 for complex128: the leaner
// classes keep more waves per SIMD, and e.g. complex128 3000 = 5 x 6 x 10 x 10 in the class of 10 runs in 189 us against 211 for
// 10 x 15 x 20, while complex64 2000 = 10 x 10 x 20 keeps its three stages: 39.7 us against 44.0 -- profiles/r03/exp_mix_maxr*.log); above 4096 points (one workgroup per CU either
// way: the LDS holds few sequences) the fewest stages win (6000 = 15 x 20 x 20: 516 us against 553 with four stages).  Ties go to the
// smaller largest factor, then to the larger smallest one (fewer butterflies to index).  `maxr` caps the factors (tuning knob mix_maxr).  Returns false when n has a prime factor above 19.
constexpr int mix_class_of(int maxr) { return maxr <= 10 ? 10 : (maxr <= 16 ? 16 : 20); }
inline bool mix_factor(int n, int* radix, int* nstage, int maxr = kMixMaxRadix, double w20 = 1.4, double w16 = 1.15) {
    if (n < 2 || n > kMixMaxN) return false;
    int best[kMixMaxStages], cur[kMixMaxStages], bestn = 0, bestmax = 0, bestmin = 0;
    double bestcost = 1e30;
    const bool big = n > 4096;
    // depth-first over non-increasing factors
    struct Rec {
        static void go(int rem, int maxf, int depth, int* cur, int* best, int& bestn, int& bestmax, int& bestmin, double& bestcost, bool big, double w20, double w16) {
            if (rem == 1) {
                const int cls = mix_class_of(cur[0]);
                const double w = big ? 1.0 : (cls == 10 ? 1.0 : (cls == 16 ? w16 : w20));
                const double cost = depth * w;
                const bool tie = cost < bestcost + 1e-9;
                if (cost < bestcost - 1e-9 || (tie && cur[0] < bestmax) || (tie && cur[0] == bestmax && cur[depth - 1] > bestmin)) {
                    bestcost = cost;
                    bestn = depth;
                    bestmax = cur[0];
                    bestmin = cur[depth - 1];
                    for (int i = 0; i < depth; ++i) best[i] = cur[i];
                }
                return;
            }
            if (depth >= kMixMaxStages) return;
            for (int f = maxf; f >= 2; --f) {
                if (rem % f || !mix_radix_ok(f)) continue;
                cur[depth] = f;
                go(rem / f, f, depth + 1, cur, best, bestn, bestmax, bestmin, bestcost, big, w20, w16);
            }
        }
    };
    Rec::go(n, maxr < kMixMaxRadix ? maxr : kMixMaxRadix, 0, cur, best, bestn, bestmax, bestmin, bestcost, big, w20, w16);
    if (bestn == 0) return false;
    // ascending: the largest factor runs last, where the twiddles are all one
    for (int i = 0; i < bestn; ++i) radix[i] = best[bestn - 1 - i];
    *nstage = bestn;
    return true;
}

inline void mix_fill_plan(int n, const int* radix, int nstage, MixPlan& p) {
    p = MixPlan{};
    p.nstage = nstage;
    for (int s = 0; s < nstage; ++s) p.radix[s] = radix[s];
    p.n = n;
    p.len[0] = n;
    for (int s = 0; s < p.nstage; ++s) {
        p.len[s + 1] = p.len[s] / p.radix[s];
        p.mg_radix[s] = mix_magic(p.radix[s]);
        p.mg_sub[s] = mix_magic(p.len[s + 1]);
        p.mg_nb[s] = mix_magic(n / p.radix[s]);
    }
    p.maxr = radix[nstage - 1];
}
inline bool mix_make_plan(int n, MixPlan& p) {
    int radix[kMixMaxStages], nstage = 0;
    if (!mix_factor(n, radix, &nstage) || nstage < 2) return false;
    mix_fill_plan(n, radix, nstage, p);
    return true;
}

inline void mix_shape_pads(const MixPlan& p, MixShape& sh, int c0, int c1) {
    const int L1 = p.len[1], L2 = p.nstage >= 3 ? p.len[2] : 1;
    if (p.nstage < 3) c1 = 0;
    sh.pad0 = c0;
    sh.pad1 = c1;
    sh.s1 = L2 + c1;
    sh.s0 = L1 + c1 * (L1 / L2) + c0;
    sh.npad = p.radix[0] * sh.s0;
}

// ---------------------------------------------------------------------------
// stages.  `tid` / `nt`: this thread and the threads of the workgroup; sl = sequence slot of the workgroup
// ---------------------------------------------------------------------------
// 24-bit multiply (full rate; the 32-bit one is quarter rate): both factors below 2^24, the low 32 bits of the product
PM_HD uint32_t mix_mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}
// one complex element as ONE 8 / 16 byte access (cx<T> alone only promises the alignment of T)
template <typename T>
PM_HD cx<T> mix_ld(const cx<T>* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef T V __attribute__((ext_vector_type(2)));
    const V w = *reinterpret_cast<const V*>(p);
    return {w[0], w[1]};
#else
    return *p;
#endif
}
template <typename T>
PM_HD void mix_st(cx<T>* p, cx<T> v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef T V __attribute__((ext_vector_type(2)));
    V w;
    w[0] = v.x;
    w[1] = v.y;
    *reinterpret_cast<V*>(p) = w;
#else
    *p = v;
#endif
}

// LDS slot of point i of sequence slot sl, and the slot distance of `d` points (rows: [sl][i]; columns: [i][sl]).  No padding: lanes run
// along i (or along sl first) with unit stride in every stage but the last, whose reads follow the digit-reversed order
template <bool COL>
PM_HD int mix_addr(int, MixShape sh, int sl, int slot) {
    return COL ? ((slot << sh.log_seqs) + sl) : int(mix_mul24(uint32_t(sl), uint32_t(sh.npad)) + uint32_t(slot));
}
// slot of point j < len[1] (first stage: the digit-0 term k s0 is added by the caller)
PM_HD int mix_slot_low(const MixPlan& p, MixShape sh, int j) {
    return sh.pad1 ? j + int(mix_mul24(uint32_t(sh.pad1), uint32_t(mix_div(j, p.mg_sub[1])))) : j;
}
// slot of the first point blk L + j (j < sub) of a butterfly of stage s >= 1, and the slot distance of its points
PM_HD int mix_slot_mid(const MixPlan& p, MixShape sh, int s, int blk, int j, int L) {
    if (s == 1) return int(mix_mul24(uint32_t(blk), uint32_t(sh.s0))) + j;        // blk is digit 0, the points are a digit-1 apart
    const int b0 = int(mix_mul24(uint32_t(blk), uint32_t(L))) + j;                // all points inside one block of len[2]
    int r = b0;
    if (sh.pad1) r += int(mix_mul24(uint32_t(sh.pad1), uint32_t(mix_div(b0, p.mg_sub[1]))));
    if (sh.pad0) r += int(mix_mul24(uint32_t(sh.pad0), uint32_t(mix_div(b0, p.mg_sub[0]))));
    return r;
}
template <bool COL>
PM_HD int mix_step(MixShape sh, int d) {
    return COL ? (d << sh.log_seqs) : d;
}
template <bool COL>
PM_HD void mix_split(MixShape sh, int b, uint32_t mg_nb, int nb, int& sl, int& j) {
    if (COL) {
        sl = b & (sh.seqs - 1);
        j = b >> sh.log_seqs;
    } else {
        sl = mix_div(b, mg_nb);
        j = b - int(mix_mul24(uint32_t(sl), uint32_t(nb)));
    }
}

// The twiddles w^k, k = 1 .. R-1, of one butterfly (w = tw[idx], a root of unity of the transform length): one table load each for
// w, w^4 and w^16 and at most three complex products for the rest (w^2 = w w, w^3 = w^2 w, w^8 = w^4 w^4, w^12 = w^8 w^4,
// w^(4b+a) = w^(4b) w^a, w^(16+a) = w^16 w^a) instead of R - 1 loads.  The loads, not the butterflies, were what the stages waited on:
// a wave's twiddle load touches up to 32 different 128 B lines (lane stride 8 k tstep bytes), one tag lookup each, and a stage issued
// R - 1 of them per butterfly -- with them switched off a 3000-point row kernel ran 15 us of 62 faster, without the butterflies only 6
// (profiles/r04/exp_mix_ablate.log).  A product costs one rounding of the table's accuracy; three of them stay far inside the parity bar.
template <typename T, int R>
struct MixTw {
    cx<T> lo[4], hi[5];
    // the table loads on their own (a stage that has other loads in flight issues them together), then the products
    struct Raw { cx<T> w1, w4, w16; };
    static PM_HD Raw load(const cx<T>* __restrict__ tw, uint32_t idx) {
        Raw r;
        r.w1 = mix_ld(tw + idx);
        r.w4 = R > 4 ? mix_ld(tw + 4u * idx) : r.w1;
        r.w16 = R > 16 ? mix_ld(tw + 16u * idx) : r.w1;
        return r;
    }
    PM_HD explicit MixTw(const Raw& r) {
        lo[0] = hi[0] = cx<T>{T(1), T(0)};
        lo[1] = r.w1;
        hi[1] = r.w4;
        hi[4] = r.w16;
        if (R > 2) lo[2] = cmul(lo[1], lo[1]);
        if (R > 3) lo[3] = cmul(lo[2], lo[1]);
        if (R > 8) hi[2] = cmul(hi[1], hi[1]);
        if (R > 12) hi[3] = cmul(hi[2], hi[1]);
    }
    PM_HD MixTw(const cx<T>* __restrict__ tw, uint32_t idx) : MixTw(load(tw, idx)) {}
    PM_HD cx<T> operator()(int k) const {     // k is a constant after unrolling
        if (k < 4) return lo[k];
        if ((k & 3) == 0) return hi[k >> 2];
        return k < 16 ? cmul(hi[k >> 2], lo[k & 3]) : cmul(hi[4], lo[k - 16]);
    }
};

// first stage: caller's array -> LDS
// (a Fetch with a finish() member returns the raw load from operator() and does its arithmetic in finish(): the loads of a trip then
// issue back to back)
template <typename F, typename T> PM_HD auto mix_finish(const F& f, cx<T> x, int) -> decltype(f.finish(x)) { return f.finish(x); }
template <typename F, typename T> PM_HD cx<T> mix_finish(const F&, cx<T> x, long) { return x; }
template <typename T>
constexpr int mix_first_unroll(int r) {
    return sizeof(T) == 4 ? (r <= 10 ? 3 : (r <= 16 ? 2 : 1)) : (r <= 10 ? 2 : 1);
}
template <typename T, bool COL, int R, int UCAP, typename Fetch>
PM_HD void mix_first(const MixPlan& p, MixShape sh, int tid, int nt, cx<T>* lds, const cx<T>* __restrict__ tw, Fetch fetch) {
    const int n = p.n, nb = p.len[1], total = sh.seqs * nb;
    const uint32_t mg_nb0 = p.mg_nb[0];
    // U butterflies of a thread per trip, all their loads issued before the first butterfly: a workgroup's trips are serial round trips to
    // memory (3000-point rows: 600 butterflies of 10 on 256 threads, three trips, while the workgroups of a CU run in step and nobody
    // computes), so the small factors take up to three at once -- within 32 complex64 / 20 complex128 elements in registers
    constexpr int U = mix_first_unroll<T>(R) < UCAP ? mix_first_unroll<T>(R) : UCAP;     // UCAP: 1 in the 1024-thread column kernels (128 registers)
#pragma unroll 1
    for (int b0 = tid; b0 < total; b0 += U * nt) {
        cx<T> a[U][R];
        typename MixTw<T, R>::Raw wr[U];
        int sl[U], j[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            sl[u] = j[u] = 0;
            if (b0 + u * nt < total) {
                mix_split<COL>(sh, b0 + u * nt, mg_nb0, nb, sl[u], j[u]);
                wr[u] = MixTw<T, R>::load(tw, PM_MIX_ABLATE(sh, 2) ? 0u : uint32_t(j[u]));
#pragma unroll
                for (int k = 0; k < R; ++k) a[u][k] = PM_MIX_ABLATE(sh, 1) ? cx<T>{T(j[u] + k), T(sl[u])} : fetch(sl[u], j[u] + k * nb);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (b0 + u * nt < total) {
#pragma unroll
                for (int k = 0; k < R; ++k) a[u][k] = mix_finish(fetch, a[u][k], 0);
                if (!PM_MIX_ABLATE(sh, 8)) MixDft<T, R>::run(a[u]);
                const int a0 = mix_addr<COL>(n, sh, sl[u], mix_slot_low(p, sh, j[u])), as = mix_step<COL>(sh, sh.s0);
                mix_st(lds + a0, a[u][0]);
                const MixTw<T, R> w(wr[u]);
#pragma unroll
                for (int k = 1; k < R; ++k) mix_st(lds + a0 + k * as, cmul(a[u][k], w(k)));
            }
        }
    }
}

// first stage on a sequence that is already in LDS at its slots (the persistent column kernel copies the tile in): point j + k nb sits
// in slot slot_low(j) + k s0, which is also where the stage leaves digit k -- in place, like a middle stage
template <typename T, bool COL, int R>
PM_HD void mix_first_lds(const MixPlan& p, MixShape sh, int tid, int nt, cx<T>* lds, const cx<T>* __restrict__ tw) {
    const int n = p.n, nb = p.len[1], total = sh.seqs * nb;
    const uint32_t mg_nb0 = p.mg_nb[0];
#pragma unroll 1
    for (int b = tid; b < total; b += nt) {
        int sl, j;
        mix_split<COL>(sh, b, mg_nb0, nb, sl, j);
        const typename MixTw<T, R>::Raw wr = MixTw<T, R>::load(tw, uint32_t(j));
        const int a0 = mix_addr<COL>(n, sh, sl, mix_slot_low(p, sh, j)), as = mix_step<COL>(sh, sh.s0);
        cx<T> a[R];
#pragma unroll
        for (int k = 0; k < R; ++k) a[k] = mix_ld(lds + a0 + k * as);
        MixDft<T, R>::run(a);
        mix_st(lds + a0, a[0]);
        const MixTw<T, R> w(wr);
#pragma unroll
        for (int k = 1; k < R; ++k) mix_st(lds + a0 + k * as, cmul(a[k], w(k)));
    }
}
// LDS slot of point i of a sequence (any i < n): the copy-in of the persistent column kernel
PM_HD int mix_slot_of(const MixPlan& p, MixShape sh, int i) {
    int r = i;
    if (sh.pad1) r += int(mix_mul24(uint32_t(sh.pad1), uint32_t(mix_div(i, p.mg_sub[1]))));
    if (sh.pad0) r += int(mix_mul24(uint32_t(sh.pad0), uint32_t(mix_div(i, p.mg_sub[0]))));
    return r;
}

// middle stage s: LDS in place
template <typename T, bool COL, int R>
PM_HD void mix_mid(const MixPlan& p, MixShape sh, int s, int tid, int nt, cx<T>* lds, const cx<T>* __restrict__ tw) {
    const int n = p.n, nb = n / R, total = sh.seqs * nb, sub = p.len[s + 1], L = sub * R, tstep = n / L;
    const uint32_t mg_nb = p.mg_nb[s], mg_sub = p.mg_sub[s];
#pragma unroll 1
    for (int b = tid; b < total; b += nt) {
        int sl, ja;
        mix_split<COL>(sh, b, mg_nb, nb, sl, ja);
        const int blk = mix_div(ja, mg_sub), j = ja - int(mix_mul24(uint32_t(blk), uint32_t(sub)));
        const int a0 = mix_addr<COL>(n, sh, sl, mix_slot_mid(p, sh, s, blk, j, L)), as = mix_step<COL>(sh, s == 1 ? sh.s1 : sub);
        const typename MixTw<T, R>::Raw wr = MixTw<T, R>::load(tw, PM_MIX_ABLATE(sh, 2) ? 0u : mix_mul24(uint32_t(j), uint32_t(tstep)));
        cx<T> a[R];
#pragma unroll
        for (int k = 0; k < R; ++k) a[k] = mix_ld(lds + a0 + k * as);
        if (!PM_MIX_ABLATE(sh, 8)) MixDft<T, R>::run(a);
        mix_st(lds + a0, a[0]);
        const MixTw<T, R> w(wr);
#pragma unroll
        for (int k = 1; k < R; ++k) mix_st(lds + a0 + k * as, cmul(a[k], w(k)));
    }
}

// last stage: LDS -> destination; butterfly o (the low digits of the bin) produces bins o + k n / R
template <typename T, bool COL, int R, typename Store>
PM_HD void mix_last(const MixPlan& p, MixShape sh, int tid, int nt, const cx<T>* lds, Store store) {
    const int n = p.n, s = p.nstage - 1, nb = n / R, total = sh.seqs * nb;
    const uint32_t mg_nb = p.mg_nb[s];
    // the digits of the butterfly index: factors, their reciprocals and the slot weights of the stages before the last (uniform values)
    int rdx[kMixMaxStages - 1], wgt[kMixMaxStages - 1];
    uint32_t mgr[kMixMaxStages - 1];
#pragma unroll
    for (int i = 0; i < kMixMaxStages - 1; ++i) {
        rdx[i] = i < s ? p.radix[i] : 1;
        wgt[i] = i < s ? (i == 0 ? sh.s0 : (i == 1 ? sh.s1 : p.len[i + 1])) : 0;      // slot strides of the digits (MixShape)
        mgr[i] = i < s ? p.mg_radix[i] : 0u;
    }
#pragma unroll 1
    for (int b = tid; b < total; b += nt) {
        int sl, o;
        mix_split<COL>(sh, b, mg_nb, nb, sl, o);
        int rem = o, pos = 0;
#pragma unroll
        for (int i = 0; i < kMixMaxStages - 1; ++i) {
            if (i < s) {
                const int q = mix_div(rem, mgr[i]);
                pos += int(mix_mul24(uint32_t(rem) - mix_mul24(uint32_t(q), uint32_t(rdx[i])), uint32_t(wgt[i])));
                rem = q;
            }
        }
        const int a0 = mix_addr<COL>(n, sh, sl, pos), as = mix_step<COL>(sh, 1);
        cx<T> a[R];
#pragma unroll
        for (int k = 0; k < R; ++k) a[k] = mix_ld(lds + a0 + k * as);
        if (!PM_MIX_ABLATE(sh, 8)) MixDft<T, R>::run(a);
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (!PM_MIX_ABLATE(sh, 4) || a[k].x == T(-12345.678)) store(sl, o + k * nb, a[k]);
    }
}


// ---------------------------------------------------------------------------
// The TRANSPOSED stages: the same factorisation run backwards turns the digit-reversed slots the forward stages leave into a forward
// DFT in NATURAL order.  F_n = P S_last .. S_0 (S_s = T_s F_R per block, P the digit reversal) and F_n is symmetric, so
// F_n = S_0^T .. S_last^T P^T with S_s^T = F_R T_s: twiddle first, then the small DFT, on the slots the forward stage used.  The middle
// pass of fft2 -> x H -> ifft2 on composite grids is forward stages, multiplier, transposed stages with the column resident in LDS
// throughout (ifft = conj fft conj: the multiplier step stores conj(x h), the last transposed stage stores the conjugate).
// ---------------------------------------------------------------------------
// transposed middle stage s: LDS in place
template <typename T, bool COL, int R>
PM_HD void mix_mid_t(const MixPlan& p, MixShape sh, int s, int tid, int nt, cx<T>* lds, const cx<T>* __restrict__ tw) {
    const int n = p.n, nb = n / R, total = sh.seqs * nb, sub = p.len[s + 1], L = sub * R, tstep = n / L;
    const uint32_t mg_nb = p.mg_nb[s], mg_sub = p.mg_sub[s];
#pragma unroll 1
    for (int b = tid; b < total; b += nt) {
        int sl, ja;
        mix_split<COL>(sh, b, mg_nb, nb, sl, ja);
        const int blk = mix_div(ja, mg_sub), j = ja - int(mix_mul24(uint32_t(blk), uint32_t(sub)));
        const int a0 = mix_addr<COL>(n, sh, sl, mix_slot_mid(p, sh, s, blk, j, L)), as = mix_step<COL>(sh, s == 1 ? sh.s1 : sub);
        const MixTw<T, R> w(tw, mix_mul24(uint32_t(j), uint32_t(tstep)));
        cx<T> a[R];
        a[0] = mix_ld(lds + a0);
#pragma unroll
        for (int k = 1; k < R; ++k) a[k] = cmul(mix_ld(lds + a0 + k * as), w(k));
        MixDft<T, R>::run(a);
#pragma unroll
        for (int k = 0; k < R; ++k) mix_st(lds + a0 + k * as, a[k]);
    }
}

// last forward stage, multiplier, last stage transposed (no twiddles either way): butterfly o holds bins o + k n / R in registers between
// the two small DFTs -- no LDS round trip, no barrier.  mul(sl, bin, v) returns conj(v h(bin, column of sl)).
template <typename T, bool COL, int R, typename Mul>
PM_HD void mix_last_mul(const MixPlan& p, MixShape sh, int tid, int nt, cx<T>* lds, Mul mul) {
    const int n = p.n, s = p.nstage - 1, nb = n / R, total = sh.seqs * nb;
    const uint32_t mg_nb = p.mg_nb[s];
    int rdx[kMixMaxStages - 1], wgt[kMixMaxStages - 1];
    uint32_t mgr[kMixMaxStages - 1];
#pragma unroll
    for (int i = 0; i < kMixMaxStages - 1; ++i) {
        rdx[i] = i < s ? p.radix[i] : 1;
        wgt[i] = i < s ? (i == 0 ? sh.s0 : (i == 1 ? sh.s1 : p.len[i + 1])) : 0;      // slot strides of the digits (MixShape)
        mgr[i] = i < s ? p.mg_radix[i] : 0u;
    }
#pragma unroll 1
    for (int b = tid; b < total; b += nt) {
        int sl, o;
        mix_split<COL>(sh, b, mg_nb, nb, sl, o);
        int rem = o, pos = 0;
#pragma unroll
        for (int i = 0; i < kMixMaxStages - 1; ++i) {
            if (i < s) {
                const int q = mix_div(rem, mgr[i]);
                pos += int(mix_mul24(uint32_t(rem) - mix_mul24(uint32_t(q), uint32_t(rdx[i])), uint32_t(wgt[i])));
                rem = q;
            }
        }
        const int a0 = mix_addr<COL>(n, sh, sl, pos), as = mix_step<COL>(sh, 1);
        cx<T> a[R];
#pragma unroll
        for (int k = 0; k < R; ++k) a[k] = mix_ld(lds + a0 + k * as);
        MixDft<T, R>::run(a);
#pragma unroll
        for (int k = 0; k < R; ++k) a[k] = mul(sl, o + k * nb, a[k]);
        MixDft<T, R>::run(a);
#pragma unroll
        for (int k = 0; k < R; ++k) mix_st(lds + a0 + k * as, a[k]);
    }
}

// transposed first stage: LDS -> destination, natural order (point j + k n / R of sequence slot sl)
template <typename T, bool COL, int R, typename Store>
PM_HD void mix_first_t(const MixPlan& p, MixShape sh, int tid, int nt, const cx<T>* lds, const cx<T>* __restrict__ tw, Store store) {
    const int n = p.n, nb = p.len[1], total = sh.seqs * nb;
    const uint32_t mg_nb0 = p.mg_nb[0];
#pragma unroll 1
    for (int b = tid; b < total; b += nt) {
        int sl, j;
        mix_split<COL>(sh, b, mg_nb0, nb, sl, j);
        const int a0 = mix_addr<COL>(n, sh, sl, mix_slot_low(p, sh, j)), as = mix_step<COL>(sh, sh.s0);
        const MixTw<T, R> w(tw, uint32_t(j));
        cx<T> a[R];
        a[0] = mix_ld(lds + a0);
#pragma unroll
        for (int k = 1; k < R; ++k) a[k] = cmul(mix_ld(lds + a0 + k * as), w(k));
        MixDft<T, R>::run(a);
#pragma unroll
        for (int k = 0; k < R; ++k) store(sl, j + k * nb, a[k]);
    }
}

// `MAXR` (a template parameter of the caller) bounds the factors a kernel class contains
#define PM_MIX_RADIX_SWITCH(r, CALL) \
    switch (r) { \
        case 2: if constexpr (MAXR >= 2) { constexpr int R = 2; CALL; } break; \
        case 3: if constexpr (MAXR >= 3) { constexpr int R = 3; CALL; } break; \
        case 4: if constexpr (MAXR >= 4) { constexpr int R = 4; CALL; } break; \
        case 5: if constexpr (MAXR >= 5) { constexpr int R = 5; CALL; } break; \
        case 6: if constexpr (MAXR >= 6) { constexpr int R = 6; CALL; } break; \
        case 7: if constexpr (MAXR >= 7) { constexpr int R = 7; CALL; } break; \
        case 8: if constexpr (MAXR >= 8) { constexpr int R = 8; CALL; } break; \
        case 9: if constexpr (MAXR >= 9) { constexpr int R = 9; CALL; } break; \
        case 10: if constexpr (MAXR >= 10) { constexpr int R = 10; CALL; } break; \
        case 11: if constexpr (MAXR >= 11) { constexpr int R = 11; CALL; } break; \
        case 12: if constexpr (MAXR >= 12) { constexpr int R = 12; CALL; } break; \
        case 13: if constexpr (MAXR >= 13) { constexpr int R = 13; CALL; } break; \
        case 14: if constexpr (MAXR >= 14) { constexpr int R = 14; CALL; } break; \
        case 15: if constexpr (MAXR >= 15) { constexpr int R = 15; CALL; } break; \
        case 16: if constexpr (MAXR >= 16) { constexpr int R = 16; CALL; } break; \
        case 17: if constexpr (MAXR >= 20) { constexpr int R = 17; CALL; } break; \
        case 19: if constexpr (MAXR >= 20) { constexpr int R = 19; CALL; } break; \
        case 18: if constexpr (MAXR >= 18) { constexpr int R = 18; CALL; } break; \
        case 20: if constexpr (MAXR >= 20) { constexpr int R = 20; CALL; } break; \
        default: break; \
    }

// output of the row mode: natural rows (the intermediate of a 2-D transform) or the 1-D API's view (window / rotation, scale, conj)
template <typename T>
struct MixRowOut {
    cx<T>* dst;
    int64_t ld;
    AxisMap ax;
    T scale;
    int conj;
    int mapped;
    // FOLD (round 4, experiment builds: it measured SLOWER than the unfolded pair of passes -- 3000^2 complex64 78.9 -> 87.7 us, 4000^2 126 -> 143,
    // complex128 3000^2 175 -> 199, profiles/r04/exp_mix_fold.log: two rows per workgroup cost the row pass more than the half-height tiles give
    // the column pass; the engine's RowStoreFold on composite lengths): the workgroup's two rows are (g, g + fold_h) of an array of 2 fold_h
    // rows, and what is stored is one radix-2 decimation-in-frequency step of the COLUMN transform -- plane 0 row g = y[g] + y[g + H],
    // plane 1 row g = (y[g] - y[g + H]) W_M^g (planes fold_h rows apart); the column pass then runs H-point tiles, half the LDS each.
    // fold_tw[g] = W_M^g; fold_swap: the input rows are rotated by H, so the two rows of the pair trade places (plane 1 changes sign)
    int fold_h, fold_swap;
    const cx<T>* fold_tw;
};

template <typename T>
PM_HD void mix_store_row(const MixRowOut<T>& o, int seq, int k, cx<T> v) {
    if (o.mapped) {
        const int q = o.ax.map(k);
        if (q < 0) return;
        v = cscale(v, o.scale);
        if (o.conj) v.y = -v.y;
        o.dst[int64_t(seq) * o.ld + q] = v;
    } else {
        o.dst[int64_t(seq) * o.ld + k] = v;
    }
}

// the phases of a workgroup's work for thread `tid`: first stage, middle stage `s` (1 .. nstage-2), last stage.  The kernel puts a barrier
// between phases; the emulator runs every thread of a phase before the next one
template <typename T, bool COL, int MAXR, int UCAP = 1, typename Fetch>
PM_HD void mix_run_first(const MixPlan& p, MixShape sh, int tid, int nt, cx<T>* lds, const cx<T>* __restrict__ tw, Fetch fetch) {
    PM_MIX_RADIX_SWITCH(p.radix[0], (mix_first<T, COL, R, UCAP>(p, sh, tid, nt, lds, tw, fetch)))
}
template <typename T, bool COL, int MAXR>
PM_HD void mix_run_first_lds(const MixPlan& p, MixShape sh, int tid, int nt, cx<T>* lds, const cx<T>* __restrict__ tw) {
    PM_MIX_RADIX_SWITCH(p.radix[0], (mix_first_lds<T, COL, R>(p, sh, tid, nt, lds, tw)))
}
template <typename T, bool COL, int MAXR>
PM_HD void mix_run_mid(const MixPlan& p, MixShape sh, int s, int tid, int nt, cx<T>* lds, const cx<T>* __restrict__ tw) {
    PM_MIX_RADIX_SWITCH(p.radix[s], (mix_mid<T, COL, R>(p, sh, s, tid, nt, lds, tw)))
}
template <typename T, bool COL, int MAXR, typename Store>
PM_HD void mix_run_last(const MixPlan& p, MixShape sh, int tid, int nt, const cx<T>* lds, Store store) {
    PM_MIX_RADIX_SWITCH(p.radix[p.nstage - 1], (mix_last<T, COL, R>(p, sh, tid, nt, lds, store)))
}

template <typename T, bool COL, int MAXR>
PM_HD void mix_run_mid_t(const MixPlan& p, MixShape sh, int s, int tid, int nt, cx<T>* lds, const cx<T>* __restrict__ tw) {
    PM_MIX_RADIX_SWITCH(p.radix[s], (mix_mid_t<T, COL, R>(p, sh, s, tid, nt, lds, tw)))
}
template <typename T, bool COL, int MAXR, typename Mul>
PM_HD void mix_run_last_mul(const MixPlan& p, MixShape sh, int tid, int nt, cx<T>* lds, Mul mul) {
    PM_MIX_RADIX_SWITCH(p.radix[p.nstage - 1], (mix_last_mul<T, COL, R>(p, sh, tid, nt, lds, mul)))
}
template <typename T, bool COL, int MAXR, typename Store>
PM_HD void mix_run_first_t(const MixPlan& p, MixShape sh, int tid, int nt, const cx<T>* lds, const cx<T>* __restrict__ tw, Store store) {
    PM_MIX_RADIX_SWITCH(p.radix[0], (mix_first_t<T, COL, R>(p, sh, tid, nt, lds, tw, store)))
}

// Branch-free element of the DirectIn view for the first stage: the load always happens (at element 0 of the workgroup's first sequence
// when the logical index falls outside the stored window or the sequence does not exist) and the value is selected afterwards, so the
// R loads of a butterfly issue back to back.  Offsets are 32-bit from a base that is uniform in the workgroup (the launcher checks that
// they fit: fft_mixed_kernels.h mix_fits).  rows: element (sl, q) at base[sl pitch + q]; columns: at base[q pitch + sl].
// ... and the common case without any of that work: a complex array whose view keeps every element (rotation only) and a workgroup
// whose sequences all exist -- 5 integer instructions per element instead of ~20
template <typename T, bool COL>
struct MixFetchWhole {
    const cx<T>* base;
    uint32_t pitch;
    int n, shift;
    T ysign;
    PM_HD cx<T> operator()(int sl, int i) const {
        int q = i + shift;
        q = q >= n ? q - n : q;
        const uint32_t off = COL ? mix_mul24(uint32_t(q), pitch) + uint32_t(sl) : mix_mul24(uint32_t(sl), pitch) + uint32_t(q);
        return mix_ld(base + off);      // raw: finish() applies the conjugation once every load of the trip is in flight
    }
    PM_HD cx<T> finish(cx<T> x) const { return cx<T>{x.x, x.y * ysign}; }
};

template <typename T, bool COL, bool REAL>
struct MixFetch {
    const void* base;
    uint32_t pitch;
    AxisMap ax;
    T ysign;        // -1: conjugated input
    int nvalid;     // sequences of this workgroup that exist
    PM_HD cx<T> operator()(int sl, int i) const {
        int q = ax.map(i);
        const bool ok = sl < nvalid && q >= 0;
        q = ok ? q : 0;
        sl = ok ? sl : 0;
        const uint32_t off = COL ? mix_mul24(uint32_t(q), pitch) + uint32_t(sl) : mix_mul24(uint32_t(sl), pitch) + uint32_t(q);
        cx<T> x;
        if (REAL)
            x = {reinterpret_cast<const T*>(base)[off], T(0)};
        else
            x = mix_ld(reinterpret_cast<const cx<T>*>(base) + off);
        x.y *= ysign;
        return ok ? x : cx<T>{T(0), T(0)};
    }
    PM_HD cx<T> finish(cx<T> x) const { return x; }
};

// ... and the pupil synthesised while loading (rows only): amp exp(2 pi i k2 opd) from packed (amplitude, OPD) pairs or from the OPD map
// and a separate amplitude array (Wavefront.from_amp_and_phase, prysm/propagation/wavefront.py:58-79) -- the complex pupil of a composite
// grid is never written to memory either
template <typename T, bool PACKED>
struct MixFetchSynth {
    const void* base;       // packed pairs (cx<T>) or the OPD map (T), at the workgroup's first row
    uint32_t pitch;
    AxisMap ax;
    int nvalid;
    double k2;
    const void* amp;        // !PACKED: amplitude array at the workgroup's first row (or null: unit amplitude)
    int amp_kind;
    uint32_t amp_pitch;
    PM_HD cx<T> operator()(int sl, int i) const {
        int q = ax.map(i);
        const bool ok = sl < nvalid && q >= 0;
        q = ok ? q : 0;
        sl = ok ? sl : 0;
        const uint32_t off = mix_mul24(uint32_t(sl), pitch) + uint32_t(q);
        cx<T> x;
        if (PACKED) {
            const cx<T> ao = mix_ld(reinterpret_cast<const cx<T>*>(base) + off);
            x = synth_value<T>(ao.y, ao.x, k2);
        } else {
            const T a = synth_amp<T>(amp, amp_kind, int64_t(mix_mul24(uint32_t(sl), amp_pitch) + uint32_t(q)));
            x = synth_value<T>(reinterpret_cast<const T*>(base)[off], a, k2);
        }
        return ok ? x : cx<T>{T(0), T(0)};
    }
    PM_HD cx<T> finish(cx<T> x) const { return x; }
};

}  // namespace pm
