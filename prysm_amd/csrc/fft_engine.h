// Register-resident Stockham FFT engine for gfx950 (MI355X).
//
// One workgroup transforms SEQ_PER_WG sequences of length N = 2^LOGN.  Every
// thread owns P = min(16, N) points of E sequences and keeps them in VGPRs for
// the whole transform; LDS (160 KiB / CU) is used only as an exchange fabric
// between radix-P stages, one "chunk" (one of the thread's E sequences, or the
// real / imaginary halves when COMP == 2) at a time.  That is what lets one
// workgroup hold 256 KiB of field data (8 columns x 4096 rows of complex64, or
// 4 columns of complex128) on a CU whose LDS is only 160 KiB: the 512 KiB VGPR
// file is the big on-chip store on CDNA4, not the LDS.
//
// Data layout invariant: thread slot t of a sequence holds
//     v[m] = x[t + m * N/P],  m = 0..P-1
// before the first stage and after the last one (natural order in, natural
// order out -- Stockham autosort), so a second transform (e.g. the inverse of
// the angular-spectrum step) can start from the registers of the first.
//
// Stage s (radix R, Ns = P^s):  butterfly j = t + q*N/P (q < P/R) takes
// x[j + k*N/R] = v[k*(P/R) + q], multiplies by W_N^{(j mod Ns) * k * N/(Ns R)},
// applies a DFT-R and scatters to y[(j/Ns) Ns R + (j mod Ns) + k Ns].
//
// Everything below the kernels is __host__ __device__ so that
// tools/emu_fft.cpp can run the exact index arithmetic on the CPU (this
// container has no GPU); it is test scaffolding for the kernel, not a product
// code path.
#pragma once
#include "pm_common.h"

namespace pm {

constexpr int ilog2(int n) { return n <= 1 ? 0 : 1 + ilog2(n >> 1); }
constexpr int ipow(int b, int e) { return e == 0 ? 1 : b * ipow(b, e - 1); }

// LOGPMAX_: log2 of the most points a thread holds per sequence -- 4 (16 points, radix-16 stages: everything the library ships) or 3
// (8 points, radix-8 stages: the lighter-wave engine of round 4's experiment build)
template <typename T_, int LOGN_, int CI_, int E_, int BO_, int COMP_, int LOGPMAX_ = 4>
struct FftCfg {
    using T = T_;
    static constexpr int LOGN = LOGN_;
    static constexpr int N = 1 << LOGN_;
    static constexpr int P = N >= (1 << LOGPMAX_) ? (1 << LOGPMAX_) : N;          // points per thread per sequence
    static constexpr int LOGP = ilog2(P);
    static constexpr int TPS = N / P;                    // threads per sequence
    static constexpr int CI = CI_;                       // sequences interleaved over adjacent lanes
    static constexpr int E = E_;                         // adjacent sequences owned by one thread
    static constexpr int BO = BO_;                       // outer sequence groups per workgroup
    static constexpr int COMP = COMP_;                   // 1: exchange complex, 2: exchange re then im
    static constexpr int NT = CI * TPS * BO;             // threads per workgroup
    static constexpr int SEQ_PER_WG = CI * E * BO;
    static constexpr int NSTAGE = (LOGN + LOGP - 1) / LOGP;
    static constexpr int PADN = N + (N >> 4);            // +1 element every 16: conflict-free stride-16 scatter
    static constexpr int LDS_ELEMS = BO * PADN * CI;     // per exchange chunk
    static constexpr size_t LDS_BYTES =
        NSTAGE > 1 ? size_t(LDS_ELEMS) * sizeof(T_) * (COMP_ == 1 ? 2 : 1) : 0;
    static constexpr int radix(int s) { return (s < LOGN / LOGP) ? P : (1 << (LOGN % LOGP)); }
    static constexpr int ns(int s) { return ipow(P, s); }
};

struct ThreadPos {
    int cl, t, bo;
};
template <typename C>
PM_HD ThreadPos thread_pos(int tid) {
    return {tid % C::CI, (tid / C::CI) % C::TPS, tid / (C::CI * C::TPS)};
}

// ---------------------------------------------------------------------------
// small DFTs, forward (exp(-2 pi i nk/R)), natural order in and out
// ---------------------------------------------------------------------------
template <typename T>
PM_HD void dft2(cx<T>& a, cx<T>& b) {
    cx<T> t = a;
    a = t + b;
    b = t - b;
}

template <typename T>
PM_HD void dft4(cx<T>& a0, cx<T>& a1, cx<T>& a2, cx<T>& a3) {
    cx<T> s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    cx<T> m = mul_mi(d13);  // -i (a1 - a3)
    a0 = s02 + s13;
    a2 = s02 - s13;
    a1 = d02 + m;
    a3 = d02 - m;
}

#if defined(__HIP_DEVICE_COMPILE__) && defined(PM_PACKED_F32)
// complex64 on packed fp32 instructions (pm_common.h): 8 instructions per radix-4 butterfly instead of 16
__device__ __forceinline__ void dft4(cx<float>& a0, cx<float>& a1, cx<float>& a2, cx<float>& a3) {
    const cx<float> s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13;
    a2 = s02 - s13;
    a1 = add_mi(d02, d13);
    a3 = sub_mi(d02, d13);
}
#endif

template <typename T, int R>
struct Dft;

template <typename T>
struct Dft<T, 1> {
    static PM_HD void run(cx<T>*) {}
};
template <typename T>
struct Dft<T, 2> {
    static PM_HD void run(cx<T>* a) { dft2(a[0], a[1]); }
};
template <typename T>
struct Dft<T, 4> {
    static PM_HD void run(cx<T>* a) { dft4(a[0], a[1], a[2], a[3]); }
};
template <typename T>
struct Dft<T, 8> {
    static PM_HD void run(cx<T>* a) {
        const T r = T(0.70710678118654752440);
        dft4(a[0], a[2], a[4], a[6]);  // even samples -> E[k] in a[0],a[2],a[4],a[6]
        dft4(a[1], a[3], a[5], a[7]);  // odd samples  -> O[k] in a[1],a[3],a[5],a[7]
        cx<T> o0 = a[1];
#if defined(__HIP_DEVICE_COMPILE__) && defined(PM_PACKED_F32)
        cx<T> o1, o2, o3;
        if constexpr (sizeof(T) == 4) {
            o1 = cmul_k(a[3], r, -r);
            o2 = cmul_k(a[5], T(0), T(-1));
            o3 = cmul_k(a[7], -r, -r);
        } else {
            o1 = cx<T>{r * (a[3].x + a[3].y), r * (a[3].y - a[3].x)};
            o2 = mul_mi(a[5]);
            o3 = cx<T>{r * (a[7].y - a[7].x), -r * (a[7].x + a[7].y)};
        }
#else
        cx<T> o1 = {r * (a[3].x + a[3].y), r * (a[3].y - a[3].x)};    // * w8^1
        cx<T> o2 = mul_mi(a[5]);                                        // * w8^2
        cx<T> o3 = {r * (a[7].y - a[7].x), -r * (a[7].x + a[7].y)};   // * w8^3
#endif
        cx<T> e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
        a[0] = e0 + o0; a[4] = e0 - o0;
        a[1] = e1 + o1; a[5] = e1 - o1;
        a[2] = e2 + o2; a[6] = e2 - o2;
        a[3] = e3 + o3; a[7] = e3 - o3;
    }
};
template <typename T>
struct Dft<T, 16> {
    static PM_HD void run(cx<T>* a) {
        const T c1 = T(0.92387953251128673848), s1 = T(0.38268343236508978178);
        const T r = T(0.70710678118654752440);
        // step 1: DFT-4 over n1 for each n2; a[n2 + 4 k1] = b[n2][k1]
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) dft4(a[n2], a[n2 + 4], a[n2 + 8], a[n2 + 12]);
        // step 2: b[n2][k1] *= w16^(n2 k1)
#if defined(__HIP_DEVICE_COMPILE__) && defined(PM_PACKED_F32)
        if constexpr (sizeof(T) == 4) {
            a[5] = cmul_k(a[5], c1, -s1);
            a[9] = cmul_k(a[9], r, -r);
            a[13] = cmul_k(a[13], s1, -c1);
            a[6] = cmul_k(a[6], r, -r);
            a[10] = cmul_k(a[10], T(0), T(-1));
            a[14] = cmul_k(a[14], -r, -r);
            a[7] = cmul_k(a[7], s1, -c1);
            a[11] = cmul_k(a[11], -r, -r);
            a[15] = cmul_k(a[15], -c1, s1);
        } else
#endif
        {
        a[5] = cmul(a[5], cx<T>{c1, -s1});     // n2=1,k1=1 : w^1
        a[9] = cmul(a[9], cx<T>{r, -r});       // n2=1,k1=2 : w^2
        a[13] = cmul(a[13], cx<T>{s1, -c1});   // n2=1,k1=3 : w^3
        a[6] = cmul(a[6], cx<T>{r, -r});       // n2=2,k1=1 : w^2
        a[10] = mul_mi(a[10]);                 // n2=2,k1=2 : w^4
        a[14] = cmul(a[14], cx<T>{-r, -r});    // n2=2,k1=3 : w^6
        a[7] = cmul(a[7], cx<T>{s1, -c1});     // n2=3,k1=1 : w^3
        a[11] = cmul(a[11], cx<T>{-r, -r});    // n2=3,k1=2 : w^6
        a[15] = cmul(a[15], cx<T>{-c1, s1});   // n2=3,k1=3 : w^9
        }
        // step 3: DFT-4 over n2 for each k1; a[4 k1 + k2] = X[k1 + 4 k2]
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) dft4(a[4 * k1], a[4 * k1 + 1], a[4 * k1 + 2], a[4 * k1 + 3]);
        // transpose the 4x4 so that a[k] = X[k]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = i + 1; jj < 4; ++jj) {
                cx<T> tmp = a[4 * i + jj];
                a[4 * i + jj] = a[4 * jj + i];
                a[4 * jj + i] = tmp;
            }
    }
};

// ---------------------------------------------------------------------------
// twiddles of one stage for one thread: W^(4a b0) (a < R/4) and W^(b b0) (b < 4) per butterfly q; the
// remaining W^((4a+b) b0) are products (one extra rounding).  They depend on the thread slot only, not on
// the sequence.
// ---------------------------------------------------------------------------
// which configurations take W and W^4 from the table and form the other four twiddles of a butterfly as products (PM_TW_TABLE: none)
template <typename C>
constexpr bool tw_powers() {
#ifdef PM_TW_TABLE
    return false;
#else
    return !(sizeof(typename C::T) == 4 && C::LOGN == 12 && C::CI == 1);
#endif
}

template <typename C, int S>
struct StageTw {
    static constexpr int R = C::radix(S), Q = C::P / R;
    static constexpr int NA = (R / 4 > 1) ? R / 4 : 1;
    cx<typename C::T> wa[Q][NA], wb[Q][4];
};

template <typename C, int S>
PM_HD void load_stage_tw(StageTw<C, S>& w, int t, const cx<typename C::T>* __restrict__ tw) {
    constexpr int R = C::radix(S), NS = C::ns(S), Q = C::P / R, NP = C::TPS;
    if constexpr (S > 0 && R > 1) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int j = t + q * NP;
            const int base = (j & (NS - 1)) * (C::N / (NS * R));
            if constexpr (R == 2) {
                w.wb[q][1] = tw[base];
            } else {
                if constexpr (!tw_powers<C>()) {
#pragma unroll
                    for (int a = 1; a < R / 4; ++a) w.wa[q][a] = tw[4 * a * base];
#pragma unroll
                    for (int b = 1; b < 4; ++b) w.wb[q][b] = tw[b * base];
                } else {
                // W and W^4 only; stage_compute forms W^2, W^3, W^8, W^12 (stage_tw_powers): in the last stage of a 4096-point transform the
                // lanes' W^8 and W^12 entries are 32 and 48 different 128 B lines per wave-load, against 4 for W
                if constexpr (R / 4 > 1) w.wa[q][1] = tw[4 * base];
                w.wb[q][1] = tw[base];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// one radix-R stage on the registers of a thread (all E sequences share the
// twiddles: in column mode they are adjacent columns with the same row index)
// ---------------------------------------------------------------------------
template <typename C, int S, int E0 = 0, int E1 = C::E>
PM_HD void stage_compute(cx<typename C::T> (&v)[C::E][C::P], const StageTw<C, S>& w) {
    using T = typename C::T;
    constexpr int R = C::radix(S), Q = C::P / R;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if constexpr (S > 0 && R > 1) {
            if constexpr (R == 2) {
#pragma unroll
                for (int e = E0; e < E1; ++e) v[e][Q + q] = cmul(v[e][Q + q], w.wb[q][1]);
            } else {
                // the twiddles of this butterfly: W^b (b < 4) and W^(4a) (a < R/4); the loader brought W and W^4, the other powers are two
                // or four complex products here (PM_TW_TABLE: all six from the table, as until round 4)
                cx<T> wa[4], wb[4];
                wb[1] = w.wb[q][1];
                if constexpr (R / 4 > 1) wa[1] = w.wa[q][1];
                if constexpr (!tw_powers<C>()) {
                    wb[2] = w.wb[q][2];
                    wb[3] = w.wb[q][3];
                    if constexpr (R / 4 > 2) wa[2] = w.wa[q][2];
                    if constexpr (R / 4 > 3) wa[3] = w.wa[q][3];
                } else {
                    wb[2] = cmul(wb[1], wb[1]);
                    wb[3] = cmul(wb[2], wb[1]);
                    if constexpr (R / 4 > 2) wa[2] = cmul(wa[1], wa[1]);
                    if constexpr (R / 4 > 3) wa[3] = cmul(wa[2], wa[1]);
                }
#pragma unroll
                for (int a = 1; a < R / 4; ++a) {
#pragma unroll
                    for (int e = E0; e < E1; ++e) v[e][(4 * a) * Q + q] = cmul(v[e][(4 * a) * Q + q], wa[a]);
                }
#pragma unroll
                for (int b = 1; b < 4; ++b) {
#pragma unroll
                    for (int e = E0; e < E1; ++e) v[e][b * Q + q] = cmul(v[e][b * Q + q], wb[b]);
#pragma unroll
                    for (int a = 1; a < R / 4; ++a) {
                        const cx<T> wab = cmul(wa[a], wb[b]);
#pragma unroll
                        for (int e = E0; e < E1; ++e)
                            v[e][(4 * a + b) * Q + q] = cmul(v[e][(4 * a + b) * Q + q], wab);
                    }
                }
            }
        }
#pragma unroll
        for (int e = E0; e < E1; ++e) {
            cx<T> a[R];
#pragma unroll
            for (int k = 0; k < R; ++k) a[k] = v[e][k * Q + q];
            Dft<T, R>::run(a);
#pragma unroll
            for (int k = 0; k < R; ++k) v[e][k * Q + q] = a[k];
        }
    }
}

// convenience: load the stage twiddles, then compute
template <typename C, int S>
PM_HD void stage_compute(cx<typename C::T> (&v)[C::E][C::P], int t, const cx<typename C::T>* __restrict__ tw) {
    StageTw<C, S> w;
    load_stage_tw<C, S>(w, t, tw);
    stage_compute<C, S>(v, w);
}

template <typename C>
PM_HD int lds_addr(int bo, int cl, int idx) {
    return bo * (C::PADN * C::CI) + (idx + (idx >> 4)) * C::CI + cl;
}

// scatter the outputs of stage S (chunk = sequence e, component comp) into LDS.
// pad(ex + k*NS) == pad(ex) + k*(NS + NS/16) for NS >= 16 (ex is a multiple of 16 when NS == 1), so the
// R scatter addresses are one base + compile-time offsets (ds_write immediates).
template <typename C, int S, typename LT>
PM_HD void exch_write(const cx<typename C::T> (&v)[C::E][C::P], int e, int comp, ThreadPos pos, LT* lds) {
    constexpr int R = C::radix(S), NS = C::ns(S), Q = C::P / R, NP = C::TPS;
    constexpr int KSTEP = (NS >= 16 ? NS + NS / 16 : NS) * C::CI;
    constexpr bool AFFINE = (NS >= 16) || (R == 16);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int j = pos.t + q * NP;
        const int ex = (j / NS) * (NS * R) + (j & (NS - 1));
        const int a0 = lds_addr<C>(pos.bo, pos.cl, ex);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int a = AFFINE ? a0 + k * KSTEP : lds_addr<C>(pos.bo, pos.cl, ex + k * NS);
            if constexpr (C::COMP == 1)
                lds[a] = v[e][k * Q + q];
            else
                lds[a] = comp == 0 ? v[e][k * Q + q].x : v[e][k * Q + q].y;
        }
    }
}

// gather v[m] = y[t + m N/P]; pad(t + m*TPS) == pad(t) + m*(TPS + TPS/16) when 16 | TPS
template <typename C, typename LT>
PM_HD void exch_read(cx<typename C::T> (&v)[C::E][C::P], int e, int comp, ThreadPos pos, const LT* lds) {
    constexpr bool AFFINE = (C::TPS % 16) == 0;
    constexpr int MSTEP = (C::TPS + C::TPS / 16) * C::CI;
    const int a0 = lds_addr<C>(pos.bo, pos.cl, pos.t);
#pragma unroll
    for (int m = 0; m < C::P; ++m) {
        const int a = AFFINE ? a0 + m * MSTEP : lds_addr<C>(pos.bo, pos.cl, pos.t + m * C::TPS);
        if constexpr (C::COMP == 1) {
            v[e][m] = lds[a];
        } else {
            if (comp == 0)
                v[e][m].x = lds[a];
            else
                v[e][m].y = lds[a];
        }
    }
}

template <typename C>
struct LdsType {
    using type = cx<typename C::T>;
};
template <typename T, int L, int CI, int E, int BO, int LP>
struct LdsType<FftCfg<T, L, CI, E, BO, 2, LP>> {
    using type = T;
};

#if defined(__HIPCC__)
// full transform of the registers of this thread; all threads of the workgroup must call it
template <typename C, int S = 0>
__device__ __forceinline__ void fft_run(cx<typename C::T> (&v)[C::E][C::P], ThreadPos pos, void* lds_raw,
                                        const cx<typename C::T>* __restrict__ tw) {
    using LT = typename LdsType<C>::type;
    LT* lds = reinterpret_cast<LT*>(lds_raw);
    stage_compute<C, S>(v, pos.t, tw);
    if constexpr (S + 1 < C::NSTAGE) {
#pragma unroll
        for (int e = 0; e < C::E; ++e) {
#pragma unroll
            for (int comp = 0; comp < C::COMP; ++comp) {
                exch_write<C, S>(v, e, comp, pos, lds);
                __syncthreads();
                exch_read<C>(v, e, comp, pos, lds);
                __syncthreads();
            }
        }
        fft_run<C, S + 1>(v, pos, lds_raw, tw);
    }
}

// Software-pipelined transform for two sequences per thread (complex64 column pass): the LDS exchange of one
// sequence is issued right before the butterflies of the other, so ds_write traffic drains under VALU work
// instead of in front of a barrier.  Same arithmetic and the same number of barriers as fft_run.
//   invariant on entry to stage S > 0: v[0] is exchanged and ready for stage S; v[1] holds the un-exchanged
//   output of stage S - 1.
// The twiddles of stage S + 1 are requested after the exchange of stage S (requesting them a stage ahead measured equal at 2048^2 ..
// 8192^2 and costs ~12 VGPRs: 8192-point rows would drop to 3 waves / SIMD).
template <typename C, int S>
__device__ __forceinline__ void fft_run_pipe2_stage(cx<typename C::T> (&v)[C::E][C::P], ThreadPos pos, void* lds_raw,
                                                    const cx<typename C::T>* __restrict__ tw, const StageTw<C, S>& w) {
    using LT = typename LdsType<C>::type;
    LT* lds = reinterpret_cast<LT*>(lds_raw);
    constexpr int SN = (S + 1 < C::NSTAGE) ? S + 1 : S;
    StageTw<C, SN> wn;
    if constexpr (S == 0) {
        stage_compute<C, 0, 0, 1>(v, w);
        if constexpr (C::NSTAGE == 1) {
            stage_compute<C, 0, 1, 2>(v, w);
        } else {
            exch_write<C, 0>(v, 0, 0, pos, lds);
            stage_compute<C, 0, 1, 2>(v, w);
            __syncthreads();
            exch_read<C>(v, 0, 0, pos, lds);
            __syncthreads();
            load_stage_tw<C, SN>(wn, pos.t, tw);
            fft_run_pipe2_stage<C, 1>(v, pos, lds_raw, tw, wn);
        }
    } else {
        exch_write<C, S - 1>(v, 1, 0, pos, lds);
        stage_compute<C, S, 0, 1>(v, w);
        __syncthreads();
        exch_read<C>(v, 1, 0, pos, lds);
        if constexpr (S + 1 < C::NSTAGE) {
            __syncthreads();
            exch_write<C, S>(v, 0, 0, pos, lds);
            stage_compute<C, S, 1, 2>(v, w);
            __syncthreads();
            exch_read<C>(v, 0, 0, pos, lds);
            __syncthreads();
            load_stage_tw<C, SN>(wn, pos.t, tw);
            fft_run_pipe2_stage<C, S + 1>(v, pos, lds_raw, tw, wn);
        } else {
            stage_compute<C, S, 1, 2>(v, w);   // no trailing barrier: callers that reuse the LDS synchronise themselves
        }
    }
}

template <typename C>
__device__ __forceinline__ void fft_run_pipe2(cx<typename C::T> (&v)[C::E][C::P], ThreadPos pos, void* lds_raw,
                                              const cx<typename C::T>* __restrict__ tw) {
    static_assert(C::E == 2 && C::COMP == 1, "two sequences per thread, complex exchange");
    StageTw<C, 0> none;
    fft_run_pipe2_stage<C, 0>(v, pos, lds_raw, tw, none);
}
#endif

}  // namespace pm
