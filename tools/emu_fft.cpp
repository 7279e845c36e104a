// CPU emulation of the HIP FFT kernels' per-thread logic (test scaffolding).
//
// The build container has no GPU.  fft_engine.h / fft_io.h are written as
// __host__ __device__ code, so this program runs the exact load -> stages ->
// LDS exchange -> store sequence of prysm_amd/csrc/fft_kernels.hip for every
// thread of every workgroup, with a std::vector standing in for LDS and phase
// boundaries standing in for __syncthreads(), and compares against a naive
// long-double DFT.  It validates index arithmetic, twiddles, butterflies,
// padding / shift / crop maps and the tiled intermediate before GPU minutes
// are spent.  It is NOT a product code path (nothing in prysm_amd calls it).
//
// build: g++ -O2 -std=c++17 -I prysm_amd/csrc tools/emu_fft.cpp -o /tmp/emu_fft
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "fft_io.h"
#include "bluestein.h"

using namespace pm;
typedef long double ld;
typedef std::complex<ld> cld;

template <typename T>
static std::vector<cx<T>> make_tw(int N) {
    std::vector<cx<T>> tw(N);
    const ld pi = acosl(-1.0L);
    for (int i = 0; i < N; ++i) tw[i] = {T(cosl(-2 * pi * i / N)), T(sinl(-2 * pi * i / N))};
    return tw;
}

template <typename C>
struct Regs {
    cx<typename C::T> v[C::E][C::P];
};

template <typename C, int S>
static void emu_stages(std::vector<Regs<C>>& regs, std::vector<typename LdsType<C>::type>& lds,
                       const cx<typename C::T>* tw) {
    for (int tid = 0; tid < C::NT; ++tid) stage_compute<C, S>(regs[tid].v, thread_pos<C>(tid).t, tw);
    if constexpr (S + 1 < C::NSTAGE) {
        for (int e = 0; e < C::E; ++e)
            for (int comp = 0; comp < C::COMP; ++comp) {
                for (int tid = 0; tid < C::NT; ++tid)
                    exch_write<C, S>(regs[tid].v, e, comp, thread_pos<C>(tid), lds.data());
                for (int tid = 0; tid < C::NT; ++tid)
                    exch_read<C>(regs[tid].v, e, comp, thread_pos<C>(tid), lds.data());
            }
        emu_stages<C, S + 1>(regs, lds, tw);
    }
}

template <typename C, bool COL, typename L, typename S>
static void emu_kernel(int nblocks, const L& lp, const S& sp, const cx<typename C::T>* tw) {
    std::vector<Regs<C>> regs(C::NT);
    std::vector<typename LdsType<C>::type> lds(C::LDS_ELEMS + 1);
    for (int g = 0; g < nblocks; ++g) {
        for (int tid = 0; tid < C::NT; ++tid) {
            ThreadPos pos = thread_pos<C>(tid);
            int unit = pair_remap(g, nblocks);
            if (COL) unit = unit * C::BO + pos.bo;
            load<C>(lp, unit, pos, regs[tid].v);
        }
        emu_stages<C, 0>(regs, lds, tw);
        for (int tid = 0; tid < C::NT; ++tid) {
            ThreadPos pos = thread_pos<C>(tid);
            int unit = pair_remap(g, nblocks);
            if (COL) unit = unit * C::BO + pos.bo;
            store<C>(sp, unit, pos, regs[tid].v);
        }
    }
}

static int g_fail = 0;
static void report(const char* what, double err, double tol) {
    const bool ok = err < tol;
    printf("%-58s err=%.3e  %s\n", what, err, ok ? "ok" : "FAIL");
    if (!ok) ++g_fail;
}

// --- 1-D row transform, natural in / natural out, with zero-pad + shift maps -------------
template <typename T, int LOGN, int BO, int COMP, int E = 1, int LOGP = 4>
static void test_row(int nseq, int in_len, int in_off, int in_shift, int out_shift, bool inverse) {
    using C = FftCfg<T, LOGN, 1, E, BO, COMP, LOGP>;
    const int N = C::N;
    std::mt19937 rng(LOGN * 131 + nseq);
    std::normal_distribution<double> nd;
    std::vector<cx<T>> x(size_t(nseq) * in_len), y(size_t(nseq) * N, cx<T>{T(-7), T(-7)});
    for (auto& e : x) e = {T(nd(rng)), T(nd(rng))};
    auto tw = make_tw<T>(N);
    RowLoadNat<T> lp{x.data(), in_len, AxisMap{N, in_len, in_off, in_shift}, nseq, inverse ? 1 : 0, 0};
    RowStoreNat<T> sp{y.data(), N, AxisMap{N, N, 0, out_shift}, nseq, inverse ? 1 : 0, T(1), 0, AxisMap{1, 1, 0, 0}};
    const int nblk = (nseq + C::BO * E - 1) / (C::BO * E);
    emu_kernel<C, false>(nblk, lp, sp, tw.data());
    const ld pi = acosl(-1.0L);
    double err = 0, nrm = 0;
    for (int s = 0; s < nseq; ++s) {
        std::vector<cld> p(N, cld(0, 0));
        for (int i = 0; i < N; ++i) {  // logical padded+shifted input
            int pp = (i + in_shift) % N, q = pp - in_off;
            if (q >= 0 && q < in_len) p[i] = cld(x[size_t(s) * in_len + q].x, x[size_t(s) * in_len + q].y);
        }
        for (int k = 0; k < N; ++k) {
            cld acc(0, 0);
            for (int n = 0; n < N; ++n) {
                ld ang = (inverse ? 2 : -2) * pi * ld((int64_t(n) * k) % N) / N;
                acc += p[n] * cld(cosl(ang), sinl(ang));
            }
            int pos = (k + out_shift) % N;
            cx<T> got = y[size_t(s) * N + pos];
            err = fmax(err, (double)std::abs(acc - cld(got.x, got.y)));
            nrm = fmax(nrm, (double)std::abs(acc));
        }
    }
    char buf[128];
    snprintf(buf, sizeof buf, "row %s N=%d BO=%d COMP=%d E=%d P=%d nseq=%d len=%d off=%d sh=%d/%d %s",
             sizeof(T) == 4 ? "c64" : "c128", N, BO, COMP, E, C::P, nseq, in_len, in_off, in_shift, out_shift,
             inverse ? "inv" : "fwd");
    report(buf, err / nrm, sizeof(T) == 4 ? 2e-6 : 1e-14);
}

// naive separable 2-D DFT of the padded / rotated input, scaled 1/sqrt(MN): B[k][c]
template <typename T>
static std::vector<cld> naive_2d(const std::vector<cx<T>>& x, int in_rows, int in_cols, int M, int N, int offy, int offx, int shy,
                                 int shx, bool inverse) {
    const ld pi = acosl(-1.0L);
    std::vector<cld> P(size_t(M) * N, cld(0, 0));
    for (int r = 0; r < M; ++r)
        for (int c = 0; c < N; ++c) {
            int qr = (r + shy) % M - offy, qc = (c + shx) % N - offx;
            if (qr >= 0 && qr < in_rows && qc >= 0 && qc < in_cols)
                P[size_t(r) * N + c] = cld(x[size_t(qr) * in_cols + qc].x, x[size_t(qr) * in_cols + qc].y);
        }
    std::vector<cld> A(size_t(M) * N), B(size_t(M) * N);
    const ld sg = inverse ? 2 : -2;
    for (int r = 0; r < M; ++r)
        for (int k = 0; k < N; ++k) {
            cld acc(0, 0);
            for (int n = 0; n < N; ++n) {
                ld ang = sg * pi * ld((int64_t(n) * k) % N) / N;
                acc += P[size_t(r) * N + n] * cld(cosl(ang), sinl(ang));
            }
            A[size_t(r) * N + k] = acc;
        }
    for (int c = 0; c < N; ++c)
        for (int k = 0; k < M; ++k) {
            cld acc(0, 0);
            for (int n = 0; n < M; ++n) {
                ld ang = sg * pi * ld((int64_t(n) * k) % M) / M;
                acc += A[size_t(n) * N + c] * cld(cosl(ang), sinl(ang));
            }
            B[size_t(k) * N + c] = acc / sqrtl(ld(M) * N);
        }
    return B;
}

// --- 2-D: row pass -> tiled intermediate -> column pass, vs naive 2-D DFT ---------------
template <typename T, int LOGM, int LOGN, int RBO, int RCOMP, int CCI, int CE, int CBO, int CCOMP>
static void test_2d(int in_rows, int in_cols, bool shifts, int epilogue, bool inverse, int out_rows, int out_cols, int log_k = 0) {
    using RC = FftCfg<T, LOGN, 1, 1, RBO, RCOMP>;      // row transform of length N (columns)
    using CC = FftCfg<T, LOGM, CCI, CE, CBO, CCOMP>;   // column transform of length M (rows)
    const int M = CC::N, N = RC::N, TC = CCI * CE;
    const int TL = TC << log_k;
    int log_tc = 0;
    while ((1 << log_tc) < TL) ++log_tc;
    std::mt19937 rng(LOGM * 17 + LOGN);
    std::normal_distribution<double> nd;
    std::vector<cx<T>> x(size_t(in_rows) * in_cols);
    for (auto& e : x) e = {T(nd(rng)), T(nd(rng))};
    const int offy = (M - in_rows + 1) / 2, offx = (N - in_cols + 1) / 2;  // pad2d: ceil(d/2)
    const int shy = shifts ? M / 2 : 0, shx = shifts ? N / 2 : 0;
    const int ntiles = (N + TC - 1) / TC;
    std::vector<cx<T>> W(size_t((N + TL - 1) / TL) * in_rows * TL, cx<T>{T(1e30), T(1e30)});
    auto twN = make_tw<T>(N);
    auto twM = make_tw<T>(M);
    // pass 1: one FFT per stored input row
    RowLoadNat<T> lp{x.data(), in_cols, AxisMap{N, in_cols, offx, shx}, in_rows, inverse ? 1 : 0, 0};
    RowStoreTiled<T> sp{W.data(), in_rows, log_tc};
    emu_kernel<RC, false>((in_rows + RC::BO - 1) / RC::BO, lp, sp, twN.data());
    // pass 2
    const int coffy = (M - out_rows + 1) / 2, coffx = (N - out_cols + 1) / 2;  // crop_center: ceil(p/2)
    std::vector<cx<T>> out(size_t(out_rows) * out_cols, cx<T>{T(-3), T(-3)});
    std::vector<T> outr(size_t(out_rows) * out_cols, T(-3));
    ColLoadTiled<T> cl{W.data(), in_rows, AxisMap{M, in_rows, offy, shy}, ntiles, log_k};
    ColStoreNat<T> cs{};
    cs.dst = epilogue ? (void*)outr.data() : (void*)out.data();
    cs.ld = out_cols;
    cs.ay = AxisMap{M, out_rows, coffy, shy};
    cs.ax = AxisMap{N, out_cols, coffx, shx};
    cs.conj = inverse ? 1 : 0;
    cs.epilogue = epilogue;
    cs.scale = T(1.0 / sqrt(double(M) * N));
    cs.weight = T(1);
    cs.mul_kind = MUL_NONE;
    cs.vec_ok = (out_cols % 2 == 0) ? 1 : 0;
    const int ngroups = (ntiles + CC::BO - 1) / CC::BO;
    emu_kernel<CC, true>(ngroups, cl, cs, twM.data());
    const std::vector<cld> B = naive_2d<T>(x, in_rows, in_cols, M, N, offy, offx, shy, shx, inverse);
    double err = 0, nrm = 0;
    for (int k = 0; k < M; ++k)
        for (int c = 0; c < N; ++c) {
            int qy = (k + shy) % M - coffy, qx = (c + shx) % N - coffx;
            if (qy < 0 || qy >= out_rows || qx < 0 || qx >= out_cols) continue;
            cld ref = B[size_t(k) * N + c];
            if (epilogue) {
                ld r2 = std::norm(ref);
                err = fmax(err, (double)fabsl(r2 - outr[size_t(qy) * out_cols + qx]));
                nrm = fmax(nrm, (double)r2);
            } else {
                cx<T> got = out[size_t(qy) * out_cols + qx];
                err = fmax(err, (double)std::abs(ref - cld(got.x, got.y)));
                nrm = fmax(nrm, (double)std::abs(ref));
            }
        }
    char buf[160];
    snprintf(buf, sizeof buf, "2d %s %dx%d in=%dx%d out=%dx%d sh=%d epi=%d %s TC=%d TL=%d", sizeof(T) == 4 ? "c64" : "c128",
             M, N, in_rows, in_cols, out_rows, out_cols, (int)shifts, epilogue, inverse ? "inv" : "fwd", TC, TL);
    report(buf, err / nrm, sizeof(T) == 4 ? 3e-6 : 1e-13);
}

// --- folded 2-D: the row pass (two rows (i, i + M/2) per thread) takes one radix-2 step of the column transform, the
// column pass runs two planes of M/2-point tiles that write the even / odd output rows -----------------------------
template <typename T, int LOGM, int LOGN, int RBO, int RCOMP, int CCI, int CE, int CBO, int CCOMP>
static void test_2d_fold(int in_cols, bool shifts, int epilogue, bool inverse, int out_rows, int out_cols, int log_k = 0) {
    using RC = FftCfg<T, LOGN, 1, 2, RBO, RCOMP>;
    using CC = FftCfg<T, LOGM - 1, CCI, CE, CBO, CCOMP>;
    const int M = 1 << LOGM, H = M / 2, N = RC::N, TC = CCI * CE, TL = TC << log_k;
    int log_tc = 0;
    while ((1 << log_tc) < TL) ++log_tc;
    std::mt19937 rng(LOGM * 19 + LOGN);
    std::normal_distribution<double> nd;
    std::vector<cx<T>> x(size_t(M) * in_cols);
    for (auto& e : x) e = {T(nd(rng)), T(nd(rng))};
    const int offx = (N - in_cols + 1) / 2;
    const int shy = shifts ? M / 2 : 0, shx = shifts ? N / 2 : 0;
    const int ntiles = (N + TC - 1) / TC, ntl = (N + TL - 1) / TL;
    const int64_t plane = int64_t(ntl) * H * TL;
    std::vector<cx<T>> W(size_t(2 * plane), cx<T>{T(1e30), T(1e30)});
    auto twN = make_tw<T>(N);
    auto twM = make_tw<T>(M);
    auto twH = make_tw<T>(H);
    RowLoadNat<T> lp{x.data(), in_cols, AxisMap{N, in_cols, offx, shx}, M, inverse ? 1 : 0, 0, 0, 0, H};
    RowStoreFold<T> sp{W.data(), plane, H, log_tc, twM.data(), shifts ? 1 : 0, 0};
    emu_kernel<RC, false>((H + RC::BO - 1) / RC::BO, lp, sp, twN.data());
    const int coffy = (M - out_rows + 1) / 2, coffx = (N - out_cols + 1) / 2;
    std::vector<cx<T>> out(size_t(out_rows) * out_cols, cx<T>{T(-3), T(-3)});
    std::vector<T> outr(size_t(out_rows) * out_cols, T(-3));
    ColLoadTiled<T> cl{W.data(), H, AxisMap{H, H, 0, 0}, ntiles, log_k, plane};
    ColStoreNat<T> cs{};
    cs.dst = epilogue ? (void*)outr.data() : (void*)out.data();
    cs.ld = 2 * out_cols;
    cs.ay = AxisMap{H, out_rows / 2, coffy / 2, shy / 2};
    cs.ax = AxisMap{N, out_cols, coffx, shx};
    cs.conj = inverse ? 1 : 0;
    cs.epilogue = epilogue;
    cs.scale = T(1.0 / sqrt(double(M) * N));
    cs.weight = T(1);
    cs.mul_kind = MUL_NONE;
    cs.vec_ok = (out_cols % 2 == 0) ? 1 : 0;
    cs.bstride = out_cols;
    const int ngroups = (ntiles + CC::BO - 1) / CC::BO;
    for (int plane_id = 0; plane_id < 2; ++plane_id)
        emu_kernel<CC, true>(ngroups, at_batch(cl, plane_id), at_batch(cs, plane_id), twH.data());
    const std::vector<cld> B = naive_2d<T>(x, M, in_cols, M, N, 0, offx, shy, shx, inverse);
    double err = 0, nrm = 0;
    for (int k = 0; k < M; ++k)
        for (int c = 0; c < N; ++c) {
            int qy = (k + shy) % M - coffy, qx = (c + shx) % N - coffx;
            if (qy < 0 || qy >= out_rows || qx < 0 || qx >= out_cols) continue;
            cld ref = B[size_t(k) * N + c];
            if (epilogue) {
                ld r2 = std::norm(ref);
                err = fmax(err, (double)fabsl(r2 - outr[size_t(qy) * out_cols + qx]));
                nrm = fmax(nrm, (double)r2);
            } else {
                cx<T> got = out[size_t(qy) * out_cols + qx];
                err = fmax(err, (double)std::abs(ref - cld(got.x, got.y)));
                nrm = fmax(nrm, (double)std::abs(ref));
            }
        }
    char buf[160];
    snprintf(buf, sizeof buf, "2d FOLD %s %dx%d in=%dx%d out=%dx%d sh=%d epi=%d %s TC=%d TL=%d", sizeof(T) == 4 ? "c64" : "c128",
             M, N, M, in_cols, out_rows, out_cols, (int)shifts, epilogue, inverse ? "inv" : "fwd", TC, TL);
    report(buf, err / nrm, sizeof(T) == 4 ? 3e-6 : 1e-13);
}

// naive ifft2(fft2(pad(x)) * H) without the 1/(MN): the reference of the fused chain
template <typename T>
static std::vector<cld> naive_fused(const std::vector<cx<T>>& x, int in_rows, int in_cols, int M, int N, int offy, int offx, int shy,
                                    int shx, const std::vector<cx<T>>& H, const std::vector<cx<T>>& hy, const std::vector<cx<T>>& hx,
                                    bool separable, bool mconj) {
    const ld pi = acosl(-1.0L);
    std::vector<cld> P(size_t(M) * N, cld(0, 0)), A(size_t(M) * N), B(size_t(M) * N);
    for (int r = 0; r < M; ++r)
        for (int c = 0; c < N; ++c) {
            int qr = (r + shy) % M - offy, qc = (c + shx) % N - offx;
            if (qr >= 0 && qr < in_rows && qc >= 0 && qc < in_cols)
                P[size_t(r) * N + c] = cld(x[size_t(qr) * in_cols + qc].x, x[size_t(qr) * in_cols + qc].y);
        }
    auto dft2 = [&](std::vector<cld>& Z, ld sg) {
        for (int r = 0; r < M; ++r)
            for (int k = 0; k < N; ++k) {
                cld acc(0, 0);
                for (int n = 0; n < N; ++n) acc += Z[size_t(r) * N + n] * std::polar<ld>(1.0L, sg * pi * ld((int64_t(n) * k) % N) / N);
                A[size_t(r) * N + k] = acc;
            }
        for (int c = 0; c < N; ++c)
            for (int k = 0; k < M; ++k) {
                cld acc(0, 0);
                for (int n = 0; n < M; ++n) acc += A[size_t(n) * N + c] * std::polar<ld>(1.0L, sg * pi * ld((int64_t(n) * k) % M) / M);
                B[size_t(k) * N + c] = acc;
            }
        Z = B;
    };
    dft2(P, -2);
    for (int k = 0; k < M; ++k)
        for (int c = 0; c < N; ++c) {
            cld h = separable ? cld(hy[k].x, hy[k].y) * cld(hx[c].x, hx[c].y) : cld(H[size_t(k) * N + c].x, H[size_t(k) * N + c].y);
            if (mconj) h = std::conj(h);
            P[size_t(k) * N + c] *= h;
        }
    dft2(P, +2);
    return P;
}

// --- fused fft2 -> x H -> ifft2 (3 passes), vs naive ---------------------------------------------------------
template <typename T, int LOGM, int LOGN, int RBO, int CCI, int CE, int CBO, int CCOMP>
static void test_fused(int in_rows, int in_cols, int out_rows, int out_cols, int log_k, bool separable, bool mconj, bool shifts) {
    using RC = FftCfg<T, LOGN, 1, 1, RBO, 1>;
    using CC = FftCfg<T, LOGM, CCI, CE, CBO, CCOMP>;
    const int M = CC::N, N = RC::N, TC = CCI * CE, TL = TC << log_k;
    int ltl = 0;
    while ((1 << ltl) < TL) ++ltl;
    std::mt19937 rng(LOGM * 19 + LOGN + log_k);
    std::normal_distribution<double> nd;
    std::vector<cx<T>> x(size_t(in_rows) * in_cols), H(size_t(M) * N), hy(M), hx(N);
    for (auto& e : x) e = {T(nd(rng)), T(nd(rng))};
    for (auto& e : H) e = {T(nd(rng)), T(nd(rng))};
    for (auto& e : hy) e = {T(nd(rng)), T(nd(rng))};
    for (auto& e : hx) e = {T(nd(rng)), T(nd(rng))};
    const int offy = (M - in_rows + 1) / 2, offx = (N - in_cols + 1) / 2;
    const int shy = shifts ? M / 2 : 0, shx = shifts ? N / 2 : 0;
    const int coffy = (M - out_rows + 1) / 2, coffx = (N - out_cols + 1) / 2;
    const int ntl = (N + TL - 1) / TL, ntiles = (N + TC - 1) / TC;
    std::vector<cx<T>> W1(size_t(ntl) * in_rows * TL, cx<T>{T(1e30), T(1e30)}), W2(size_t(ntl) * M * TL, cx<T>{T(1e30), T(1e30)});
    auto twN = make_tw<T>(N);
    auto twM = make_tw<T>(M);
    RowLoadNat<T> lp{x.data(), in_cols, AxisMap{N, in_cols, offx, shx}, in_rows, 0, 0};
    RowStoreTiled<T> sp{W1.data(), in_rows, ltl};
    emu_kernel<RC, false>((in_rows + RC::BO - 1) / RC::BO, lp, sp, twN.data());
    // pass B
    ColLoadTiled<T> cl{W1.data(), in_rows, AxisMap{M, in_rows, offy, shy}, ntiles, log_k};
    MidMul<T> mm{separable ? MUL_SEPARABLE : MUL_FULL, mconj ? 1 : 0, separable ? hy.data() : H.data(), hx.data(), N, N};
    ColStoreTiled<T> cst{W2.data(), M, ntiles, log_k};
    {
        std::vector<Regs<CC>> regs(CC::NT);
        std::vector<typename LdsType<CC>::type> lds(CC::LDS_ELEMS + 1);
        const int ngroups = (ntiles + CC::BO - 1) / CC::BO;
        for (int g = 0; g < ngroups; ++g) {
            for (int tid = 0; tid < CC::NT; ++tid) {
                ThreadPos pos = thread_pos<CC>(tid);
                load<CC>(cl, g * CC::BO + pos.bo, pos, regs[tid].v);
            }
            emu_stages<CC, 0>(regs, lds, twM.data());
            for (int tid = 0; tid < CC::NT; ++tid) {
                ThreadPos pos = thread_pos<CC>(tid);
                mid_multiply_conj<CC>(mm, g * CC::BO + pos.bo, pos, regs[tid].v);
            }
            emu_stages<CC, 0>(regs, lds, twM.data());
            for (int tid = 0; tid < CC::NT; ++tid) {
                ThreadPos pos = thread_pos<CC>(tid);
                for (int e = 0; e < CC::E; ++e)
                    for (int m = 0; m < CC::P; ++m) regs[tid].v[e][m].y = -regs[tid].v[e][m].y;
                store<CC>(cst, g * CC::BO + pos.bo, pos, regs[tid].v);
            }
        }
    }
    // pass C
    std::vector<cx<T>> out(size_t(out_rows) * out_cols, cx<T>{T(-3), T(-3)});
    RowLoadTiled<T> rl{W2.data(), M, ltl, 0, M, 1};
    RowStoreNat<T> rs{out.data(), out_cols, AxisMap{N, out_cols, coffx, shx}, M, 1, T(1.0 / (double(M) * N)), 1,
                      AxisMap{M, out_rows, coffy, shy}};
    emu_kernel<RC, false>((M + RC::BO - 1) / RC::BO, rl, rs, twN.data());
    std::vector<cld> P = naive_fused<T>(x, in_rows, in_cols, M, N, offy, offx, shy, shx, H, hy, hx, separable, mconj);
    double err = 0, nrm = 0;
    for (int r = 0; r < M; ++r)
        for (int c = 0; c < N; ++c) {
            int qy = (r + shy) % M - coffy, qx = (c + shx) % N - coffx;
            if (qy < 0 || qy >= out_rows || qx < 0 || qx >= out_cols) continue;
            cld ref = P[size_t(r) * N + c] / (ld(M) * N);
            cx<T> got = out[size_t(qy) * out_cols + qx];
            err = fmax(err, (double)std::abs(ref - cld(got.x, got.y)));
            nrm = fmax(nrm, (double)std::abs(ref));
        }
    char buf[160];
    snprintf(buf, sizeof buf, "fused %s %dx%d in=%dx%d out=%dx%d TL=%d sep=%d conj=%d sh=%d", sizeof(T) == 4 ? "c64" : "c128", M, N,
             in_rows, in_cols, out_rows, out_cols, TL, (int)separable, (int)mconj, (int)shifts);
    report(buf, err / nrm, sizeof(T) == 4 ? 5e-6 : 1e-13);
}

// --- folded fused chain: fold in the first row pass, M/2-point column FFT x H x IFFT per plane, unfold in the last ----
template <typename T, int LOGM, int LOGN, int RBO, int RCOMP, int CCI, int CE, int CBO, int CCOMP>
static void test_fused_fold(int in_cols, int out_rows, int out_cols, int log_k, bool separable, bool mconj, bool shifts) {
    using RC = FftCfg<T, LOGN, 1, 2, RBO, RCOMP>;
    using CC = FftCfg<T, LOGM - 1, CCI, CE, CBO, CCOMP>;
    const int M = 1 << LOGM, Hh = M / 2, N = RC::N, TC = CCI * CE, TL = TC << log_k;
    int ltl = 0;
    while ((1 << ltl) < TL) ++ltl;
    std::mt19937 rng(LOGM * 23 + LOGN + log_k);
    std::normal_distribution<double> nd;
    std::vector<cx<T>> x(size_t(M) * in_cols), H(size_t(M) * N), hy(M), hx(N);
    for (auto& e : x) e = {T(nd(rng)), T(nd(rng))};
    for (auto& e : H) e = {T(nd(rng)), T(nd(rng))};
    for (auto& e : hy) e = {T(nd(rng)), T(nd(rng))};
    for (auto& e : hx) e = {T(nd(rng)), T(nd(rng))};
    const int offx = (N - in_cols + 1) / 2;
    const int shy = shifts ? M / 2 : 0, shx = shifts ? N / 2 : 0;
    const int coffy = (M - out_rows + 1) / 2, coffx = (N - out_cols + 1) / 2;
    const int ntl = (N + TL - 1) / TL, ntiles = (N + TC - 1) / TC;
    const int64_t plane = int64_t(ntl) * Hh * TL;
    std::vector<cx<T>> W(size_t(2 * plane), cx<T>{T(1e30), T(1e30)});
    auto twN = make_tw<T>(N);
    auto twM = make_tw<T>(M);
    auto twH = make_tw<T>(Hh);
    RowLoadNat<T> lp{x.data(), in_cols, AxisMap{N, in_cols, offx, shx}, M, 0, 0, 0, 0, Hh};
    RowStoreFold<T> sp{W.data(), plane, Hh, ltl, twM.data(), shifts ? 1 : 0, 0};
    emu_kernel<RC, false>((Hh + RC::BO - 1) / RC::BO, lp, sp, twN.data());
    ColLoadTiled<T> cl0{W.data(), Hh, AxisMap{Hh, Hh, 0, 0}, ntiles, log_k, plane};
    MidMul<T> mm0{separable ? MUL_SEPARABLE : MUL_FULL, mconj ? 1 : 0, separable ? hy.data() : H.data(), hx.data(), 2 * N, N,
                  separable ? 1 : N, 0, 2};
    ColStoreTiled<T> cst0{W.data(), Hh, ntiles, log_k, plane};
    for (int b = 0; b < 2; ++b) {
        const auto cl = at_batch(cl0, b);
        const auto mm = at_batch(mm0, b);
        const auto cst = at_batch(cst0, b);
        std::vector<Regs<CC>> regs(CC::NT);
        std::vector<typename LdsType<CC>::type> lds(CC::LDS_ELEMS + 1);
        const int ngroups = (ntiles + CC::BO - 1) / CC::BO;
        for (int g = 0; g < ngroups; ++g) {
            for (int tid = 0; tid < CC::NT; ++tid) {
                ThreadPos pos = thread_pos<CC>(tid);
                load<CC>(cl, g * CC::BO + pos.bo, pos, regs[tid].v);
            }
            emu_stages<CC, 0>(regs, lds, twH.data());
            for (int tid = 0; tid < CC::NT; ++tid) {
                ThreadPos pos = thread_pos<CC>(tid);
                mid_multiply_conj<CC>(mm, g * CC::BO + pos.bo, pos, regs[tid].v);
            }
            emu_stages<CC, 0>(regs, lds, twH.data());
            for (int tid = 0; tid < CC::NT; ++tid) {
                ThreadPos pos = thread_pos<CC>(tid);
                for (int e = 0; e < CC::E; ++e)
                    for (int m = 0; m < CC::P; ++m) regs[tid].v[e][m].y = -regs[tid].v[e][m].y;
                store<CC>(cst, g * CC::BO + pos.bo, pos, regs[tid].v);
            }
        }
    }
    std::vector<cx<T>> out(size_t(out_rows) * out_cols, cx<T>{T(-3), T(-3)});
    RowLoadFold<T> rl{W.data(), plane, Hh, ltl, twM.data(), 1, 0};
    RowStoreNat<T> rs{out.data(), out_cols, AxisMap{N, out_cols, coffx, shx}, M, 1, T(1.0 / (double(M) * N)), 1,
                      AxisMap{M, out_rows, coffy, shy}, 0, Hh};
    emu_kernel<RC, false>((Hh + RC::BO - 1) / RC::BO, rl, rs, twN.data());
    std::vector<cld> P = naive_fused<T>(x, M, in_cols, M, N, 0, offx, shy, shx, H, hy, hx, separable, mconj);
    double err = 0, nrm = 0;
    for (int r = 0; r < M; ++r)
        for (int c = 0; c < N; ++c) {
            int qy = (r + shy) % M - coffy, qx = (c + shx) % N - coffx;
            if (qy < 0 || qy >= out_rows || qx < 0 || qx >= out_cols) continue;
            cld ref = P[size_t(r) * N + c] / (ld(M) * N);
            cx<T> got = out[size_t(qy) * out_cols + qx];
            err = fmax(err, (double)std::abs(ref - cld(got.x, got.y)));
            nrm = fmax(nrm, (double)std::abs(ref));
        }
    char buf[160];
    snprintf(buf, sizeof buf, "fused FOLD %s %dx%d in=%dx%d out=%dx%d TL=%d sep=%d conj=%d sh=%d", sizeof(T) == 4 ? "c64" : "c128", M, N,
             M, in_cols, out_rows, out_cols, TL, (int)separable, (int)mconj, (int)shifts);
    report(buf, err / nrm, sizeof(T) == 4 ? 5e-6 : 1e-13);
}

// --- Bluestein (bluestein.h / bluestein.hip): a non-power-of-two length n through engine transforms of length MB.  Rows run
// the emulated engine passes with exactly the load / store descriptors blue_rows builds; columns restate the natural column
// pass by its contract (pad window on the load, crop window on the store, conj-in / conj-out inverse) --------------------
template <typename T, int LOGMB, int BO>
static void test_blue_rows(int n, int nseq, int in_len, int in_off, int in_shift, bool conj_in, bool real_in) {
    using C = FftCfg<T, LOGMB, 1, 1, BO, 1>;
    const int mb = blue_conv_len(n);
    if (mb != C::N) { printf("blue rows: MB mismatch %d vs %d\n", mb, C::N); ++g_fail; return; }
    std::mt19937 rng(n * 7 + nseq);
    std::normal_distribution<double> nd;
    std::vector<cx<T>> x(size_t(nseq) * in_len);
    std::vector<T> xr(size_t(nseq) * in_len);
    for (auto& e : x) e = {T(nd(rng)), T(nd(rng))};
    for (auto& e : xr) e = T(nd(rng));
    std::vector<cx<T>> tab;
    blue_make_tables<T>(n, mb, tab);
    const cx<T>*w = tab.data(), *bf = tab.data() + n;
    BlueIn<T> in{real_in ? (const void*)xr.data() : (const void*)x.data(), in_len, 1, AxisMap{n, in_len, in_off, in_shift}, conj_in ? 1 : 0,
                 real_in ? 1 : 0};
    std::vector<cx<T>> a(size_t(nseq) * n), b(size_t(nseq) * mb), out(size_t(nseq) * n, cx<T>{T(-7), T(-7)});
    for (int s = 0; s < nseq; ++s)
        for (int j = 0; j < n; ++j) a[size_t(s) * n + j] = cmul(blue_fetch(in, s, j), w[j]);
    auto tw = make_tw<T>(mb);
    const int nblk = (nseq + C::BO - 1) / C::BO;
    {
        RowLoadNat<T> lp{a.data(), n, AxisMap{mb, n, 0, 0}, nseq, 0, 0};
        RowStoreNat<T> sp{b.data(), mb, AxisMap{mb, mb, 0, 0}, nseq, 0, T(1), 0, AxisMap{1, 1, 0, 0}};
        emu_kernel<C, false>(nblk, lp, sp, tw.data());
    }
    for (int s = 0; s < nseq; ++s)
        for (int k = 0; k < mb; ++k) b[size_t(s) * mb + k] = cmul(b[size_t(s) * mb + k], bf[k]);
    {
        RowLoadNat<T> lp{b.data(), mb, AxisMap{mb, mb, 0, 0}, nseq, 1, 0};
        RowStoreNat<T> sp{out.data(), n, AxisMap{mb, n, 0, 0}, nseq, 1, T(1), 0, AxisMap{1, 1, 0, 0}};
        emu_kernel<C, false>(nblk, lp, sp, tw.data());
    }
    for (int s = 0; s < nseq; ++s)
        for (int k = 0; k < n; ++k) out[size_t(s) * n + k] = cmul(out[size_t(s) * n + k], w[k]);
    const ld pi = acosl(-1.0L);
    double err = 0, nrm = 0;
    for (int s = 0; s < nseq; ++s) {
        std::vector<cld> p(n, cld(0, 0));
        for (int i = 0; i < n; ++i) {
            int pp = (i + in_shift) % n, q = pp - in_off;
            if (q >= 0 && q < in_len) {
                p[i] = real_in ? cld(xr[size_t(s) * in_len + q], 0) : cld(x[size_t(s) * in_len + q].x, x[size_t(s) * in_len + q].y);
                if (conj_in) p[i] = std::conj(p[i]);
            }
        }
        for (int k = 0; k < n; ++k) {
            cld acc(0, 0);
            for (int j = 0; j < n; ++j) {
                ld ang = -2 * pi * ld((int64_t(j) * k) % n) / n;
                acc += p[j] * cld(cosl(ang), sinl(ang));
            }
            cx<T> got = out[size_t(s) * n + k];
            err = fmax(err, (double)std::abs(acc - cld(got.x, got.y)));
            nrm = fmax(nrm, (double)std::abs(acc));
        }
    }
    char buf[128];
    snprintf(buf, sizeof buf, "bluestein rows %s n=%d MB=%d nseq=%d len=%d off=%d sh=%d conj=%d real=%d", sizeof(T) == 4 ? "c64" : "c128", n,
             mb, nseq, in_len, in_off, in_shift, int(conj_in), int(real_in));
    report(buf, err / nrm, sizeof(T) == 4 ? 3e-6 : 1e-14);
}

// natural column pass by contract: logical sequence p[i] = (ay_in.map(i) >= 0 ? src[map][c] : 0), conj-in, DFT, conj-out, scale,
// bin k stored at row ay_out.map(k)
template <typename T>
static void naive_col_pass(const std::vector<cx<T>>& src, AxisMap ain, std::vector<cx<T>>& dst, AxisMap aout, int ncols, bool inv) {
    const int N = ain.n;
    const ld pi = acosl(-1.0L);
    for (int c = 0; c < ncols; ++c) {
        std::vector<cld> p(N, cld(0, 0));
        for (int i = 0; i < N; ++i) {
            const int q = ain.map(i);
            if (q >= 0) p[i] = cld(src[size_t(q) * ncols + c].x, src[size_t(q) * ncols + c].y);
        }
        for (int k = 0; k < N; ++k) {
            const int q = aout.map(k);
            if (q < 0) continue;
            cld acc(0, 0);
            for (int j = 0; j < N; ++j) {
                ld ang = (inv ? 2 : -2) * pi * ld((int64_t(j) * k) % N) / N;
                acc += p[j] * cld(cosl(ang), sinl(ang));
            }
            dst[size_t(q) * ncols + c] = {T(acc.real()), T(acc.imag())};
        }
    }
}

template <typename T>
static void test_blue_cols(int n, int ncols, int in_rows, int in_off, int in_shift, int out_len, int out_off, int out_shift, int epilogue) {
    const int mb = blue_conv_len(n);
    std::mt19937 rng(n * 11 + ncols);
    std::normal_distribution<double> nd;
    std::vector<cx<T>> W(size_t(in_rows) * ncols);
    for (auto& e : W) e = {T(nd(rng)), T(nd(rng))};
    std::vector<cx<T>> tab;
    blue_make_tables<T>(n, mb, tab);
    const cx<T>*w = tab.data(), *bf = tab.data() + n;
    BlueIn<T> in{W.data(), 1, ncols, AxisMap{n, in_rows, in_off, in_shift}, 0, 0};
    std::vector<cx<T>> a(size_t(n) * ncols), b(size_t(mb) * ncols);
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < ncols; ++c) a[size_t(i) * ncols + c] = cmul(blue_fetch(in, c, i), w[i]);
    naive_col_pass<T>(a, AxisMap{mb, n, 0, 0}, b, AxisMap{mb, mb, 0, 0}, ncols, false);
    for (int k = 0; k < mb; ++k)
        for (int c = 0; c < ncols; ++c) b[size_t(k) * ncols + c] = cmul(b[size_t(k) * ncols + c], bf[k]);
    naive_col_pass<T>(b, AxisMap{mb, mb, 0, 0}, a, AxisMap{mb, n, 0, 0}, ncols, true);
    // epilogue: scale 0.5, conj, crop / rotate rows, |.|^2 optional
    std::vector<cx<T>> oc(size_t(out_len) * ncols, cx<T>{T(-7), T(-7)});
    std::vector<T> orl(size_t(out_len) * ncols, T(-7));
    ColStoreNat<T> cs{};
    cs.dst = epilogue ? (void*)orl.data() : (void*)oc.data();
    cs.ld = ncols;
    cs.ay = AxisMap{n, out_len, out_off, out_shift};
    cs.ax = AxisMap{ncols, ncols, 0, 0};
    cs.conj = 1;
    cs.epilogue = epilogue;
    cs.scale = T(0.5);
    cs.weight = T(1);
    cs.mul_kind = MUL_NONE;
    for (int k = 0; k < n; ++k)
        for (int c = 0; c < ncols; ++c) store_one(cs, k, c, cmul(a[size_t(k) * ncols + c], w[k]));
    const ld pi = acosl(-1.0L);
    double err = 0, nrm = 0;
    for (int c = 0; c < ncols; ++c) {
        std::vector<cld> p(n, cld(0, 0));
        for (int i = 0; i < n; ++i) {
            int pp = (i + in_shift) % n, q = pp - in_off;
            if (q >= 0 && q < in_rows) p[i] = cld(W[size_t(q) * ncols + c].x, W[size_t(q) * ncols + c].y);
        }
        for (int k = 0; k < n; ++k) {
            int q = (k + out_shift) % n - out_off;
            if (q < 0 || q >= out_len) continue;
            cld acc(0, 0);
            for (int j = 0; j < n; ++j) {
                ld ang = -2 * pi * ld((int64_t(j) * k) % n) / n;
                acc += p[j] * cld(cosl(ang), sinl(ang));
            }
            acc = std::conj(acc * ld(0.5));
            if (epilogue) {
                err = fmax(err, fabs(double(std::norm(acc)) - double(orl[size_t(q) * ncols + c])));
                nrm = fmax(nrm, double(std::norm(acc)));
            } else {
                cx<T> got = oc[size_t(q) * ncols + c];
                err = fmax(err, (double)std::abs(acc - cld(got.x, got.y)));
                nrm = fmax(nrm, (double)std::abs(acc));
            }
        }
    }
    char buf[128];
    snprintf(buf, sizeof buf, "bluestein cols %s n=%d MB=%d ncols=%d rows=%d off=%d sh=%d out=%d/%d/%d epi=%d", sizeof(T) == 4 ? "c64" : "c128",
             n, mb, ncols, in_rows, in_off, in_shift, out_len, out_off, out_shift, epilogue);
    report(buf, err / nrm, sizeof(T) == 4 ? 3e-6 : 1e-13);
}

// --- both-axes Bluestein chain with the chirp multiplies inside the first load (RowLoadChirp) and the last store (RowStoreChirp):
// the three emulated engine passes of blue2d_fused_run (capi.hip) against a naive n1 x n2 DFT of the caller's view, epilogue included
template <typename T, int LOGM, int LOGN, int RBO, int RE, int CCI, int CE, int CBO, int CCOMP>
static void test_blue2d_fused(int n1, int n2, int in_rows, int in_cols, int offy, int offx, int shy, int shx, bool real_in, bool inverse,
                              int out_rows, int out_cols, int ooffy, int ooffx, int oshy, int oshx, int epilogue, int log_k) {
    using RC = FftCfg<T, LOGN, 1, RE, RBO, 1>;
    using CC = FftCfg<T, LOGM, CCI, CE, CBO, CCOMP>;
    const int M = CC::N, N = RC::N, TC = CCI * CE, TL = TC << log_k;
    if (M != blue_conv_len(n1) || N != blue_conv_len(n2)) { printf("blue2d fused: convolution length mismatch\n"); ++g_fail; return; }
    int ltl = 0;
    while ((1 << ltl) < TL) ++ltl;
    std::mt19937 rng(n1 * 31 + n2);
    std::normal_distribution<double> nd;
    std::vector<cx<T>> x(size_t(in_rows) * in_cols);
    std::vector<T> xr(size_t(in_rows) * in_cols);
    for (auto& e : x) e = {T(nd(rng)), T(nd(rng))};
    for (auto& e : xr) e = T(nd(rng));
    std::vector<cx<T>> t1, t2;
    blue_make_tables<T>(n1, M, t1);
    blue_make_tables<T>(n2, N, t2);
    const int ntl = (N + TL - 1) / TL, ntiles = (N + TC - 1) / TC;
    std::vector<cx<T>> W1(size_t(ntl) * n1 * TL, cx<T>{T(1e30), T(1e30)}), W2(size_t(ntl) * n1 * TL, cx<T>{T(1e30), T(1e30)});
    auto twN = make_tw<T>(N);
    auto twM = make_tw<T>(M);
    Blue2dIn<T> view{real_in ? (const void*)xr.data() : (const void*)x.data(), in_cols, AxisMap{n1, in_rows, offy, shy},
                     AxisMap{n2, in_cols, offx, shx}, inverse ? 1 : 0, real_in ? 1 : 0};
    RowLoadChirp<T> lp{view, t1.data(), t2.data(), n1};
    RowStoreTiled<T> sp{W1.data(), n1, ltl};
    emu_kernel<RC, false>((n1 + RC::BO * RE - 1) / (RC::BO * RE), lp, sp, twN.data());
    ColLoadTiled<T> cl{W1.data(), n1, AxisMap{M, n1, 0, 0}, ntiles, log_k};
    MidMul<T> mm{MUL_SEPARABLE, 0, t1.data() + n1, t2.data() + n2, 0, N};
    ColStoreTiledCrop<T> cst{W2.data(), n1, ntiles, log_k};
    {
        std::vector<Regs<CC>> regs(CC::NT);
        std::vector<typename LdsType<CC>::type> lds(CC::LDS_ELEMS + 1);
        const int ngroups = (ntiles + CC::BO - 1) / CC::BO;
        for (int g = 0; g < ngroups; ++g) {
            for (int tid = 0; tid < CC::NT; ++tid) {
                ThreadPos pos = thread_pos<CC>(tid);
                load<CC>(cl, g * CC::BO + pos.bo, pos, regs[tid].v);
            }
            emu_stages<CC, 0>(regs, lds, twM.data());
            for (int tid = 0; tid < CC::NT; ++tid) {
                ThreadPos pos = thread_pos<CC>(tid);
                mid_multiply_conj<CC>(mm, g * CC::BO + pos.bo, pos, regs[tid].v);
            }
            emu_stages<CC, 0>(regs, lds, twM.data());
            for (int tid = 0; tid < CC::NT; ++tid) {
                ThreadPos pos = thread_pos<CC>(tid);
                for (int e = 0; e < CC::E; ++e)
                    for (int m = 0; m < CC::P; ++m) regs[tid].v[e][m].y = -regs[tid].v[e][m].y;
                store<CC>(cst, g * CC::BO + pos.bo, pos, regs[tid].v);
            }
        }
    }
    std::vector<cx<T>> oc(size_t(out_rows) * out_cols, cx<T>{T(-7), T(-7)});
    std::vector<T> orl(size_t(out_rows) * out_cols, T(-7));
    ColStoreNat<T> cs{};
    cs.dst = epilogue ? (void*)orl.data() : (void*)oc.data();
    cs.ld = out_cols;
    cs.ay = AxisMap{n1, out_rows, ooffy, oshy};
    cs.ax = AxisMap{n2, out_cols, ooffx, oshx};
    cs.conj = inverse ? 1 : 0;
    cs.epilogue = epilogue;
    cs.scale = T(0.25);
    cs.weight = T(1);
    cs.mul_kind = MUL_NONE;
    RowLoadTiled<T> rl{W2.data(), n1, ltl, 0, n1, 1};
    RowStoreChirp<T> rs{cs, t1.data(), t2.data(), n1, n2, 1};
    emu_kernel<RC, false>((n1 + RC::BO * RE - 1) / (RC::BO * RE), rl, rs, twN.data());
    // truth: separable naive DFT of the logical n1 x n2 array (conj-in / conj-out for the inverse), scale 0.25
    const ld pi = acosl(-1.0L);
    std::vector<cld> L(size_t(n1) * n2), R(size_t(n1) * n2), F(size_t(n1) * n2);
    for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n2; ++j) {
            const cx<T> v = fetch2d(view, i, j);
            L[size_t(i) * n2 + j] = cld(v.x, v.y);
        }
    for (int i = 0; i < n1; ++i)
        for (int k = 0; k < n2; ++k) {
            cld acc(0, 0);
            for (int j = 0; j < n2; ++j) acc += L[size_t(i) * n2 + j] * std::polar(ld(1), -2 * pi * ld((int64_t(j) * k) % n2) / n2);
            R[size_t(i) * n2 + k] = acc;
        }
    for (int k = 0; k < n1; ++k)
        for (int c = 0; c < n2; ++c) {
            cld acc(0, 0);
            for (int i = 0; i < n1; ++i) acc += R[size_t(i) * n2 + c] * std::polar(ld(1), -2 * pi * ld((int64_t(i) * k) % n1) / n1);
            F[size_t(k) * n2 + c] = acc;
        }
    double err = 0, nrm = 0;
    for (int k = 0; k < n1; ++k)
        for (int c = 0; c < n2; ++c) {
            const int qy = cs.ay.map(k), qx = cs.ax.map(c);
            if (qy < 0 || qx < 0) continue;
            cld ref = F[size_t(k) * n2 + c] * ld(0.25);
            if (inverse) ref = std::conj(ref);
            if (epilogue) {
                err = fmax(err, fabs(double(std::norm(ref)) - double(orl[size_t(qy) * out_cols + qx])));
                nrm = fmax(nrm, double(std::norm(ref)));
            } else {
                const cx<T> got = oc[size_t(qy) * out_cols + qx];
                err = fmax(err, (double)std::abs(ref - cld(got.x, got.y)));
                nrm = fmax(nrm, (double)std::abs(ref));
            }
        }
    char buf[200];
    snprintf(buf, sizeof buf, "bluestein 2-D fused %s %dx%d (conv %dx%d) in=%dx%d@%d,%d sh=%d,%d real=%d inv=%d out=%dx%d epi=%d E=%d", sizeof(T) == 4 ? "c64" : "c128",
             n1, n2, M, N, in_rows, in_cols, offy, offx, shy, shx, int(real_in), int(inverse), out_rows, out_cols, epilogue, RE);
    report(buf, err / nrm, sizeof(T) == 4 ? 4e-6 : 1e-13);
}

// (the emulation of the grouped-wavelength kernels of round 4 went with fft_spectral2.h: experiments/README.md)

int main() {
    test_blue2d_fused<float, 6, 7, 32, 1, 4, 2, 16, 1>(20, 50, 20, 50, 0, 0, 10, 25, false, false, 20, 50, 0, 0, 10, 25, 0, 1);      // focus-like: both rotations
    test_blue2d_fused<float, 7, 6, 64, 2, 4, 2, 8, 1>(50, 24, 25, 12, 13, 6, 25, 12, false, true, 30, 20, 10, 2, 25, 12, 0, 0);     // Q = 2 pad, inverse, crop, 2 rows / thread
    test_blue2d_fused<float, 6, 6, 64, 1, 4, 2, 16, 1>(30, 30, 30, 30, 0, 0, 0, 0, true, false, 30, 30, 0, 0, 15, 15, 1, 1);          // real input, |.|^2
    test_blue2d_fused<double, 6, 7, 32, 1, 4, 1, 16, 2>(24, 40, 24, 40, 0, 0, 12, 20, false, false, 24, 40, 0, 0, 12, 20, 0, 2);
    test_blue2d_fused<double, 7, 5, 128, 2, 4, 1, 8, 2>(40, 12, 20, 12, 10, 0, 20, 6, false, true, 40, 12, 0, 0, 0, 0, 0, 1);
    test_blue_rows<float, 8, 16>(100, 5, 100, 0, 0, false, false);
    test_blue_rows<float, 8, 16>(127, 3, 60, 34, 63, true, false);      // padded (Q ~ 2) + ifftshift rotation + inverse
    test_blue_rows<float, 9, 8>(129, 4, 129, 0, 64, false, true);       // real input, MB = 512
    test_blue_rows<double, 8, 16>(97, 3, 97, 0, 48, false, false);      // prime length
    test_blue_rows<double, 11, 2>(1000, 2, 500, 250, 500, true, false);
    test_blue_cols<float>(100, 6, 100, 0, 50, 100, 0, 50, 0);
    test_blue_cols<float>(97, 5, 40, 29, 48, 60, 18, 48, 1);            // padded rows, crop, |.|^2
    test_blue_cols<double>(150, 3, 150, 0, 0, 150, 0, 75, 0);
    // rows: every stage structure (P<16, single stage, 16x2, 16x16, 16x16x8, 16^3, 16^3x2)
    test_row<float, 1, 256, 1>(300, 2, 0, 0, 0, false);
    test_row<float, 3, 256, 1>(5, 8, 0, 0, 0, false);
    test_row<float, 4, 256, 1>(3, 16, 0, 0, 0, false);
    test_row<float, 5, 128, 1>(3, 32, 0, 0, 0, false);
    test_row<float, 6, 64, 1>(70, 64, 0, 32, 32, false);
    test_row<float, 7, 32, 1>(3, 100, 14, 64, 0, true);
    test_row<float, 8, 16, 1>(17, 256, 0, 0, 128, false);
    test_row<float, 9, 8, 1>(9, 500, 6, 256, 256, false);
    test_row<float, 10, 4, 1>(5, 1024, 0, 0, 0, true);
    test_row<float, 11, 2, 1>(3, 2048, 0, 1024, 1024, false);
    test_row<float, 12, 1, 1>(2, 4096, 0, 0, 0, false);
    test_row<float, 13, 1, 1>(1, 8192, 0, 0, 0, false);
    test_row<double, 8, 16, 1>(5, 256, 0, 128, 128, false);
    test_row<double, 12, 1, 2>(2, 4096, 0, 0, 0, true);
    test_row<double, 10, 4, 2>(5, 700, 162, 512, 0, false);
    // eight points per thread (radix-8 stages: round 4's experiment engine): 8^4, 8^3 x 4, 8^2 x 2, two rows per thread, rotations
    test_row<float, 12, 1, 1, 2, 3>(3, 4096, 0, 2048, 2048, false);
    test_row<float, 11, 1, 1, 2, 3>(5, 2000, 24, 1024, 0, true);
    test_row<float, 7, 4, 1, 1, 3>(9, 128, 0, 64, 64, false);
    test_row<double, 10, 1, 2, 1, 3>(3, 1024, 0, 0, 512, false);
    // two rows per thread (row pass variant 4): odd row counts leave a half-filled last pair
    test_row<float, 12, 1, 1, 2>(3, 4096, 0, 2048, 2048, false);
    test_row<float, 11, 2, 1, 2>(7, 2000, 24, 1024, 0, true);
    test_row<double, 11, 2, 1, 2>(5, 2048, 0, 0, 1024, false);
    test_row<float, 6, 64, 1, 2>(131, 64, 0, 32, 32, false);
    // 2-D through the tiled intermediate
    test_2d<float, 6, 5, 128, 1, 4, 2, 16, 1>(64, 32, true, 0, false, 64, 32);
    test_2d<float, 5, 6, 64, 1, 4, 2, 32, 1>(16, 32, true, 0, false, 32, 64);      // Q=2 pad
    test_2d<float, 5, 6, 64, 1, 4, 2, 32, 1>(32, 64, true, 0, true, 16, 32);       // adjoint: inverse + crop
    test_2d<float, 7, 7, 32, 1, 4, 2, 8, 1>(128, 128, true, 1, false, 128, 128);   // abs2 epilogue
    test_2d<float, 8, 5, 128, 1, 4, 2, 4, 1>(200, 20, false, 0, false, 256, 32);
    test_2d<float, 9, 4, 256, 1, 2, 2, 4, 1>(512, 16, true, 0, false, 512, 16);    // TC=4 variant
    test_2d<float, 6, 6, 64, 1, 4, 2, 16, 1>(64, 64, true, 0, false, 64, 64, 1);   // layout tile 16 wide
    test_2d<float, 5, 7, 32, 1, 4, 2, 32, 1>(16, 100, true, 0, false, 32, 128, 2);  // layout tile 32 wide + pad
    test_2d<double, 6, 6, 64, 1, 4, 1, 16, 2>(64, 64, true, 0, true, 40, 64, 2);
    test_2d<double, 6, 6, 64, 1, 4, 1, 16, 2>(64, 64, true, 0, false, 64, 64);
    test_2d<double, 5, 7, 32, 2, 4, 1, 32, 1>(20, 100, true, 0, true, 32, 128);
    test_2d<double, 8, 4, 256, 1, 2, 1, 8, 2>(256, 16, false, 1, false, 256, 16);
    test_2d_fold<float, 6, 5, 128, 1, 4, 2, 32, 1>(32, true, 0, false, 64, 32, 1);
    test_2d_fold<float, 5, 6, 64, 1, 4, 2, 64, 1>(50, false, 0, false, 32, 64);             // padded columns, no rotation
    test_2d_fold<float, 6, 6, 64, 1, 4, 2, 32, 1>(64, true, 1, false, 64, 64, 2);            // |.|^2 epilogue
    test_2d_fold<float, 6, 5, 128, 1, 4, 2, 32, 1>(32, true, 0, true, 32, 16);               // inverse + even crop
    test_2d_fold<double, 6, 6, 64, 1, 4, 1, 32, 2>(64, true, 0, false, 64, 64, 2);
    test_2d_fold<double, 7, 5, 128, 2, 4, 1, 16, 2>(20, true, 0, true, 128, 32, 1);
    test_fused<float, 5, 6, 64, 4, 2, 32, 1>(32, 64, 32, 64, 1, true, false, false);
    test_fused<float, 6, 5, 128, 4, 2, 16, 1>(40, 20, 64, 32, 2, false, false, false);   // padded input
    test_fused<float, 5, 6, 64, 4, 2, 32, 1>(32, 64, 16, 32, 0, true, true, false);      // adjoint: conj(H) + crop
    test_fused<double, 6, 6, 64, 4, 1, 16, 2>(64, 64, 64, 64, 2, true, false, true);      // with rotations (conv)
    test_fused<double, 5, 7, 32, 4, 1, 32, 1>(20, 100, 32, 128, 1, false, true, false);
    test_fused_fold<float, 6, 5, 128, 1, 4, 2, 32, 1>(32, 64, 32, 1, true, false, false);
    test_fused_fold<float, 5, 6, 64, 1, 4, 2, 64, 1>(40, 32, 64, 0, false, false, true);       // full H, rotations (conv), padded columns
    test_fused_fold<float, 6, 5, 128, 1, 4, 2, 32, 1>(32, 31, 16, 2, true, true, false);        // conj(H) + odd crop of the rows
    test_fused_fold<double, 6, 6, 64, 1, 4, 1, 32, 2>(64, 64, 64, 2, true, false, true);
    test_fused_fold<double, 7, 5, 128, 2, 4, 1, 16, 2>(20, 128, 32, 1, false, true, false);
    printf(g_fail ? "EMU FAILED (%d)\n" : "EMU OK\n", g_fail);
    return g_fail ? 1 : 0;
}
