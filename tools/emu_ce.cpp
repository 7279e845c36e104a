// CPU emulation of the composite register engine's per-thread logic (prysm_amd/csrc/fft_ce.h), test scaffolding like emu_fft.cpp /
// emu_mix.cpp: every thread of a workgroup runs each phase in turn with a std::vector standing in for LDS, against a naive long-double DFT.
// build: g++ -O2 -std=c++17 -I prysm_amd/csrc -I tools tools/emu_ce.cpp -o /tmp/emu_ce
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "fft_io.h"
#include "fft_ce.h"

using namespace pm;
typedef long double ld;
typedef std::complex<ld> cld;

template <typename C, int s>
static void run_stages(std::vector<cx<typename C::T>>& regs, std::vector<typename CeLds<C>::type>& lds, const cx<typename C::T>* tw) {
    using T = typename C::T;
    using LT = typename CeLds<C>::type;
    auto V = [&](int tid) -> cx<T>(&)[C::P] { return *reinterpret_cast<cx<T>(*)[C::P]>(regs.data() + size_t(tid) * C::P); };
    if constexpr (s > 0) {
        for (int comp = 0; comp < C::COMP; ++comp) {
            for (int tid = 0; tid < C::NT; ++tid) ce_exch_write<C, s, LT>(V(tid), comp, ce_pos<C>(tid), lds.data());
            for (int tid = 0; tid < C::NT; ++tid) ce_exch_read<C, s, LT>(V(tid), comp, ce_pos<C>(tid), lds.data());
        }
    }
    for (int tid = 0; tid < C::NT; ++tid) ce_stage<C, s>(V(tid), ce_pos<C>(tid).t, tw);
    if constexpr (s + 1 < C::PL::S) run_stages<C, s + 1>(regs, lds, tw);
}

template <typename C>
static double run_case(int nseq, int shift, int off, int len) {
    using T = typename C::T;
    using PL = typename C::PL;
    constexpr int n = PL::N;
    const bool col = C::COL;
    const ld pi = acosl(-1.0L);
    std::vector<cx<T>> tw(n);
    for (int i = 0; i < n; ++i) tw[i] = {T(cosl(-2 * pi * i / n)), T(sinl(-2 * pi * i / n))};
    std::mt19937 rng(n * 7 + nseq);
    std::uniform_real_distribution<double> U(-1, 1);
    // rows: x[seq][q], q < len; cols: x[q][seq]
    std::vector<cx<T>> x(size_t(len) * nseq), y(size_t(n) * nseq, cx<T>{T(0), T(0)});
    for (auto& v : x) v = {T(U(rng)), T(U(rng))};
    CeIn<T> in{x.data(), col ? nseq : len, AxisMap{n, len, off, shift}, nseq, T(1)};
    const int sy = 3 % n, sx = nseq > 2 ? 2 : 0;
    CeRowOut<T> ro{y.data(), n, 0, AxisMap{n, n, 0, 0}, T(1), T(1)};
    CeColOut<T> co{y.data(), nseq, n, sy, nseq, sx, T(0.5), T(0.5), 0, T(1)};
    std::vector<typename CeLds<C>::type> lds(C::lds_elems() + 64);
    std::vector<cx<T>> regs(size_t(C::NT) * C::P);
    auto V = [&](int tid) -> cx<T>(&)[C::P] { return *reinterpret_cast<cx<T>(*)[C::P]>(regs.data() + size_t(tid) * C::P); };
    const bool win = !(off == 0 && len == n);
    for (int g = 0; g * C::SEQS < nseq; ++g) {
        for (int tid = 0; tid < C::NT; ++tid) {
            const CePos pos = ce_pos<C>(tid);
            const int seq0 = g * C::SEQS, slc = seq0 + pos.sl < nseq ? pos.sl : nseq - 1 - seq0;
            if (win) ce_load<C, true>(V(tid), in, seq0, slc, pos.t); else ce_load<C, false>(V(tid), in, seq0, slc, pos.t);
        }
        run_stages<C, 0>(regs, lds, tw.data());
        for (int tid = 0; tid < C::NT; ++tid) {
            const CePos pos = ce_pos<C>(tid);
            const int seq = g * C::SEQS + pos.sl;
            if (seq >= nseq) continue;
            if (col) ce_store_col<C>(V(tid), co, g * C::SEQS, pos.sl, pos.t); else ce_store_row<C>(V(tid), ro, g * C::SEQS, pos.sl, pos.t);
        }
    }
    double err = 0, ref = 0;
    for (int s = 0; s < nseq; ++s) {
        std::vector<cld> xs(n);
        for (int i = 0; i < n; ++i) {
            int p = i + shift; if (p >= n) p -= n;
            const int q = p - off;
            if (q < 0 || q >= len) { xs[i] = 0; continue; }
            const cx<T> v = col ? x[size_t(q) * nseq + s] : x[size_t(s) * len + q];
            xs[i] = cld(v.x, v.y);
        }
        for (int k = (s * 5) % 37; k < n; k += (n > 3000 ? 401 : (n > 600 ? 101 : 1))) {
            cld acc = 0;
            for (int i = 0; i < n; ++i) { const ld a = -2 * pi * ld((int64_t(i) * k) % n) / n; acc += xs[i] * cld(cosl(a), sinl(a)); }
            cx<T> v;
            if (col) {
                acc *= ld(0.5);
                v = y[size_t((k + sy) % n) * nseq + (s + sx) % nseq];
            } else {
                v = y[size_t(s) * n + k];
            }
            err = std::max(err, double(std::abs(acc - cld(v.x, v.y))));
            ref = std::max(ref, double(std::abs(acc)));
        }
    }
    return err / ref;
}

// pupil synthesis in the row loads: packed (amplitude, OPD) pairs (SYN 3) or an OPD map with a float amplitude array (SYN 2), a window + rotation
template <typename C, int SYN>
static double run_synth(int nseq) {
    using T = typename C::T;
    using PL = typename C::PL;
    constexpr int n = PL::N;
    const int len = n / 2 + 3, off = n / 4, shift = n / 2;
    const ld pi = acosl(-1.0L);
    const double k2 = 1.37;
    std::vector<cx<T>> tw(n);
    for (int i = 0; i < n; ++i) tw[i] = {T(cosl(-2 * pi * i / n)), T(sinl(-2 * pi * i / n))};
    std::mt19937 rng(n + SYN);
    std::uniform_real_distribution<double> U(-1, 1);
    std::vector<cx<T>> pk(size_t(len) * nseq), y(size_t(n) * nseq, cx<T>{T(0), T(0)});
    std::vector<T> opd(size_t(len) * nseq);
    std::vector<float> amp(size_t(len) * nseq);
    for (size_t i = 0; i < pk.size(); ++i) {
        amp[i] = float(0.5 + 0.5 * U(rng));
        opd[i] = T(3 * U(rng));
        pk[i] = {T(amp[i]), opd[i]};
    }
    CeIn<T> in{SYN == 3 ? pk.data() : reinterpret_cast<const cx<T>*>(opd.data()), len, AxisMap{n, len, off, shift}, nseq, T(1)};
    const CeSynth sy{SYN, k2, amp.data(), 1, len};
    CeRowOut<T> ro{y.data(), n, 0, AxisMap{n, n, 0, 0}, T(1), T(1)};
    std::vector<typename CeLds<C>::type> lds(C::lds_elems() + 64);
    std::vector<cx<T>> regs(size_t(C::NT) * C::P);
    auto V = [&](int tid) -> cx<T>(&)[C::P] { return *reinterpret_cast<cx<T>(*)[C::P]>(regs.data() + size_t(tid) * C::P); };
    for (int g = 0; g * C::SEQS < nseq; ++g) {
        for (int tid = 0; tid < C::NT; ++tid) {
            const CePos pos = ce_pos<C>(tid);
            const int seq0 = g * C::SEQS, slc = seq0 + pos.sl < nseq ? pos.sl : nseq - 1 - seq0;
            ce_load_synth<C, SYN>(V(tid), in, sy, seq0, slc, pos.t);
        }
        run_stages<C, 0>(regs, lds, tw.data());
        for (int tid = 0; tid < C::NT; ++tid) {
            const CePos pos = ce_pos<C>(tid);
            if (g * C::SEQS + pos.sl < nseq) ce_store_row<C>(V(tid), ro, g * C::SEQS, pos.sl, pos.t);
        }
    }
    double err = 0, ref = 0;
    for (int s = 0; s < nseq; ++s) {
        std::vector<cld> xs(n);
        for (int i = 0; i < n; ++i) {
            int p = i + shift; if (p >= n) p -= n;
            const int q = p - off;
            if (q < 0 || q >= len) { xs[i] = 0; continue; }
            const ld ang = 2 * pi * ld(k2) * ld(opd[size_t(s) * len + q]);
            xs[i] = ld(amp[size_t(s) * len + q]) * cld(cosl(ang), sinl(ang));
        }
        for (int k = s % 7; k < n; k += 37) {
            cld acc = 0;
            for (int i = 0; i < n; ++i) { const ld a = -2 * pi * ld((int64_t(i) * k) % n) / n; acc += xs[i] * cld(cosl(a), sinl(a)); }
            const cx<T> v = y[size_t(s) * n + k];
            err = std::max(err, double(std::abs(acc - cld(v.x, v.y))));
            ref = std::max(ref, double(std::abs(acc)));
        }
    }
    return err / ref;
}

static int fails = 0;
template <typename C>
static void check(const char* name) {
    using PL = typename C::PL;
    const double tol = sizeof(typename C::T) == 4 ? 3e-6 : 1e-14;
    const double e1 = run_case<C>(C::SEQS * 2, 0, 0, PL::N), e2 = run_case<C>(C::SEQS + 1, PL::N / 2, 0, PL::N), e3 = run_case<C>(3, 7, PL::N / 4, PL::N / 2);
    const bool ok = e1 < tol && e2 < tol && e3 < tol;
    printf("%-44s n %5d seqs %d %s comp %d  lds %6zu B  err %.2e %.2e %.2e  %s\n", name, PL::N, C::SEQS, C::COL ? "col" : "row", C::COMP, C::LDS_BYTES, e1, e2, e3, ok ? "ok" : "FAIL");
    if (!ok) ++fails;
}
#define CHECK(...) check<__VA_ARGS__>(#__VA_ARGS__)

int main() {
    CHECK(CeCfg<float, CePlan<30, 10, 10>, 5, false, 2, 22, 3>);
    CHECK(CeCfg<float, CePlan<30, 10, 10>, 4, true, 2, 24, 4>);
    CHECK(CeCfg<double, CePlan<30, 10, 10>, 2, false, 2, 1, 1>);
    CHECK(CeCfg<float, CePlan<10, 10, 10>, 8, false, 1, 10, 1>);
    CHECK(CeCfg<float, CePlan<10, 10, 10>, 8, true, 1>);
    CHECK(CeCfg<double, CePlan<20, 10, 10>, 4, true, 2, 24, 4>);
    CHECK(CeCfg<float, CePlan<20, 10, 10>, 4, false, 2>);
    CHECK(CeCfg<float, CePlan<30, 10, 5>, 8, false, 2, 53, 62>);
    CHECK(CeCfg<float, CePlan<30, 10, 5>, 3, true, 1>);
    CHECK(CeCfg<float, CePlan<20, 20, 10>, 2, false, 2, 58, 1>);
    CHECK(CeCfg<float, CePlan<24, 8, 8>, 4, true, 2>);
    CHECK(CeCfg<float, CePlan<24, 8, 4, 4>, 2, false, 2>);
    CHECK(CeCfg<double, CePlan<30, 10, 10, 2>, 1, false, 2>);
    CHECK(CeCfg<float, CePlan<20, 10, 5, 5>, 2, true, 2>);
    CHECK(CeCfg<float, CePlan<30, 30>, 4, false, 1>);
    CHECK(CeCfg<float, CePlan<20, 20, 20>, 1, false, 2>);
    CHECK(CeCfg<float, CePlan<12, 6, 4>, 4, true, 1>);
    CHECK(CeCfg<double, CePlan<15, 15, 5>, 2, false, 1>);
    // every shape the library ships (tools/ce_gen.py)
#include "emu_ce_plans.inc"
    {
        using CS = CeCfg<float, CePlan<10, 10, 10>, 2, false, 1, 2, 1>;
        const double e3 = run_synth<CS, 3>(5), e2 = run_synth<CS, 2>(3);
        const bool ok = e3 < 3e-6 && e2 < 3e-6;
        printf("pupil synthesis in the row loads (packed pairs, OPD + amplitude): err %.2e %.2e  %s\n", e3, e2, ok ? "ok" : "FAIL");
        if (!ok) ++fails;
    }
    printf(fails ? "FAILED %d\n" : "all ok\n", fails);
    return fails ? 1 : 0;
}
