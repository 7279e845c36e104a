// CPU emulation of the mixed-radix kernel's per-thread logic (prysm_amd/csrc/fft_mixed.h), test scaffolding like emu_fft.cpp: every
// thread of a workgroup runs each phase in turn with a std::vector standing in for LDS, against a naive long-double DFT.
// build: g++ -O2 -std=c++17 -I prysm_amd/csrc tools/emu_mix.cpp -o /tmp/emu_mix
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <type_traits>
#include <vector>

#include "fft_mixed.h"

using namespace pm;
typedef long double ld;
typedef std::complex<ld> cld;

template <typename T>
static double run_case(int n, int nseq, bool col, int seqs, int nt, int shift, int c0 = 0, int c1 = 0) {
    MixPlan p;
    if (!mix_make_plan(n, p)) return -1;
    MixShape sh{seqs, 0};
    while ((1 << sh.log_seqs) < seqs) ++sh.log_seqs;
    mix_shape_pads(p, sh, c0, c1);
    const ld pi = acosl(-1.0L);
    std::vector<cx<T>> tw(n);
    for (int i = 0; i < n; ++i) tw[i] = {T(cosl(-2 * pi * i / n)), T(sinl(-2 * pi * i / n))};
    std::mt19937 rng(n * 7 + nseq);
    std::uniform_real_distribution<double> U(-1, 1);
    // rows: x[seq][i]; cols: x[i][seq]
    std::vector<cx<T>> x(size_t(n) * nseq), y(size_t(n) * nseq, cx<T>{T(0), T(0)});
    for (auto& v : x) v = {T(U(rng)), T(U(rng))};
    BlueIn<T> in{x.data(), col ? 1 : n, col ? nseq : 1, AxisMap{n, n, 0, shift}, 0, 0};
    std::vector<cx<T>> lds(size_t(seqs) * sh.npad + 64);
    const int ngroups = (nseq + seqs - 1) / seqs;
    for (int g = 0; g < ngroups; ++g) {
        const int seq0 = g * seqs;
        const cx<T>* base0 = x.data() + (col ? seq0 : size_t(seq0) * n);
        const int nvalid = std::min(seqs, nseq - seq0);
        MixFetch<T, true, false> fc{base0, uint32_t(nseq), in.ax, T(1), nvalid};
        MixFetch<T, false, false> fr{base0, uint32_t(n), in.ax, T(1), nvalid};
        MixFetchWhole<T, true> wc{base0, uint32_t(nseq), n, shift, T(1)};
        MixFetchWhole<T, false> wr{base0, uint32_t(n), n, shift, T(1)};
        const bool whole = nvalid == seqs && (g % 2 == 0);     // alternate the two loaders over the groups
        auto fetch = [&](int sl, int i) { return whole ? (col ? wc.finish(wc(sl, i)) : wr.finish(wr(sl, i))) : (col ? fc(sl, i) : fr(sl, i)); };
        auto store = [&](int sl, int k, cx<T> v) {
            if (seq0 + sl >= nseq) return;
            if (col) y[size_t(k) * nseq + seq0 + sl] = v; else y[size_t(seq0 + sl) * n + k] = v;
        };
        auto run = [&](auto colc) {
            constexpr bool COL = decltype(colc)::value;
            if (COL && col && nvalid == seqs && (g % 4 == 1 || g % 4 == 2)) {
                // the persistent column kernel's way in: the tile copied to its LDS slots, then the first stage in place
                for (int i = 0; i < n; ++i)
                    for (int sl = 0; sl < seqs; ++sl) lds[(size_t(mix_slot_of(p, sh, i)) << sh.log_seqs) + sl] = wc.finish(wc(sl, i));
                for (int tid = 0; tid < nt; ++tid) mix_run_first_lds<T, COL, 20>(p, sh, tid, nt, lds.data(), tw.data());
            } else {
                for (int tid = 0; tid < nt; ++tid) mix_run_first<T, COL, 20, 3>(p, sh, tid, nt, lds.data(), tw.data(), fetch);
            }
            for (int ph = 1; ph + 1 < p.nstage; ++ph)
                for (int tid = 0; tid < nt; ++tid) mix_run_mid<T, COL, 20>(p, sh, ph, tid, nt, lds.data(), tw.data());
            for (int tid = 0; tid < nt; ++tid) mix_run_last<T, COL, 20>(p, sh, tid, nt, lds.data(), store);
        };
        // rows: the kernel interleaves the sequences of a workgroup in LDS (the column mode's layout and lane order) when their number is a
        // power of two -- what mix_rows_impl always launches; the [sequence][point] layout stays covered by the other counts
        if (col || (seqs & (seqs - 1)) == 0) run(std::true_type{}); else run(std::false_type{});
    }
    double err = 0, ref = 0;
    for (int s = 0; s < nseq; ++s) {
        std::vector<cld> xs(n);
        for (int i = 0; i < n; ++i) {
            int q = i + shift; if (q >= n) q -= n;
            const cx<T> v = col ? x[size_t(q) * nseq + s] : x[size_t(s) * n + q];
            xs[i] = cld(v.x, v.y);
        }
        for (int k = 0; k < n; k += (n > 600 ? 37 : 1)) {
            cld acc = 0;
            for (int i = 0; i < n; ++i) { const ld a = -2 * pi * ld((int64_t(i) * k) % n) / n; acc += xs[i] * cld(cosl(a), sinl(a)); }
            const cx<T> v = col ? y[size_t(k) * nseq + s] : y[size_t(s) * n + k];
            err = std::max(err, double(std::abs(acc - cld(v.x, v.y))));
            ref = std::max(ref, double(std::abs(acc)));
        }
    }
    return err / ref;
}

// the middle pass of fft2 -> x H -> ifft2 on a composite column length: forward stages, multiplier between the two small DFTs of the
// last stage, transposed stages, conjugate on the way out == the unnormalised inverse transform of (fft(x) h) per column
template <typename T>
static double run_mid_case(int n, int nseq, bool col, int seqs, int nt, int c0 = 0, int c1 = 0) {
    MixPlan p;
    if (!mix_make_plan(n, p)) return -1;
    MixShape sh{seqs, 0};
    while ((1 << sh.log_seqs) < seqs) ++sh.log_seqs;
    mix_shape_pads(p, sh, c0, c1);
    const ld pi = acosl(-1.0L);
    std::vector<cx<T>> tw(n);
    for (int i = 0; i < n; ++i) tw[i] = {T(cosl(-2 * pi * i / n)), T(sinl(-2 * pi * i / n))};
    std::mt19937 rng(n * 5 + nseq);
    std::uniform_real_distribution<double> U(-1, 1);
    std::vector<cx<T>> x(size_t(n) * nseq), y(size_t(n) * nseq, cx<T>{T(0), T(0)}), h(size_t(n) * nseq);
    for (auto& v : x) v = {T(U(rng)), T(U(rng))};
    for (auto& v : h) v = {T(U(rng)), T(U(rng))};       // h[bin][seq] (cols) or h[seq][bin] (rows)
    std::vector<cx<T>> lds(size_t(seqs) * sh.npad + 64);
    const int ngroups = (nseq + seqs - 1) / seqs;
    for (int g = 0; g < ngroups; ++g) {
        const int seq0 = g * seqs;
        const cx<T>* base0 = x.data() + (col ? seq0 : size_t(seq0) * n);
        const int nvalid = std::min(seqs, nseq - seq0);
        MixFetch<T, true, false> fc{base0, uint32_t(nseq), AxisMap{n, n, 0, 0}, T(1), nvalid};
        MixFetch<T, false, false> fr{base0, uint32_t(n), AxisMap{n, n, 0, 0}, T(1), nvalid};
        auto fetch = [&](int sl, int i) { return col ? fc(sl, i) : fr(sl, i); };
        auto mul = [&](int sl, int k, cx<T> v) {
            const int sq = seq0 + sl < nseq ? seq0 + sl : 0;
            const cx<T> hh = col ? h[size_t(k) * nseq + sq] : h[size_t(sq) * n + k];
            const cx<T> r = cmul(v, hh);
            return cx<T>{r.x, -r.y};
        };
        auto store = [&](int sl, int k, cx<T> v) {
            if (seq0 + sl >= nseq) return;
            v.y = -v.y;
            if (col) y[size_t(k) * nseq + seq0 + sl] = v; else y[size_t(seq0 + sl) * n + k] = v;
        };
        auto run = [&](auto colc) {
            constexpr bool COL = decltype(colc)::value;
            for (int tid = 0; tid < nt; ++tid) mix_run_first<T, COL, 20>(p, sh, tid, nt, lds.data(), tw.data(), fetch);
            for (int ph = 1; ph + 1 < p.nstage; ++ph)
                for (int tid = 0; tid < nt; ++tid) mix_run_mid<T, COL, 20>(p, sh, ph, tid, nt, lds.data(), tw.data());
            for (int tid = 0; tid < nt; ++tid) mix_run_last_mul<T, COL, 20>(p, sh, tid, nt, lds.data(), mul);
            for (int ph = p.nstage - 2; ph >= 1; --ph)
                for (int tid = 0; tid < nt; ++tid) mix_run_mid_t<T, COL, 20>(p, sh, ph, tid, nt, lds.data(), tw.data());
            for (int tid = 0; tid < nt; ++tid) mix_run_first_t<T, COL, 20>(p, sh, tid, nt, lds.data(), tw.data(), store);
        };
        if (col) run(std::true_type{}); else run(std::false_type{});
    }
    double err = 0, ref = 0;
    for (int s = 0; s < nseq; ++s) {
        std::vector<cld> X(n), Y(n);
        for (int k = 0; k < n; ++k) {
            cld acc = 0;
            for (int i = 0; i < n; ++i) {
                const cx<T> v = col ? x[size_t(i) * nseq + s] : x[size_t(s) * n + i];
                const ld a = -2 * pi * ld((int64_t(i) * k) % n) / n;
                acc += cld(v.x, v.y) * cld(cosl(a), sinl(a));
            }
            const cx<T> hh = col ? h[size_t(k) * nseq + s] : h[size_t(s) * n + k];
            X[k] = acc * cld(hh.x, hh.y);
        }
        for (int i = 0; i < n; i += (n > 300 ? 29 : 1)) {
            cld acc = 0;
            for (int k = 0; k < n; ++k) { const ld a = 2 * pi * ld((int64_t(i) * k) % n) / n; acc += X[k] * cld(cosl(a), sinl(a)); }
            const cx<T> v = col ? y[size_t(i) * nseq + s] : y[size_t(s) * n + i];
            err = std::max(err, double(std::abs(acc - cld(v.x, v.y))));
            ref = std::max(ref, double(std::abs(acc)));
        }
    }
    return err / ref;
}

int main() {
    int bad = 0;
    for (int n : {36, 60, 100, 125, 180, 243, 360, 1000, 1001, 1536, 2310, 3000}) {
        // (the second and third with padded LDS slots: MixShape pad0 / pad1)
        const double m1 = run_mid_case<double>(n, 5, true, 4, 96), m2 = run_mid_case<double>(n, 3, false, 2, 64, 3, 2), m3 = run_mid_case<float>(n, 6, true, 2, 128, 1, 4);
        printf("middle pass n=%5d  cols f64 %.2e  rows f64 %.2e  cols f32 %.2e\n", n, m1, m2, m3);
        if (!(m1 < 1e-12) || !(m2 < 1e-12) || !(m3 < 2e-5)) { ++bad; printf("   ^^^ FAIL\n"); }
    }
    const int lens[] = {36, 40, 42, 44, 48, 50, 52, 54, 56, 60, 64, 45, 49, 60, 77, 90, 96, 100, 120, 121, 125, 143, 144, 169, 180, 243, 250, 256, 343, 360, 500, 625, 729, 1000,
                        1001, 1331, 1500, 272, 323, 380, 1020, 1900, 4913, 6137, 2187, 2310, 2592, 3000, 3125, 4000, 4004, 4096, 5000, 6000, 6561, 7000, 8000, 324, 400, 441, 484, 576, 625, 676, 729, 784, 900, 1024, 660, 780, 810, 840, 960};
    for (int n : lens) {
        MixPlan p;
        if (!mix_make_plan(n, p)) { printf("n=%d no plan\n", n); ++bad; continue; }
        printf("n=%5d stages", n);
        for (int s = 0; s < p.nstage; ++s) printf(" %d", p.radix[s]);
        const double e1 = run_case<double>(n, 3, false, 2, 64, 0, n % 7, n % 3);      // padded LDS slots, pads varying with the length
        const double e2 = run_case<double>(n, 5, true, 4, 96, n / 2, n % 5, n % 4);
        const double e3 = run_case<float>(n, 4, false, 3, 128, 1);      // three rows per workgroup: the [sequence][point] layout
        printf("  rows f64 %.2e  cols f64 %.2e  rows f32 %.2e\n", e1, e2, e3);
        if (!(e1 < 1e-13) || !(e2 < 1e-13) || !(e3 < 2e-5)) { ++bad; printf("   ^^^ FAIL\n"); }
    }
    // the planner over every length: a plan exists exactly for the lengths whose primes are <= 19 and that need at least two factors; its
    // factors multiply to n, are ascending, at most 20, and the block lengths / reciprocals are consistent
    int planned = 0;
    for (int n = 2; n <= kMixMaxN; ++n) {
        int m = n;
        for (int pr : {2, 3, 5, 7, 11, 13, 17, 19})
            while (m % pr == 0) m /= pr;
        const bool smooth = m == 1;
        MixPlan pl;
        const bool ok = mix_make_plan(n, pl);
        if (ok != (smooth && !mix_radix_ok(n))) { printf("planner: n=%d smooth=%d planned=%d\n", n, int(smooth), int(ok)); ++bad; continue; }
        if (!ok) continue;
        ++planned;
        long prod = 1;
        bool good = pl.nstage >= 2 && pl.nstage <= kMixMaxStages && pl.len[0] == n && pl.len[pl.nstage] == 1;
        for (int s = 0; s < pl.nstage; ++s) {
            prod *= pl.radix[s];
            good = good && mix_radix_ok(pl.radix[s]) && (s == 0 || pl.radix[s] >= pl.radix[s - 1]) && pl.len[s + 1] * pl.radix[s] == pl.len[s];
            for (int a : {0, 1, pl.len[s] - 1, n - 1})
                good = good && mix_div(a, pl.mg_radix[s]) == a / pl.radix[s] && mix_div(a, pl.mg_sub[s]) == a / pl.len[s + 1] &&
                       mix_div(a, pl.mg_nb[s]) == a / (n / pl.radix[s]);
        }
        if (!good || prod != n || pl.maxr != pl.radix[pl.nstage - 1]) { printf("planner: bad plan for n=%d\n", n); ++bad; }
    }
    printf("planner: %d lengths planned\n", planned);
    MixPlan q;
    if (mix_make_plan(12, q) || mix_make_plan(16, q) || mix_make_plan(17, q) || mix_make_plan(1024 * 17, q) || mix_make_plan(2 * 23, q) || !mix_make_plan(2 * 19, q)) { printf("planned an unsupported length\n"); ++bad; }
    printf(bad ? "FAILED (%d)\n" : "all ok\n", bad);
    return bad != 0;
}
