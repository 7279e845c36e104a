// Host side of the mixed-radix path that does not depend on the precision: the table of factorisations (fft_mixed.h).  The kernels are
// instantiated in fft_mixed_rows_f32.hip ... fft_mixed_cols_f64.hip from fft_mixed_kernels.h.
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>

#include "fft_mixed.h"
#include "pm_internal.h"

namespace pm {

// the factorisations of every length, built on first use (a depth-first search per length: ~1 ms in all), one table per cap on the
// largest factor (the kernel classes 10 / 16 / 20: the knob mix_maxr keeps the planner within a class when the length has a plan there)
struct MixFactors {
    unsigned char nstage[kMixMaxN + 1];
    unsigned char radix[kMixMaxN + 1][kMixMaxStages];
    MixFactors(int maxr, double w20, double w16 = 1.15) {
        for (int n = 0; n <= kMixMaxN; ++n) {
            int r[kMixMaxStages], ns = 0;
            nstage[n] = 0;
            if (n >= 2 && mix_factor(n, r, &ns, maxr, w20, w16) && ns >= 2) {
                nstage[n] = (unsigned char)ns;
                for (int s = 0; s < ns; ++s) radix[n][s] = (unsigned char)r[s];
            }
        }
    }
};
// es: bytes per element (8: complex64, 16: complex128) -- the weight of the class of 20 differs (fft_mixed.h mix_factor)
static const MixFactors& mix_factors(int maxr = kMixMaxRadix, size_t es = 16) {
    static const MixFactors f20d(20, 1.4), f20s(20, 1.3), f16(16, 1.4), f10(10, 1.4);
    return maxr <= 10 ? f10 : (maxr <= 16 ? f16 : (es == 8 ? f20s : f20d));
}

bool mix_length(int64_t n) { return n >= 2 && n <= kMixMaxN && mix_factors().nstage[n] != 0; }

bool mix_plan_for(int n, size_t es, MixPlan& p) {
    if (!mix_length(n)) return false;
    const MixFactors* f = &mix_factors(tuning().mix_maxr, es);
    if (f->nstage[n] == 0) f = &mix_factors(kMixMaxRadix, es);
    int r[kMixMaxStages];
    for (int s = 0; s < f->nstage[n]; ++s) r[s] = f->radix[n][s];
    mix_fill_plan(n, r, f->nstage[n], p);
    return true;
}

// ---------------------------------------------------------------- LDS padding of a launch shape (MixShape pad0 / pad1)
// A model of the LDS banks (MI355X_MICROARCH.md, LDS table) run over the accesses of the first waves of every stage: a complex64
// element moves as ds_read_b64 (two groups of 32 lanes, 64 banks of 4 bytes) / ds_write_b64 (four groups of 16 lanes, 32 banks), a
// complex128 element as the 16-byte forms (groups of 16 / 8 lanes); every extra distinct address on a bank inside a group costs one
// LDS cycle.  Unpadded it reproduces the counters: 3000 = 10 x 15 x 20 complex64 rows 1.14 conflict cycles per conflict-free cycle
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE measured 56 %: 1.14 / 2.14 = 53 %), columns 0.50 (measured 32 %: 33 %), complex128
// 0.54 - 0.67 (36 - 38 %: 35 - 40 %).  The pads with the fewest modelled cycles that keep the workgroups per CU are taken.
namespace {
struct LdsCost { long cycles, base; };
static void lds_group_cost(const long* addr, const bool* on, int lo, int hi, int dwords, int nbanks, LdsCost& c) {
    int count[64];
    long seen[64][8];
    for (int b = 0; b < nbanks; ++b) count[b] = 0;
    bool any = false;
    for (int l = lo; l < hi; ++l) {
        if (!on[l]) continue;
        any = true;
        for (int d = 0; d < dwords; ++d) {
            const long w = addr[l] / 4 + d;
            const int b = int(w % nbanks);
            bool dup = false;
            for (int i = 0; i < count[b] && i < 8; ++i) dup = dup || seen[b][i] == w;
            if (!dup) {
                if (count[b] < 8) seen[b][count[b]] = w;
                ++count[b];
            }
        }
    }
    if (!any) return;
    int worst = 1;
    for (int b = 0; b < nbanks; ++b) worst = count[b] > worst ? count[b] : worst;
    c.cycles += worst;
    c.base += 1;
}
static void lds_access_cost(const long* addr, const bool* on, size_t es, bool read, LdsCost& c) {
    const int dwords = int(es / 4);
    if (es == 8) {
        if (read) { lds_group_cost(addr, on, 0, 32, dwords, 64, c); lds_group_cost(addr, on, 32, 64, dwords, 64, c); }
        else for (int g = 0; g < 4; ++g) lds_group_cost(addr, on, g * 16, g * 16 + 16, dwords, 32, c);
    } else {
        if (read) for (int g = 0; g < 4; ++g) lds_group_cost(addr, on, g * 16, g * 16 + 16, dwords, 64, c);
        else for (int g = 0; g < 8; ++g) lds_group_cost(addr, on, g * 8, g * 8 + 8, dwords, 32, c);
    }
}
static LdsCost mix_lds_model(const MixPlan& p, const MixShape& sh, size_t es, bool col) {
    LdsCost c{0, 0};
    const int n = p.n, ns = p.nstage;
    auto slot = [&](int pt) { return pt + sh.pad1 * (ns >= 3 ? pt / p.len[2] : 0) + sh.pad0 * (pt / p.len[1]); };
    auto at = [&](int sl, int pt) { const long s = slot(pt); return long(col ? s * sh.seqs + sl : long(sl) * sh.npad + s) * long(es); };
    for (int s = 0; s < ns; ++s) {
        const int R = p.radix[s], nb = n / R, total = sh.seqs * nb, sub = p.len[s + 1], L = sub * R;
        const int waves = (total + 63) / 64 < 12 ? (total + 63) / 64 : 12;     // the pattern repeats: the first waves are a fair sample
        for (int w = 0; w < waves; ++w) {
            int sl[64], base[64], step[64];
            bool on[64];
            for (int l = 0; l < 64; ++l) {
                const int b = w * 64 + l;
                on[l] = b < total;
                if (!on[l]) continue;
                const int ja = col ? b / sh.seqs : b % nb;
                sl[l] = col ? b % sh.seqs : b / nb;
                if (s < ns - 1) {
                    base[l] = (ja / sub) * L + ja % sub;
                    step[l] = sub;
                } else {
                    int rem = ja, pos = 0;
                    for (int i = 0; i < ns - 1; ++i) {
                        pos += (rem % p.radix[i]) * p.len[i + 1];
                        rem /= p.radix[i];
                    }
                    base[l] = pos;
                    step[l] = 1;
                }
            }
            for (int k = 0; k < R; ++k) {
                long addr[64];
                for (int l = 0; l < 64; ++l) addr[l] = on[l] ? at(sl[l], base[l] + k * step[l]) : 0;
                if (s > 0) lds_access_cost(addr, on, es, true, c);
                if (s < ns - 1) lds_access_cost(addr, on, es, false, c);
            }
        }
    }
    return c;
}
}  // namespace

void mix_pick_pads(const MixPlan& p, size_t es, bool col, MixShape& sh) {
    mix_shape_pads(p, sh, 0, 0);
    if (!tuning().mix_pad) return;
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int>, std::pair<int, int>> cache;
    int plan_id = p.nstage;      // the factorisation follows the knob mix_maxr: the pads belong to the plan, not just to the length
    for (int s = 0; s < p.nstage; ++s) plan_id = plan_id * 21 + p.radix[s];
    const auto key = std::make_tuple(p.n, int(es), int(col), sh.seqs, plan_id);
    {
        std::lock_guard<std::mutex> lk(mu);
        const auto it = cache.find(key);
        if (it != cache.end()) {
            mix_shape_pads(p, sh, it->second.first, it->second.second);
            return;
        }
    }
    // keep the workgroups per CU of the unpadded shape (160 KiB of LDS, 1 KiB of slack), and the hard per-workgroup limit
    const size_t plain = size_t(sh.seqs) * size_t(p.n) * es;
    const size_t per_cu = size_t(160) * 1024, k = per_cu / (plain ? plain : 1);
    size_t budget = k ? per_cu / k - 1024 : plain;
    if (budget > size_t(156) * 1024) budget = size_t(156) * 1024;
    if (budget < plain) budget = plain;
    int best0 = 0, best1 = 0;
    double best = 1e300;
    for (int c1 = 0; c1 <= 4; ++c1) {
        if (c1 && p.nstage < 3) break;
        for (int c0 = 0; c0 <= 8; ++c0) {
            MixShape t = sh;
            mix_shape_pads(p, t, c0, c1);
            if (size_t(t.seqs) * size_t(t.npad) * es > budget) continue;
            const LdsCost c = mix_lds_model(p, t, es, col);
            const double r = c.base ? double(c.cycles) / double(c.base) : 1.0;
            if (r < best - 1e-9) {
                best = r;
                best0 = c0;
                best1 = c1;
            }
        }
    }
    mix_shape_pads(p, sh, best0, best1);
    std::lock_guard<std::mutex> lk(mu);
    cache[key] = {best0, best1};
}

}  // namespace pm
