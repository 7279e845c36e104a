// Host side of the mixed-radix path that does not depend on the precision: the table of factorisations (fft_mixed.h).  The kernels are
// instantiated in fft_mixed_rows_f32.hip ... fft_mixed_cols_f64.hip from fft_mixed_kernels.h.
#include "fft_mixed.h"
#include "pm_internal.h"

namespace pm {

// the factorisations of every length, built on first use (a depth-first search per length: ~1 ms in all), one table per cap on the
// largest factor (the kernel classes 10 / 16 / 20: the knob mix_maxr keeps the planner within a class when the length has a plan there)
struct MixFactors {
    unsigned char nstage[kMixMaxN + 1];
    unsigned char radix[kMixMaxN + 1][kMixMaxStages];
    MixFactors(int maxr, double w20) {
        for (int n = 0; n <= kMixMaxN; ++n) {
            int r[kMixMaxStages], ns = 0;
            nstage[n] = 0;
            if (n >= 2 && mix_factor(n, r, &ns, maxr, w20) && ns >= 2) {
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

}  // namespace pm
