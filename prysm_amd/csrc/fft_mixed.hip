// Host side of the mixed-radix path that does not depend on the precision: the table of factorisations (fft_mixed.h).  The kernels are
// instantiated in fft_mixed_rows_f32.hip ... fft_mixed_cols_f64.hip from fft_mixed_kernels.h.
#include "fft_mixed.h"
#include "pm_internal.h"

namespace pm {

// the factorisations of every length, built on first use (a depth-first search per length: ~1 ms in all)
struct MixFactors {
    unsigned char nstage[kMixMaxN + 1];
    unsigned char radix[kMixMaxN + 1][kMixMaxStages];
    MixFactors() {
        for (int n = 0; n <= kMixMaxN; ++n) {
            int r[kMixMaxStages], ns = 0;
            nstage[n] = 0;
            if (n >= 2 && mix_factor(n, r, &ns) && ns >= 2) {
                nstage[n] = (unsigned char)ns;
                for (int s = 0; s < ns; ++s) radix[n][s] = (unsigned char)r[s];
            }
        }
    }
};
static const MixFactors& mix_factors() {
    static const MixFactors f;
    return f;
}

bool mix_length(int64_t n) { return n >= 2 && n <= kMixMaxN && mix_factors().nstage[n] != 0; }

bool mix_plan_for(int n, MixPlan& p) {
    if (!mix_length(n)) return false;
    const MixFactors& f = mix_factors();
    int r[kMixMaxStages];
    for (int s = 0; s < f.nstage[n]; ++s) r[s] = f.radix[n][s];
    mix_fill_plan(n, r, f.nstage[n], p);
    return true;
}

}  // namespace pm
