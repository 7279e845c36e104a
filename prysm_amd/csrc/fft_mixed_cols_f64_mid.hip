// mixed-radix cols pass, complex128: the kernel class of factors up to 20 (fft_mixed_kernels.h)
#include "fft_mixed_kernels.h"

namespace pm {

template <> int mix_cols_launch<double, 20>(const MixPlan* p, MixShape sh, const DirectIn<double>& in, const ColStoreNat<double>& out, const cx<double>* tw, int log_g, int groups, int nt, size_t lds,
                                  hipStream_t st) {
    return mix_cols_launch_impl<double, 20>(p, sh, in, out, tw, log_g, groups, nt, lds, st);
}

}  // namespace pm
